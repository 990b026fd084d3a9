#!/usr/bin/env python3
"""bench.py — local-BA LM iterations/s on a 50 KF x 20k MapPoint window (BASELINE.json).

A "step" is one complete local-BA solve (what Optimizer::LocalBundleAdjustment runs at
/root/reference/src/Optimizer.cc:754-755 plus the gate at :757-775) of one synthetic
covisibility window per rank through movba_lba_solve: SURVEY.md 8(d)'s timed region, from
"flattened host arrays ready" to "poses, points, chi2 and outlier flags back in host
memory" (structure pass, H2D, every launch, D2H) — the same region the CPU baseline is
timed on.  The metric counts the linear solves (accepted + rejected LM trials) those steps
performed.  The rate with the window already resident in HBM (movba_lba_run alone) is
reported beside it as config.resident_window, never as `value`.  With N > 1 every rank owns
an independent window (BASELINE cfg5: weak scaling, no data-path collective) and the
optimised keyframe poses are all-gathered over RCCL inside the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
ROCPROF_NAME = {"k_schur": "movba::k_schur", "k_pcg": "movba::k_pcg_rows", "k_point<backsub>": "movba::k_point<true"}


def measured_traffic(kernel_class):
    """HBM-side bytes per launch of a kernel class from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc_traffic.json, made by scripts/profile_gpu.sh + scripts/summarise_profiles.py on the
    same bench command: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, launches with work only).
    FETCH_SIZE under-reports 16-byte-per-lane loads by 2x on gfx950: the summary's corrected figure is used when present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    for name, v in d["kernels"].items():
        if ROCPROF_NAME.get(kernel_class, "?") in name:
            # corrected as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE counts half the bytes of 16-byte-per-lane
            # loads; the share of such loads per kernel is read off the ISA (scripts/isa_load_widths.py), the summary keeps both
            return v.get("hbm_bytes_per_launch_corrected", v["hbm_bytes_per_launch_raw"]), os.path.basename(files[-1])
    return None, None

KERNEL_CLASSES = ["k_schur", "k_pcg", "k_point<backsub>", "k_decide", "setup(init+linearize+lambda)", "k_finalize"]


def algorithmic_bytes(K, F, P, E):
    """SURVEY.md §8(d) B_iter split over the three kernels of one LM trial:
    compulsory fp64 traffic with Jacobians recomputed, dense upper-block reduced system."""
    S = K * (K + 1) // 2 * 288 + K * 48
    per = {
        "k_schur": E * 32 + P * 24 + (K + F) * 56 + S,                 # edge pass 1, writes S and rhs
        "k_pcg": S + K * 56,                                           # reads S and rhs, writes poses
        "k_point<backsub>": E * (32 + 8) + P * (24 + 24) + (K + F) * 56,   # edge pass 2, chi2 write
    }
    per["B_iter"] = sum(per.values())
    return per


def main_windows(args):
    """--windows W: the W windows of BASELINE cfg5 (cfg3 shape, seeds 2000 ...) whatever the number of ranks; rank r takes windows
    r, r + N, ... and solves them in ONE batched run per step (movba_lba_upload each, movba_lba_run_batch, movba_lba_download
    each: host arrays in -> host arrays out for all of them), then the poses of all W windows are all-gathered (RCCL).  Strong
    scaling: the job is the same at every N.  value = LM iterations of all W windows per second of the slowest rank."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from movba import capi, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the local-BA path has no CPU fallback")
    if args.windows % world:
        raise SystemExit("--windows must be a multiple of the number of ranks (equal blocks for the all-gather)")
    rehearsal = os.environ.get("MOVBA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world, **({} if rehearsal else {"device_id": dev}))
    shape = (50, 10, 20000, 2, 10)
    mine = shard.windows_for_rank(args.windows, rank, world)
    ws = [synth.make_window(shape[0], shape[1], shape[2], shard.window_seed(i), run_lo=shape[3], run_hi=shape[4]) for i in mine]
    NP = ws[0].n_poses
    stream = torch.cuda.Stream(device=dev)                  # the handles of a batch share one (non-default) stream
    pose_buf = torch.empty((len(ws), NP, 7), dtype=torch.float64, device=dev)
    solvers = []
    for k, w in enumerate(ws):
        sv = capi.Solver(device=local_rank, stream=stream.cuda_stream)
        sv.set_pose_export(pose_buf[k].data_ptr(), NP * 7 * 8)
        sv.prepare(w, pinned=True)
        solvers.append(sv)

    def step():
        for sv in solvers:
            sv.upload_prepared()
        capi.run_batch(solvers)
        for sv in solvers:
            sv.download_prepared()
        if world > 1:
            stream.synchronize()
            return shard.gather_poses(pose_buf.cpu()) if rehearsal else shard.gather_poses(pose_buf)
        return pose_buf.unsqueeze(0)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    barrier()
    dt = time.perf_counter() - t0
    res = [sv.download_prepared(pack=True) for sv in solvers]
    solves_local = sum(r["n_solves"] for r in res) * args.steps
    red_dev = torch.device("cpu") if rehearsal else dev
    tt = torch.tensor([dt], dtype=torch.float64, device=red_dev); ss = torch.tensor([float(solves_local)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX); dist.all_reduce(ss, op=dist.ReduceOp.SUM)
    dt_max, solves_total = float(tt.item()), float(ss.item())
    if rank == 0 and os.environ.get("MOVBA_BENCH_DUMP_POSES"):
        np.save(os.environ["MOVBA_BENCH_DUMP_POSES"], gathered.detach().cpu().numpy())
    if rank == 0:
        K, F, P = ws[0].n_free, ws[0].n_poses - ws[0].n_free, ws[0].n_points
        E_all = [w.n_edges for w in ws]
        b_iter = sum(algorithmic_bytes(K, F, P, e)["B_iter"] for e in E_all) / len(E_all)
        chain = b_iter * solves_total / dt_max / 1e9
        out = {"metric": "local-BA iterations/sec (50 KF x 20k MapPoint window)", "value": solves_total / dt_max, "unit": "LM iterations/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{args.windows} LocalBundleAdjustment windows of 50 KF + 10 fixed x 20 000 MapPoints (BASELINE cfg5: seeds 2000-{2000 + args.windows - 1}), "
                                      f"{len(ws)} per GPU in one batched run (movba_lba_run_batch), Huber on, 10 LM iterations",
                          "windows_per_rank": len(ws), "window_solves_per_s": args.windows * args.steps / dt_max,
                          "parallelism": f"{world} rank(s) x {len(ws)} window(s); " + ("gloo pose all-gather, all ranks on ONE GPU (rehearsal)" if rehearsal else "RCCL pose all-gather of all windows") if world > 1 else f"1 rank x {len(ws)} windows, no collective",
                          "timed_region": "per window movba_lba_upload (host arrays in) -> one movba_lba_run_batch -> movba_lba_download (results in host memory), then the pose all-gather",
                          "note": "opt-in mode (--windows): the default bench line (one window per rank, weak scaling) is the one SCALE compares across N; "
                                  "no 2/4/8-GPU curve of either mode has been measured by this repository (one-GPU boxes)"},
               "roofline": {"bound": "hbm", "kernel": "whole batched LM iteration (all kernels + launch gaps)", "achieved": chain, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": chain / HBM_PEAK_GBS, "traffic": None, "B_iter_bytes": b_iter}}
        print(json.dumps(out), flush=True)
    for sv in solvers:
        sv.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the informational extras (resident / batched / adapter runs): profiling passes")
    ap.add_argument("--windows", type=int, default=0,
                    help="opt-in: BASELINE cfg5 as SURVEY 8(e) words it - ALWAYS this many windows (8: seeds 2000-2007), dealt round-robin "
                         "to the ranks and solved by each rank in one batched run (strong scaling); default 0 = one window per rank")
    args = ap.parse_args()
    if args.windows > 0:
        return main_windows(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from movba import capi, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the local-BA path has no CPU fallback")
    # MOVBA_BENCH_REHEARSAL=1: the N > 1 code path on a box with ONE GPU (every rank on device 0, gloo instead of RCCL,
    # which refuses two ranks on one device): exercises launch, sharding, timing and the JSON line, not the interconnect
    rehearsal = os.environ.get("MOVBA_BENCH_REHEARSAL") == "1"
    # MOVBA_BENCH_RCCL_WORLD1=1: a ONE-rank run that still initialises the `nccl` (= RCCL) process group and sends the poses
    # through all_gather_into_tensor inside the timed region: a real ncclCommInitRank and a real RCCL kernel behind the solve,
    # on a box with one GPU (the N > 1 branch below is otherwise never taken there)
    rccl_world1 = os.environ.get("MOVBA_BENCH_RCCL_WORLD1") == "1" and world == 1
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or rccl_world1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- workload: cfg3 on rank 0 (the configuration the metric is quoted on), cfg5 seeds on the others ----
    shape = dict(cfg2=(10, 2, 2000, 2, 6), cfg3=(50, 10, 20000, 2, 10))[args.config]
    seed = (1003 if args.config == "cfg3" else 1002) if rank == 0 else shard.window_seed(rank)
    w = synth.make_window(shape[0], shape[1], shape[2], seed, run_lo=shape[3], run_hi=shape[4])
    K, F, P, E = w.n_free, w.n_poses - w.n_free, w.n_points, w.n_edges

    stream = torch.cuda.current_stream(dev)
    solver = capi.Solver(device=local_rank, stream=stream.cuda_stream)
    pose_buf = torch.empty((1, w.n_poses, 7), dtype=torch.float64, device=dev)

    solver.set_pose_export(pose_buf.data_ptr(), pose_buf.numel() * 8)     # the solve's last kernel leaves the poses here
    solver.prepare(w, pinned=True)                       # descriptor + result buffers once, like a C++ caller's own (the adapter's:
                                                         # movba_host_alloc memory, which the solve's last kernel writes into directly)

    def step():
        # host arrays in -> structure pass + H2D + whole LM loop on the device + D2H -> host arrays out
        solver.solve_prepared(pack=False)
        if world > 1:
            return shard.gather_poses(pose_buf.cpu()) if rehearsal else shard.gather_poses(pose_buf)
        if rccl_world1:
            return shard.gather_poses(pose_buf, force_collective=True)
        return pose_buf

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    # untimed pass with every kernel class bracketed by HIP events: find the dominant kernel
    solver.set_profile_mask(0x3f); solver.reset_profile()
    step(); torch.cuda.synchronize(dev)
    prof_all = solver.profile()["kernels"]
    dominant = max(KERNEL_CLASSES[:3], key=lambda k: prof_all[k]["ms"])
    solver.set_profile_mask(1 << KERNEL_CLASSES.index(dominant)); solver.reset_profile()

    # HIP events bracket the dominant kernel's launches during the first tenth of the timed steps only (20 launches at the
    # default 20 steps): an event pair per launch costs the solve ~2.5 us of GPU-side serialisation each - 5 % of a step that
    # carries them (scripts/event_cost.py: 1.174 / 1.186 / 1.235 ms per step with events on 0 / 5 / 20 of 20 steps)
    n_evt = max(1, args.steps // 10)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == n_evt:
            solver.set_profile_mask(0)
        gathered = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = solver.profile()
    res = solver.solve_prepared()                        # (untimed) the same solve once more, results unpacked
    solves_local = res["n_solves"] * args.steps         # the same window every step: bit-identical solves

    red_dev = torch.device("cpu") if rehearsal else dev
    tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    ss = torch.tensor([float(solves_local)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ss, op=dist.ReduceOp.SUM)
    dt_max, solves_total = float(tt.item()), float(ss.item())

    if rank == 0 and os.environ.get("MOVBA_BENCH_DUMP_POSES"):
        # (tests: the gathered poses of the last timed step, one block per rank)
        np.save(os.environ["MOVBA_BENCH_DUMP_POSES"], gathered.detach().cpu().numpy())
    if rank == 0:
        ab = algorithmic_bytes(K, F, P, E)
        dk = prof["kernels"][dominant]
        avg_s = dk["ms"] / max(dk["launches"], 1) * 1e-3
        achieved = ab[dominant] / avg_s / 1e9 if avg_s > 0 else 0.0
        chain_gbs = ab["B_iter"] * res["n_solves"] * args.steps / dt / 1e9
        traffic, traffic_src = measured_traffic(dominant)
        per_kernel = {}
        for kc in KERNEL_CLASSES[:3]:
            v = prof_all[kc]
            a_s = v["ms"] / max(v["launches"], 1) * 1e-3
            gbs = ab[kc] / a_s / 1e9 if a_s > 0 else 0.0
            tr, _ = measured_traffic(kc)
            per_kernel[kc] = {"avg_launch_us": a_s * 1e6, "algorithmic_bytes_per_launch": ab[kc], "achieved": gbs,
                              "frac": gbs / HBM_PEAK_GBS, "traffic": tr}
        out = {
            "metric": "local-BA iterations/sec (50 KF x 20k MapPoint window)" if args.config == "cfg3"
                      else "local-BA iterations/sec (10 KF x 2k MapPoint window)",
            "value": solves_total / dt_max,
            "unit": "LM iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"LocalBundleAdjustment {K} KF + {F} fixed x {P} MapPoints, E={E} mono edges, Huber on, "
                                   f"10 LM iterations ({args.config}, seed {seed}); one window per GPU",
                       "lm_iterations_per_step": res["n_solves"], "pcg_iterations_per_step": res["pcg_iters"],
                       "window_solves_per_s": world * args.steps / dt_max,
                       "parallelism": (f"{world} independent window(s), " + ("gloo pose all-gather, all ranks on ONE GPU (rehearsal)" if rehearsal else "RCCL pose all-gather")) if world > 1
                                      else ("1 window, RCCL pose all-gather over a one-rank nccl group (MOVBA_BENCH_RCCL_WORLD1)" if rccl_world1 else "1 window")},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ab[dominant], "avg_launch_us": avg_s * 1e6,
                         "launches_timed": dk["launches"],
                         "launches_timed_note": f"HIP events around every launch of the dominant kernel during the first {n_evt} of the {args.steps} timed steps",

                         "chain": {"B_iter_bytes": ab["B_iter"], "achieved": chain_gbs, "frac": chain_gbs / HBM_PEAK_GBS,
                                   "note": "whole LM iteration (all kernels + launch gaps); latency-bound, not bandwidth-bound"},
                         "per_kernel": per_kernel, "per_kernel_note": "HIP events around every launch of one untimed solve",
                         "kernel_ms_per_step_all_classes": {k: v["ms"] for k, v in prof_all.items()}},
        }
        out["per_kernel"] = per_kernel
        out["config"]["timed_region"] = ("movba_lba_solve: host arrays in -> structure pass, H2D, all launches, D2H -> results in "
                                         "host memory (SURVEY 8d; result arrays in movba_host_alloc memory, as the adapter's are); identical region "
                                         "for cpu_baseline")
        out["config"]["host_phase_ms_per_step"] = {k: prof[k] / args.steps for k in ("structure_ms", "upload_ms", "download_ms")}
        extras = world == 1 and not args.no_extras
        if extras:
            # for information only (never `value`): the LM loop alone with the window already resident in HBM
            # (movba_lba_upload once, then movba_lba_run per step)
            solver.set_profile_mask(0)
            solver.upload(w)
            solver.run(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                solver.run()
            torch.cuda.synchronize(dev)
            tr_ = time.perf_counter() - t1
            out["config"]["resident_window"] = {"lm_iterations_per_s": res["n_solves"] * args.steps / tr_,
                                                "ms_per_window_solve": 1e3 * tr_ / args.steps,
                                                "note": "movba_lba_run only; no structure pass, H2D or D2H in the region"}
        if extras:
            # for information only (never `value`): movba_lba_run_batch over 8 resident cfg5-shaped windows on this one GPU
            # (multi-session serving): one launch per kernel over all windows, groups of windows out of phase on streams
            try:
                nb = 8
                bws = [synth.make_window(shape[0], shape[1], shape[2], shard.window_seed(i), run_lo=shape[3], run_hi=shape[4]) for i in range(nb)]
                bstream = torch.cuda.Stream(device=dev)     # the handles of a batch share one (non-default) stream
                bsolvers = [capi.Solver(device=local_rank, stream=bstream.cuda_stream) for _ in range(nb)]
                for bs_, bw_ in zip(bsolvers, bws):
                    bs_.upload(bw_)
                capi.run_batch(bsolvers); capi.run_batch(bsolvers)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(5):
                    capi.run_batch(bsolvers)
                torch.cuda.synchronize(dev)
                tb = (time.perf_counter() - t1) / 5
                # the SAME eight resident windows one after the other (movba_lba_run each): what the batch is compared with
                for bs_ in bsolvers:
                    bs_.run()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(5):
                    for bs_ in bsolvers:
                        bs_.run()
                torch.cuda.synchronize(dev)
                tseq = (time.perf_counter() - t1) / 5
                lm = sum(bs_.download()["n_solves"] for bs_ in bsolvers)
                out["config"]["batched_windows"] = {"n": nb, "ms_per_batch": 1e3 * tb, "window_solves_per_s": nb / tb,
                                                    "lm_iterations_per_s": lm / tb, "ms_one_after_the_other": 1e3 * tseq,
                                                    "vs_one_resident_window_at_a_time": tseq / tb,
                                                    "note": "resident windows (seeds 2000-2007), bit-identical to their solo solves; compared "
                                                            "with movba_lba_run on the same eight windows one after the other"}
                for bs_ in bsolvers:
                    bs_.close()
            except Exception as exc:                        # context only
                out["config"]["batched_windows"] = {"error": str(exc)}
        if extras and args.config == "cfg3":
            # for information only (never `value`): the same solve on cfg3-sized windows of the covisibility patterns the
            # reference produces (movba/synth.py: shuffled ids, a path that comes back, a hub where every pair shares points)
            # and cfg3 itself on the one-launch direct solver
            try:
                pats = {}
                def time_window(sv, ww):
                    sv.prepare(ww, pinned=True)
                    for _ in range(2):
                        sv.solve_prepared(pack=False)
                    t1 = time.perf_counter()
                    for _ in range(5):
                        sv.solve_prepared(pack=False)
                    tc = (time.perf_counter() - t1) / 5
                    rr = sv.solve_prepared()
                    sv.upload(ww); sv.run(); torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    for _ in range(5):
                        sv.run()
                    torch.cuda.synchronize(dev)
                    trs = (time.perf_counter() - t1) / 5
                    return {"ms_per_window_solve": 1e3 * tc, "ms_resident": 1e3 * trs, "lm_iterations_per_step": rr["n_solves"],
                            "pcg_iterations_per_step": rr["pcg_iters"], "direct_trials": rr["n_direct"], "E": ww.n_edges}
                ps = capi.Solver(device=local_rank, stream=stream.cuda_stream)
                for pname in ("shuffled", "revisit", "hub"):
                    pats[pname] = time_window(ps, synth.pattern_cfg(pname))
                ps.close()
                pd = capi.Solver(device=local_rank, stream=stream.cuda_stream, direct=True)
                pats["cfg3_on_the_direct_solver"] = time_window(pd, w)
                pd.close()
                pats["note"] = ("cfg3-sized windows, same timed regions as `value` (ms_per_window_solve) and config.resident_window (ms_resident); "
                                "direct_trials > 0: the one-launch dense Cholesky solved those trials (dense covisibility)")
                out["config"]["covisibility_patterns"] = pats
            except Exception as exc:                        # context only
                out["config"]["covisibility_patterns"] = {"error": str(exc)}
        if extras:
            # for information only (never `value`): the other BASELINE configurations in the same run - cfg2 (LocalBundleAdjustment
            # 10 KF x 2k MapPoints) through the whole movba_lba_solve call, cfg1 (PoseOptimization on one Frame of 500 matches) through
            # movba_pose_opt with and without the hypothesis stage - each beside the single-threaded oracle on this box's host
            try:
                from oracle import oracle as _orc            # (CPU legs of the extras: the oracle as the timed "port", as below)
                oc = {}
                w2 = synth.cfg("cfg2")
                s2 = capi.Solver(device=local_rank, stream=stream.cuda_stream)
                s2.prepare(w2, pinned=True)
                for _ in range(3):
                    s2.solve_prepared(pack=False)
                t1 = time.perf_counter()
                for _ in range(20):
                    s2.solve_prepared(pack=False)
                t2 = (time.perf_counter() - t1) / 20
                r2 = s2.solve_prepared()
                r2 = {"n_solves": int(r2["n_solves"]), "poses": np.array(r2["poses"])}      # (copies: the arrays are views of the handle's pinned memory)
                s2.close()
                c2 = {"ms_per_window_solve": 1e3 * t2, "lm_iterations_per_s": r2["n_solves"] / t2, "lm_iterations_per_step": r2["n_solves"], "E": w2.n_edges}
                if not args.no_cpu_baseline:
                    _orc.build()
                    t1 = time.perf_counter()
                    for _ in range(5):
                        o2 = _orc.solve(w2)
                    tc = (time.perf_counter() - t1) / 5
                    c2["cpu_1thread_ms_per_window_solve"] = 1e3 * tc
                    c2["pose_max_abs_vs_oracle"] = float(np.abs(r2["poses"] - o2["poses"]).max())
                oc["cfg2_local_ba_10kf_2k"] = c2
                f1 = synth.make_frame(n=500)
                hub, gate = float(np.float32(np.sqrt(5.991))), 5.991
                s1 = capi.Solver(device=local_rank, stream=stream.cuda_stream)
                def tpose(**kw):
                    for _ in range(5):
                        rr = s1.pose_opt(f1["Xw"], f1["obs"], f1["pose0"], f1["cam"], hub, gate, **kw)
                    ts = []
                    for _ in range(40):
                        t1 = time.perf_counter(); rr = s1.pose_opt(f1["Xw"], f1["obs"], f1["pose0"], f1["cam"], hub, gate, **kw); ts.append(time.perf_counter() - t1)
                    ts.sort()
                    return 1e3 * ts[len(ts) // 2], rr
                m_lm, r_lm = tpose()
                m_hyp, r_hyp = tpose(ransac_iters=50, ransac_seed=7, confidence=0.95, lo_iters=10)
                s1.close()
                c1 = {"ms_motion_only_lm": m_lm, "ms_full_pipeline": m_hyp, "inliers": int(r_hyp["n_inliers"]), "matches": 500,
                      "full_pipeline": "50 P3P hypotheses scored by sigma-consensus, confidence 0.95 as the stopping rule, one LO step, then the 4 x 10 motion-only LM"}
                if not args.no_cpu_baseline:
                    t1 = time.perf_counter()
                    for _ in range(20):
                        _orc.pose_opt(f1["Xw"], f1["obs"], f1["pose0"], f1["cam"], hub, gate)
                    c1["cpu_1thread_ms_motion_only_lm"] = 1e3 * (time.perf_counter() - t1) / 20
                    # the same work as ms_full_pipeline on one host core: the oracle's hypothesis stage over the same samples
                    # (stopping rule, LO step), then its motion-only LM from the pose that stage returns
                    smp = capi.ransac_samples(500, 50, 7)
                    t1 = time.perf_counter()
                    for _ in range(20):
                        hy = _orc.pose_ransac(f1["Xw"], f1["obs"], f1["pose0"], f1["cam"], gate, smp, confidence=0.95, lo_its=10)
                        po = _orc.pose_opt(f1["Xw"], f1["obs"], hy["pose"] if hy["n_inliers"] >= 4 else f1["pose0"], f1["cam"], hub, gate)
                    c1["cpu_1thread_ms_full_pipeline"] = 1e3 * (time.perf_counter() - t1) / 20
                    c1["pose_max_abs_vs_oracle_full_pipeline"] = float(np.abs(np.array(r_hyp["pose"]) - po["pose"]).max())
                oc["cfg1_pose_optimization_500"] = c1
                oc["note"] = "whole-call wall times through the C-ABI, pinned result arrays, same timed region as `value` for cfg2"
                out["config"]["other_baseline_configs"] = oc
            except Exception as exc:                        # context only
                out["config"]["other_baseline_configs"] = {"error": str(exc)}
        if extras:
            # for information only: host-side cost of the Optimizer.h adapter around the solve on this window (mock map classes,
            # mov-slam_amd/host/adapter_test): window selection + flattening, and the write-back under the map mutex
            try:
                import struct, subprocess, tempfile
                host = os.path.join(ROOT, "mov-slam_amd", "host")
                exe = os.path.join(host, "adapter_test")
                if not os.path.exists(exe):
                    subprocess.check_call(["make", "-C", host, "-s"])
                with tempfile.TemporaryDirectory() as tmp:
                    fin, fout = os.path.join(tmp, "w.bin"), os.path.join(tmp, "o.bin")
                    wf = w
                    with open(fin, "wb") as fh:
                        fh.write(struct.pack("4i", wf.n_poses, wf.n_points, wf.n_edges, 0))
                        for arr, dt in ((wf.pose_fixed, np.uint8), (wf.poses, np.float64), (wf.points, np.float64), (wf.edge_pose, np.int32),
                                        (wf.edge_point, np.int32), (wf.obs, np.float64)):
                            fh.write(np.ascontiguousarray(arr, dt).tobytes())
                    def adapter_times(extra_env):
                        subprocess.check_call([exe, "lba", fin, fout], env=dict(os.environ, MOVBA_ADAPTER_REPS="4", **extra_env))
                        raw = open(fout, "rb").read()
                        # (timing triple in front of the per-point / per-keyframe / status trailer of adapter_test)
                        off = 20 + 28 * wf.n_poses + 12 * wf.n_points
                        n_er = struct.unpack_from("5i", raw, 0)[3]
                        off += 8 * n_er + 8 + 8 + 20 * wf.n_points
                        return struct.unpack_from("3d", raw, off)
                    tm = adapter_times({})
                    tm_ref = adapter_times({"MOVBA_ADAPTER_REFERENCE_NORMALS": "1"})
                out["config"]["adapter_host_ms"] = {"extraction": tm[0], "solve_call": tm[1], "write_back": tm[2],
                                                    "write_back_with_the_reference_normal_update": tm_ref[2],
                                                    "note": "Optimizer::LocalBundleAdjustment over mock KeyFrame/MapPoint classes (taking the reference's locks), fourth call of the process "
                                                            "on a fresh copy of the map (buffers, arena and handle warm, as in a running system): one "
                                                            "GetObservations() copy per point, normal/depth stored from the GPU result — which needs the one-line SetMinMaxDistance "
                                                            "accessor INTEGRATION.md 1.4 adds to MoV-SLAM's MapPoint.h; write_back_with_the_reference_normal_update is the "
                                                            "figure for an unpatched MapPoint (MapPoint::UpdateNormalAndDepth per point)"}
            except Exception as exc:                        # context only
                out["config"]["adapter_host_ms"] = {"error": str(exc)}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle                       # CPU baseline leg: the oracle as the timed "port"
            oracle.build(force=True)                        # -march=native of THIS box's host cores
            cpu_model = "unknown"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip(); break
            except OSError:
                pass
            n, solves, t_cpu0 = 0, 0, time.perf_counter()
            while True:
                o = oracle.solve(w)
                n += 1; solves += o["n_solves"]
                el = time.perf_counter() - t_cpu0
                if el >= args.cpu_seconds or n >= 200:
                    break
            out["cpu_baseline"] = {"value": solves / el, "unit": "LM iterations/s", "cores": 1, "kind": "port",
                                   "sample": f"{n} full solves of the same window ({solves} LM iterations, {el:.1f} s), "
                                             "single-threaded restated-g2o oracle (oracle/lba_oracle.c, -O3 -march=native)",
                                   "host_cpus": os.cpu_count(), "cpu_model": cpu_model, "ms_per_window_solve": 1e3 * el / n,
                                   "region": "host arrays in -> host arrays out (lba_oracle_solve), as the GPU value"}
            # context: the same restatement with its edge loops spread over the host cores of this box's share
            # (OpenMP build; the dense Cholesky stays serial).  g2o as the reference builds it is single-threaded, so the
            # figure above stays THE baseline.
            try:
                ncore = oracle.set_omp_threads(min(16, oracle.host_core_share()))      # (torch's OpenMP runtime is already up: the environment variable would be ignored)
                oracle.solve(w, omp=True)                   # thread pool start-up
                n2, solves2, t2 = 0, 0, time.perf_counter()
                while True:
                    o2 = oracle.solve(w, omp=True)
                    n2 += 1; solves2 += o2["n_solves"]
                    el2 = time.perf_counter() - t2
                    if el2 >= min(args.cpu_seconds, 5.0) or n2 >= 200:
                        break
                out["cpu_baseline_all_cores"] = {"value": solves2 / el2, "unit": "LM iterations/s", "cores": ncore,
                                                 "kind": "port-openmp", "ms_per_window_solve": 1e3 * el2 / n2,
                                                 "sample": f"{n2} full solves ({el2:.1f} s), OpenMP build of the same oracle source"}
            except Exception as exc:                        # context only: never fail the bench for it
                out["cpu_baseline_all_cores"] = {"error": str(exc)}
            # parity spot-check of what was timed
            out["config"]["parity_vs_oracle"] = {
                "pose_max_abs": float(np.abs(res["poses"] - o["poses"]).max()),
                "outlier_mismatches": int((res["outlier"] != o["outlier"]).sum())}
        print(json.dumps(out), flush=True)
    solver.close()
    if world > 1 or rccl_world1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
