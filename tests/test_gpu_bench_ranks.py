"""bench.py's N > 1 path on the GPU box (one GPU there: MOVBA_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo for
the pose all-gather, RCCL refuses two ranks on one device): launch, one window per rank (BASELINE cfg5: weak scaling, no
data-path collective), barrier + max-over-ranks timing, the gathered poses and the ONE JSON line of rank 0.  The 8-GPU curve
itself is the driver's to measure."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from movba import shard, synth

pytestmark = pytest.mark.gpu


def test_bench_with_two_ranks_gathers_the_poses_of_both_windows(built_lib, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dump = str(tmp_path / "poses.npy")
    env = dict(os.environ, MOVBA_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", MOVBA_BENCH_DUMP_POSES=dump)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)     # a fresh child process, not an exec of this one
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["unit"] == "LM iterations/s" and "roofline" in out and "rehearsal" in out["config"]["parallelism"]
    g = np.load(dump)
    assert g.shape[0] == 2 and g.shape[-1] == 7
    # rank 0 solved cfg3 (seed 1003), rank 1 the cfg5 window of seed 2001: each block equals the solo solve of its window
    sv = built_lib.Solver()
    solves = 0
    try:
        for rank, seed in ((0, 1003), (1, shard.window_seed(1))):
            w = synth.make_window(50, 10, 20000, seed, run_lo=2, run_hi=10)
            r = sv.solve(w)
            assert np.array_equal(g[rank].reshape(-1, 7), r["poses"]), rank
            solves += r["n_solves"]
    finally:
        sv.close()
    # whole-job value = the LM iterations of ALL ranks' steps / the max-over-ranks time of the timed region
    assert abs(out["value"] - solves * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]


def test_bench_sends_the_poses_through_rccl_in_a_one_rank_group(built_lib, tmp_path):
    """The `nccl` branch of bench.py on the hardware there is: MOVBA_BENCH_RCCL_WORLD1=1 initialises the RCCL process group at
    world size 1 (ncclCommInitRank) and gathers the poses with all_gather_into_tensor (an RCCL kernel on the stream behind the
    solve), inside the timed region, in a fresh child process.  What it cannot show is the interconnect: the 2/4/8-GPU curve
    stays the driver's to measure."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dump = str(tmp_path / "poses1.npy")
    env = dict(os.environ, MOVBA_BENCH_RCCL_WORLD1="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MOVBA_BENCH_DUMP_POSES=dump)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and "RCCL" in out["config"]["parallelism"] and out["value"] > 0
    g = np.load(dump)
    sv = built_lib.Solver()
    try:
        r = sv.solve(synth.make_window(50, 10, 20000, 1003, run_lo=2, run_hi=10))
    finally:
        sv.close()
    assert g.shape[0] == 1 and np.array_equal(g[0].reshape(-1, 7), r["poses"])


@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_windows_mode_solves_the_eight_cfg5_windows_whatever_the_rank_count(built_lib, tmp_path, ranks):
    """bench.py --windows 8 (cfg5's eight windows, 8 / N per rank through movba_lba_run_batch, all of their poses
    gathered): with one rank (all eight in one batch, no collective) and with two (rehearsal mode: both ranks on device 0, gloo),
    the gathered poses are the solo solves of seeds 2000 ... 2007 in rank-major order, and the line says strong scaling."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dump = str(tmp_path / "poses8.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MOVBA_BENCH_DUMP_POSES=dump)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--windows", "8", "--steps", "2", "--warmup", "1"]
    if ranks == 1:
        cmd = [sys.executable] + tail
    else:
        env["MOVBA_BENCH_REHEARSAL"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["scaling"] == "strong" and out["config"]["windows_per_rank"] == 8 // ranks and out["value"] > 0
    g = np.load(dump)
    assert g.shape[:2] == (ranks, 8 // ranks) and g.shape[-1] == 7
    sv = built_lib.Solver()
    solves = 0
    try:
        for r in range(ranks):
            for k, wid in enumerate(shard.windows_for_rank(8, r, ranks)):
                res = sv.solve(synth.make_window(50, 10, 20000, shard.window_seed(wid), run_lo=2, run_hi=10))
                assert np.array_equal(g[r, k], res["poses"]), (r, k)
                solves += res["n_solves"]
    finally:
        sv.close()
    assert abs(out["value"] - solves * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
