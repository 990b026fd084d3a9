"""Soak of the whole-call path (run on a GPU box): one thread plays LocalMapping - movba_lba_solve calls over a rotation of
windows (device grouping + ingest upload, banded / PCG / direct reduced solvers, pinned and ordinary caller arrays) on ONE
handle -, a second thread plays Tracking - movba_pose_opt calls on a handle of its own, at the same time (SURVEY 8(b):
PoseOptimization runs concurrently with LocalBundleAdjustment).  Every call must return the bits its window returned the
first time.      python tests/dev/soak_calls.py [seconds]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import capi, synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
errs, counts = [], {"lba": 0, "lba_pinned": 0, "pose": 0}
stop = threading.Event()


def lba_thread():
    s = capi.Solver()
    ws = [("cfg3", synth.cfg("cfg3")), ("cfg2", synth.cfg("cfg2")), ("small", synth.cfg("small")),
          ("stereo", synth.make_window(12, 3, 1500, seed=57, run_lo=2, run_hi=7, stereo_frac=0.5)),
          ("hub", synth.pattern_cfg("hub")), ("30 keyframes", synth.make_window(30, 4, 5000, seed=3, run_lo=2, run_hi=8))]
    first, preps = {}, {}
    try:
        it = 0
        while not stop.is_set():
            name, w = ws[it % len(ws)]
            pinned = (it // len(ws)) % 2 == 1                 # every other round: the caller's arrays in the library's pinned memory
            if pinned:
                # (buffers built once per window and kept, as a C++ caller keeps them: capi's pinned blocks live until close())
                if name not in preps:
                    s.prepare(w, pinned=True); preps[name] = s._prep
                s._prep = preps[name]
                r = s.solve_prepared()
                counts["lba_pinned"] += 1
            else:
                r = s.solve(w)
                counts["lba"] += 1
            key = {k: r[k].copy() for k in ("poses", "points", "chi2", "outlier")}
            if name not in first:
                first[name] = key
            else:
                for k, v in key.items():
                    if not np.array_equal(v, first[name][k]):
                        errs.append(f"{name} (call {it}, pinned {pinned}): {k} differs from the window's first solve")
                        stop.set()
            if r["status"] != 0 or r.get("n_sync_timeouts", 0):
                errs.append(f"{name} (call {it}): status {r['status']}, waits given up {r.get('n_sync_timeouts')}"); stop.set()
            it += 1
    except Exception as e:                                   # noqa: BLE001 - a soak reports whatever went wrong
        errs.append(f"LBA thread: {e!r}"); stop.set()
    finally:
        t_c = time.time(); s.close(); print(f"(LBA handle closed in {time.time() - t_c:.2f} s, thread done {time.time() - t0:.0f} s in)", flush=True)


def pose_thread():
    s = capi.Solver()
    fs = [synth.make_frame(n=n, seed=sd) for n, sd in ((500, 1001), (1200, 7), (4000, 77))]
    first = {}
    try:
        it = 0
        while not stop.is_set():
            i = it % len(fs); f = fs[i]
            r = s.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], float(np.float32(np.sqrt(5.991))), 5.991,
                           ransac_iters=50 if it % 2 else 0, ransac_seed=7)
            key = (i, it % 2)
            if key not in first:
                first[key] = r["pose"].copy()
            elif not np.array_equal(r["pose"], first[key]):
                errs.append(f"pose frame {key} (call {it}): differs from its first solve"); stop.set()
            counts["pose"] += 1
            it += 1
    except Exception as e:                                   # noqa: BLE001
        errs.append(f"pose thread: {e!r}"); stop.set()
    finally:
        t_c = time.time(); s.close(); print(f"(pose handle closed in {time.time() - t_c:.2f} s, thread done {time.time() - t0:.0f} s in)", flush=True)


t0 = time.time()
ts = [threading.Thread(target=lba_thread), threading.Thread(target=pose_thread)]
for t in ts: t.start()
while time.time() - t0 < secs and not stop.is_set():
    time.sleep(10.0)
    print(f"[{time.time() - t0:5.0f} s] {counts['lba']} + {counts['lba_pinned']} (pinned) LBA calls, {counts['pose']} pose calls, {len(errs)} errors", flush=True)
stop.set()
t_stop = time.time()
for t in ts: t.join()
print(f"(threads joined {time.time() - t_stop:.1f} s after the stop)")
print(f"{counts['lba']} + {counts['lba_pinned']} (pinned) movba_lba_solve calls and {counts['pose']} movba_pose_opt calls in {time.time() - t0:.0f} s on two handles at once: "
      f"{len(errs)} errors")
for e in errs[:10]: print("  " + e)
sys.exit(1 if errs else 0)
