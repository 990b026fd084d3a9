"""Wall time of movba_pose_opt calls (cfg1: one Frame, 500 matches) against the single-threaded oracle."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
from movba import synth, capi
s = capi.Solver()
for n in (500, 1200):
    f = synth.make_frame(n=n)
    hub, gate = float(np.float32(np.sqrt(5.991))), 5.991
    for _ in range(5): r = s.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], hub, gate)
    ts = []
    for _ in range(50):
        t = time.perf_counter(); r = s.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], hub, gate); ts.append(time.perf_counter() - t)
    ts.sort()
    tr = []
    for _ in range(50):
        t = time.perf_counter(); rr = s.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], hub, gate, ransac_iters=50, ransac_seed=7); tr.append(time.perf_counter() - t)
    tr.sort()
    line = (f"n={n}: movba_pose_opt call min {ts[0]*1e3:.3f} ms median {ts[25]*1e3:.3f} ms, inliers {r['n_inliers']} ({r['lm_iters']} LM iterations)"
            f" | with the 50-hypothesis P3P stage: median {tr[25]*1e3:.3f} ms, inliers {rr['n_inliers']} ({rr['lm_iters']} LM iterations)")
    if "--oracle" in sys.argv:
        from oracle import oracle
        t = time.perf_counter()
        for _ in range(20): o = oracle.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], hub, gate)
        line += f" | oracle (1 thread) {(time.perf_counter()-t)/20*1e3:.3f} ms"
    print(line)
