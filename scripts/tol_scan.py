"""PCG stopping tolerance against iteration count, resident solve time and distance from the 1e-10 result (cfg3 and patterns)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
import numpy as np
from movba import synth, capi
names = sys.argv[1:] or ["cfg3", "cfg2"]
for name in names:
    w = synth.cfg(name)
    ref = None
    for tol in (1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
        s = capi.Solver(pcg_rel_tol=tol)
        s.upload(w)
        for _ in range(5): s.run()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); s.run(); ts.append(time.perf_counter() - t0)
        r = s.download()
        poses = r["poses"].copy(); pts = r["points"].copy(); cost = r["trace"]["f1"].copy()
        if ref is None: ref = (poses, pts, cost)
        dp = np.abs(poses - ref[0]).max(); dx = np.abs(pts - ref[1]).max()
        dc = (np.abs(cost - ref[2]) / np.abs(ref[2])).max() if cost is not None and cost.shape == ref[2].shape else float("nan")
        print(f"{name} tol {tol:.0e}: pcg iterations {int(np.sum(r['trace']['pcg']))} solves {r['n_solves']}  resident {1e3 * sorted(ts)[len(ts)//2]:.3f} ms  "
              f"max |dpose| {dp:.2e}  max |dpoint| {dx:.2e}  max rel dcost {dc:.2e}", flush=True)
        s.close()
