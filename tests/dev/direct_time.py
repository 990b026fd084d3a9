"""Time per window solve of the direct (dense Cholesky) path beside the on-chip PCG and the oracle (run on a GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import capi, synth
from oracle import oracle


def timed(s, w, n=5, **kw):
    s.solve(w, **kw)
    t = time.perf_counter()
    for _ in range(n):
        r = s.solve(w, **kw)
    return 1e3 * (time.perf_counter() - t) / n, r


cases = [("cfg3", synth.cfg("cfg3"), {}), ("80kf", synth.make_window(80, 10, 40000, 5, run_lo=2, run_hi=10), {}),
         ("150kf", synth.make_window(150, 6, 6000, 9, run_lo=2, run_hi=10), {}),
         ("150kf-60k", synth.make_window(150, 10, 60000, 9, run_lo=2, run_hi=10), {}),
         ("400kf", synth.make_window(400, 8, 12000, 9, run_lo=2, run_hi=12), dict(max_iters=3))]
pcg = capi.Solver(profile=True); direct = capi.Solver(direct=True, profile=True)
for name, w, kw in cases:
    t0 = time.perf_counter(); o = oracle.solve(w, **kw); to = 1e3 * (time.perf_counter() - t0)
    pcg.reset_profile(); direct.reset_profile()
    tp, rp = timed(pcg, w, **kw)
    td, rd = timed(direct, w, **kw)
    kp, kd = pcg.profile()["kernels"], direct.profile()["kernels"]
    print(f"{name}: K={w.n_free} E={w.n_edges} solves {rp['n_solves']}  default {tp:.3f} ms (direct trials {rp['n_direct']}, solve-kernels "
          f"{kp['k_pcg']['ms'] / 6:.3f} ms)  forced-direct {td:.3f} ms (solve-kernels {kd['k_pcg']['ms'] / 6:.3f} ms)  oracle {to:.1f} ms  "
          f"dpose default {np.abs(rp['poses'] - o['poses']).max():.1e} direct {np.abs(rd['poses'] - o['poses']).max():.1e}", flush=True)
