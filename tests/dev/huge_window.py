"""One window beyond the one-launch direct solver (more than 432 free keyframes: the multi-launch solver of dense_solve.hip,
the sort-based structure pass up to 1 024 keyframes, the host pass beyond) against the oracle, a few LM iterations."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle
for K, F, P in ((500, 8, 12000), (1100, 8, 20000)):
    w = synth.make_window(K, F, P, 9, run_lo=2, run_hi=12); w.max_iters = 2
    t = time.time(); ro = oracle.solve(w); tc = time.time() - t
    s = capi.Solver(); rg = s.solve(w)
    t = time.perf_counter(); rg = s.solve(w); tg = time.perf_counter() - t
    print(f"K={K} P={P} E={w.n_edges}: oracle {tc:.1f} s, gpu call {1e3 * tg:.1f} ms, solves {ro['n_solves']}/{rg['n_solves']} direct {rg['n_direct']} "
          f"dq {np.abs(ro['poses'][:, :4] - rg['poses'][:, :4]).max():.2e} dt {np.abs(ro['poses'][:, 4:] - rg['poses'][:, 4:]).max():.2e} "
          f"outl {int((ro['outlier'] != rg['outlier']).sum())}", flush=True)
    s.close()
