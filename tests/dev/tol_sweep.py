"""PCG stopping tolerance vs parity and CG iteration count (run on a GPU box)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle

for name in sys.argv[1:] or ["cfg2", "cfg3"]:
    w = synth.cfg(name)
    ro = oracle.solve(w)
    for tol in (1e-12, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4):
        s = capi.Solver(pcg_rel_tol=tol)
        rg = s.solve(w)
        s.upload(w); ts = []
        for _ in range(5):
            t = time.time(); s.run(); ts.append(time.time() - t)
        print(f"{name} tol {tol:.0e}: pcg {rg['pcg_iters']:4d} run {min(ts)*1e3:.3f} ms solves {rg['n_solves']}/{ro['n_solves']}"
              f" dq {np.abs(ro['poses'][:, :4] - rg['poses'][:, :4]).max():.2e} dt {np.abs(ro['poses'][:, 4:] - rg['poses'][:, 4:]).max():.2e}"
              f" pt {np.abs(ro['points'] - rg['points']).max():.2e} chi2 {np.abs(ro['chi2'] - rg['chi2']).max():.2e}"
              f" outl {int((ro['outlier'] != rg['outlier']).sum())} accept_same {np.array_equal(ro['trace']['accept'], rg['trace']['accept'])}")
        del s
