"""GPU vs oracle on ONE window shape: per-trial traces side by side (run on a GPU box).
    python tests/dev/one_case.py K F P lo hi stereo seed"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle
K, F, P, lo, hi = (int(v) for v in sys.argv[1:6]); stereo = float(sys.argv[6]); seed = int(sys.argv[7])
w = synth.make_window(K, F, P, seed=seed, run_lo=lo, run_hi=max(lo, hi), stereo_frac=stereo)
ro = oracle.solve(w)
for name, s in (("default", capi.Solver()), ("forced-direct", capi.Solver(pcg_max_iters=1))):
    rg = s.solve(w)
    print(name, "solves", ro["n_solves"], rg["n_solves"], "direct_from", rg["direct_from"], "n_direct", rg["n_direct"], "chol_fail", rg["n_chol_fail"])
    n = min(ro["n_solves"], rg["n_solves"])
    for k in range(n):
        print(f"  [{k}] lam {ro['trace']['lam'][k]:.6e} {rg['trace']['lam'][k]:.6e}  f1 {ro['trace']['f1'][k]:.10e} {rg['trace']['f1'][k]:.10e}  rho {ro['trace']['rho'][k]:+.3e} {rg['trace']['rho'][k]:+.3e}  acc {ro['trace']['accept'][k]} {rg['trace']['accept'][k]}  pcg {rg['trace']['pcg'][k]}")
    print("  dpose", np.abs(ro["poses"] - rg["poses"]).max())
