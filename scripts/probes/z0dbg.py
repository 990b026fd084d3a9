import sys, os
sys.path.insert(0, "mov-slam_amd"); sys.path.insert(0, ".")
import numpy as np
from movba import synth, capi
from oracle import oracle
w = synth.cfg("small")
e0 = int(np.flatnonzero(w.edge_pose == 0)[0]); l0 = int(w.edge_point[e0])
w.points = w.points.copy(); w.points[l0, 2] = 0.0
s = capi.Solver()
r = s.solve(w, max_iters=3); o = oracle.solve(w, max_iters=3)
for k in ("accept", "f0", "f1", "rho", "lam"):
    print(k, r["trace"][k], o["trace"][k])
print(r["n_chol_fail"], r["n_band"], r["n_direct"])
