"""A few solves on the direct path for rocprofv3 --kernel-trace --stats (kernel time per class of the dense solver)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
from movba import capi, synth
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w = synth.cfg("cfg3") if which == "cfg3" else synth.make_window(400, 8, 12000, 9, run_lo=2, run_hi=12)
s = capi.Solver(pcg_max_iters=1)
for _ in range(4):
    r = s.solve(w, max_iters=10 if which == "cfg3" else 3)
print(which, r["n_solves"], r["n_direct"])
