"""Task-by-task timeline of one launch of the one-launch direct solver (MOVBA_DENSE_STAMPS=1; run on a GPU box)."""
import os, sys
os.environ["MOVBA_DENSE_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
from movba import capi, synth
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w = synth.cfg(name) if name in ("cfg2", "cfg3", "small") else synth.make_window(150, 6, 6000, 9, run_lo=2, run_hi=10)
s = capi.Solver(pcg_max_iters=1)
r = s.solve(w)
print("solves", r["n_solves"], "direct", r["n_direct"], "chol_fail", r["n_chol_fail"])
s.close()
