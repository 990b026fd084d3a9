"""One resident cfg3 (or argv[1]) solve, for rocprofv3 traces."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
w = synth.cfg(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
s = capi.Solver()
s.upload(w)
for _ in range(3):
    s.run()
r = s.download()
print("solves", r["n_solves"], "pcg", r["trace"]["pcg"].tolist())
