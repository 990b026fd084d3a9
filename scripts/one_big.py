import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
w = synth.make_window(64, 8, 60000, seed=77, run_lo=2, run_hi=12)
s = capi.Solver(); s.upload(w); s.run(); r = s.download()
print("pcg", r["trace"]["pcg"].tolist())
