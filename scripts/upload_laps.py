"""Host-side laps of movba_lba_upload (MOVBA_TIME_UPLOAD=1 prints them to stderr) for cfg3, after warm-up."""
import os, sys
os.environ["MOVBA_TIME_UPLOAD"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
w = synth.cfg(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "cfg3")
s = capi.Solver()
s.prepare(w, pinned="--staged" not in sys.argv)
for i in range(4):
    print(f"--- solve {i}", file=sys.stderr, flush=True)
    s.solve_prepared(pack=False)
