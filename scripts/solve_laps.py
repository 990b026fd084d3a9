"""The three phases of movba_lba_solve as the library's own clock sees them (MOVBA_TIME_SOLVE=1), beside the caller's clock
around the call: what the binding adds."""
import os, sys, time
os.environ["MOVBA_TIME_SOLVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
w = synth.cfg(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
s = capi.Solver()
s.prepare(w, pinned=True)
for i in range(8):
    t = time.perf_counter(); s.solve_prepared(pack=False); dt = time.perf_counter() - t
    print(f"caller's clock: {dt*1e3:.3f} ms", file=sys.stderr, flush=True)
