"""Host-side laps of movba_lba_upload (MOVBA_TIME_UPLOAD=1) and the phases of the whole call for windows beyond the device
structure pass (more than 80 free keyframes): where a large window's call time goes."""
import os, sys, time
os.environ["MOVBA_TIME_UPLOAD"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
cases = {"150kf-60k": lambda: synth.make_window(150, 10, 60000, 9, run_lo=2, run_hi=10),
         "150kf": lambda: synth.make_window(150, 6, 6000, 9, run_lo=2, run_hi=10),
         "400kf": lambda: synth.make_window(400, 8, 12000, 9, run_lo=2, run_hi=12)}
name = sys.argv[1] if len(sys.argv) > 1 else "150kf-60k"
w = cases[name]()
s = capi.Solver(profile=True)
s.prepare(w)
for i in range(3):
    print(f"--- solve {i}", file=sys.stderr, flush=True)
    t = time.perf_counter(); s.solve_prepared(pack=False); dt = time.perf_counter() - t
    print(f"call {1e3 * dt:.3f} ms", file=sys.stderr, flush=True)
p = s.profile()
print({k: round(v, 3) for k, v in p.items() if k.endswith("_ms")}, file=sys.stderr)
