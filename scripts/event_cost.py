"""What the HIP events bench.py puts around the dominant kernel's launches cost the timed steps (whole movba_lba_solve calls, cfg3)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
w = synth.cfg("cfg3")
s = capi.Solver()
s.prepare(w, pinned=True)
for _ in range(5): s.solve_prepared(pack=False)
def run(n_evt, steps=20, mask=2):
    s.set_profile_mask(mask if n_evt > 0 else 0); s.reset_profile()
    t0 = time.perf_counter()
    for i in range(steps):
        if i == n_evt: s.set_profile_mask(0)
        s.solve_prepared(pack=False)
    return 1e3 * (time.perf_counter() - t0) / steps
for rep in range(3):
    print("events on 0 / 5 / 20 of 20 steps: %.4f  %.4f  %.4f ms per step" % (run(0), run(5), run(20)), flush=True)
