"""Time one library variant (MOVBA_LIB) on cfg3: total solve and per-kernel event times."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
w = synth.cfg(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
s = capi.Solver()
s.upload(w)
ts = []
for _ in range(30):
    t = time.perf_counter(); s.run(); ts.append(time.perf_counter() - t)
r = s.download(); ts.sort()
s2 = capi.Solver(profile=True); s2.upload(w)
for _ in range(3): s2.run()
s2.reset_profile()
for _ in range(5): s2.run()
pr = s2.profile()["kernels"]
print(os.path.basename(os.environ.get("MOVBA_LIB", "default")), f"min {ts[0]*1e3:.3f} ms med {ts[15]*1e3:.3f} ms pcg {r['pcg_iters']}  " +
      "  ".join(f"{k.split('(')[0]} {1e3*v['ms']/max(v['launches'],1):.1f}us" for k, v in pr.items() if v['launches']))
