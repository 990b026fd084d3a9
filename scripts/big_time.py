"""Per-kernel event times on a larger window: K F P run_hi (default 64 8 60000 12)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
K, F, P, hi = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (64, 8, 60000, 12)
w = synth.make_window(K, F, P, seed=77, run_lo=2, run_hi=hi)
s2 = capi.Solver(profile=True); s2.upload(w)
for _ in range(2): s2.run()
s2.reset_profile()
for _ in range(5): s2.run()
r = s2.download(); pr = s2.profile()["kernels"]
print(f"K={K} F={F} P={P} E={w.n_edges} pcg {r['pcg_iters']}  " + "  ".join(f"{k.split('(')[0]} {1e3*v['ms']/max(v['launches'],1):.1f}us x{v['launches']//5}" for k, v in pr.items() if v['launches']))
