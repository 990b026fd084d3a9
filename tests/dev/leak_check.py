"""Create / solve / destroy handles repeatedly and watch free device memory (run on a GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
import torch, resource
from movba import synth, capi
ws = [synth.cfg("cfg2"), synth.cfg("small"), synth.make_window(30, 4, 5000, seed=3, run_lo=2, run_hi=8)]
free0 = None
for rep in range(40):
    s = capi.Solver()
    for w in ws: s.solve(w)
    s.prepare(ws[0], pinned=True); s.solve_prepared()        # pinned result arrays (movba_host_alloc / movba_host_free)
    f = synth.make_frame(n=600)
    s.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], 5.0, 25.0, ransac_iters=50, ransac_seed=2)
    s.close()
    free, total = torch.cuda.mem_get_info()
    if rep == 4: free0 = free
    if rep % 10 == 9: print(f"rep {rep}: free {free / 2**20:.1f} MiB")
print("leak per create/destroy cycle: %.3f MiB of device memory; host max RSS %.0f MiB" % ((free0 - free) / 2**20 / 35, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024))
