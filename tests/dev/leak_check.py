"""Create / solve / destroy handles repeatedly and watch free device memory (run on a GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
import torch
from movba import synth, capi
ws = [synth.cfg("cfg2"), synth.cfg("small"), synth.make_window(30, 4, 5000, seed=3, run_lo=2, run_hi=8)]
free0 = None
for rep in range(40):
    s = capi.Solver()
    for w in ws: s.solve(w)
    s.close()
    free, total = torch.cuda.mem_get_info()
    if rep == 4: free0 = free
    if rep % 10 == 9: print(f"rep {rep}: free {free / 2**20:.1f} MiB")
print("leak per create/destroy cycle: %.3f MiB" % ((free0 - free) / 2**20 / 35))
