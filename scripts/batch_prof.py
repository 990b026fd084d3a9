"""A few batched runs of n cfg3-shaped windows for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
import torch
from movba import capi, synth, shard
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
st = torch.cuda.Stream(device=0)
ws = [synth.make_window(50, 10, 20000, shard.window_seed(i), run_lo=2, run_hi=10) for i in range(n)]
solvers = [capi.Solver(device=0, stream=st.cuda_stream) for _ in range(n)]
for s, w in zip(solvers, ws):
    s.upload(w)
for _ in range(5):
    capi.run_batch(solvers)
print("done", n)
