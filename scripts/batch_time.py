"""Window solves per second: n resident cfg3-shaped windows through movba_lba_run_batch against one after the other.
--short: three batched runs and nothing else (the counter passes of scripts/profile_patterns.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
import torch
from movba import capi, synth, shard
st = torch.cuda.Stream(device=0)
short = "--short" in sys.argv
for n in [int(v) for v in sys.argv[1:] if v != "--short"] or (1, 2, 4, 8, 16):
    ws = [synth.make_window(50, 10, 20000, shard.window_seed(i), run_lo=2, run_hi=10) for i in range(n)]
    solvers = [capi.Solver(device=0, stream=st.cuda_stream) for _ in range(n)]
    for s, w in zip(solvers, ws):
        s.upload(w)
    capi.run_batch(solvers); capi.run_batch(solvers)
    if short:
        capi.run_batch(solvers)
        for s in solvers: s.close()
        continue
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        capi.run_batch(solvers)
    tb = (time.perf_counter() - t0) / reps
    for s in solvers: s.run()
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in solvers: s.run()
    ts = (time.perf_counter() - t0) / reps
    lm = sum(s.download()["n_solves"] for s in solvers)
    print(f"n={n}: batch {1e3 * tb:.3f} ms ({n / tb:.0f} window solves/s, {lm / tb:.0f} LM it/s)  sequential {1e3 * ts:.3f} ms ({n / ts:.0f} window solves/s)  speed-up {ts / tb:.2f}x", flush=True)
    for s in solvers: s.close()
