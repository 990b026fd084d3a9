"""Aggregate throughput of W independent windows solved concurrently on ONE GPU (one handle + stream + host thread each)."""
import sys, time, os, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi, shard

def run(W, reps=20):
    ws = [synth.make_window(50, 10, 20000, 1003 if k == 0 else shard.window_seed(k), run_lo=2, run_hi=10) for k in range(W)]
    solvers = [capi.Solver() for _ in range(W)]
    for s, w in zip(solvers, ws): s.upload(w)
    for s in solvers: s.run()
    bar = threading.Barrier(W + 1)
    def work(s):
        bar.wait()
        for _ in range(reps): s.run()
        bar.wait()
    th = [threading.Thread(target=work, args=(s,)) for s in solvers]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    its = sum(s.download()["n_solves"] for s in solvers) * reps
    print(f"W={W:2d}: {W*reps/dt:8.1f} window solves/s  {its/dt:9.1f} LM it/s  ({dt/reps*1e3:.3f} ms per round of {W})", flush=True)

for W in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]:
    run(W)
