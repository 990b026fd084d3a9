"""Resident-window solve time (no profiling events) for a few option sets (run on a GPU box)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w = synth.cfg(name)
for ra in (2, 3, 4, 6):
    s = capi.Solver(run_ahead=ra)
    s.upload(w)
    ts = []
    for _ in range(30):
        t = time.perf_counter(); s.run(); ts.append(time.perf_counter() - t)
    r = s.download()
    ts.sort()
    print(f"{name} run_ahead {ra}: min {ts[0]*1e3:.3f} ms median {ts[len(ts)//2]*1e3:.3f} ms  solves {r['n_solves']} pcg {r['pcg_iters']}")
