"""Wall time of the three phases of a solve call (upload / run / download) from the host's point of view."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
for name in sys.argv[1:] or ["cfg2", "cfg3"]:
    w = synth.cfg(name)
    s = capi.Solver()
    for _ in range(3): s.solve(w)
    acc = [[], [], [], []]
    for _ in range(20):
        t0 = time.perf_counter(); s.upload(w); t1 = time.perf_counter(); s.run(); t2 = time.perf_counter(); s.download(); t3 = time.perf_counter()
        s.solve(w); t4 = time.perf_counter()
        for a, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): a.append(v * 1e3)
    med = [sorted(a)[len(a) // 2] for a in acc]
    print(f"{name}: upload {med[0]:.3f}  run {med[1]:.3f}  download {med[2]:.3f}  | one solve() call {med[3]:.3f} ms (medians of 20)")
