"""Structure + upload + solve + download wall times of whole movba_lba_solve calls (run on a GPU box)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
for name in sys.argv[1:] or ["cfg2", "cfg3"]:
    w = synth.cfg(name)
    s = capi.Solver(profile=False)
    for _ in range(3): s.solve(w)
    s.reset_profile()
    ts = []
    for _ in range(20):
        t = time.perf_counter(); s.solve(w); ts.append(time.perf_counter() - t)
    ts.sort(); pr = s.profile()
    print(f"{name}: whole solve call min {ts[0]*1e3:.3f} ms median {ts[10]*1e3:.3f} ms | per call: structure {pr['structure_ms']/20:.3f} upload {pr['upload_ms']/20:.3f} download {pr['download_ms']/20:.3f} ms")
