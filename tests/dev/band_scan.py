"""Where the single-workgroup banded factorisation overtakes the PCG: resident solve time over windows of growing size."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
for K, hi in ((8, 6), (12, 8), (16, 10), (20, 10), (24, 10), (28, 10), (32, 10), (40, 10), (16, 16), (24, 24)):
    w = synth.make_window(K, 3, 400 * K, 900 + K + hi, run_lo=2, run_hi=hi)
    out = []
    for kw in ({"solver": 3}, {"solver": 2}):
        s = capi.Solver(**kw)
        s.upload(w)
        for _ in range(3): s.run()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); s.run(); ts.append(time.perf_counter() - t0)
        r = s.download()
        out.append((1e3 * sorted(ts)[7], r["n_band"], r["n_direct"], int(np.sum(np.maximum(r["trace"]["pcg"], 0)))))
        s.close()
    print(f"K={K:3d} tracks 2-{hi:2d} E={w.n_edges:6d}: no band {out[0][0]:.3f} ms (direct trials {out[0][2]}, cg {out[0][3]})   band {out[1][0]:.3f} ms (band trials {out[1][1]})", flush=True)
