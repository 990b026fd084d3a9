"""The single-workgroup banded factorisation (solver = 2) against the oracle and against the default solver choice, with times."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle
names = sys.argv[1:] or ["small", "cfg2", "cfg3"]
for name in names:
    w = synth.cfg(name) if name not in ("shuffled", "revisit", "hub") else synth.pattern_cfg(name)
    o = oracle.solve(w)
    for label, kw in (("default", {}), ("band", dict(solver=2))):
        s = capi.Solver(**kw)
        r = s.solve(w)
        s.upload(w)
        for _ in range(3): s.run()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); s.run(); ts.append(time.perf_counter() - t0)
        r2 = s.download()
        dq = np.abs(r["poses"][:, :4] - o["poses"][:, :4]).max(); dt = np.abs(r["poses"][:, 4:] - o["poses"][:, 4:]).max(); dp = np.abs(r["points"] - o["points"]).max()
        same = np.array_equal(r["poses"], r2["poses"])
        print(f"{name:8s} {label:8s}: status {r['status']} n_band {r['n_band']} n_direct {r['n_direct']} solves {r['n_solves']} pcg {r['trace']['pcg'].tolist()}  "
              f"dq {dq:.2e} dt {dt:.2e} dpt {dp:.2e}  accept same {np.array_equal(r['trace']['accept'], o['trace']['accept'])}  rerun bits {same}  resident {1e3 * sorted(ts)[10]:.3f} ms", flush=True)
        s.close()
