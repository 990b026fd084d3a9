import sys, os, time
ROOT = "/root/repo" if not os.environ.get("GRAFT_REPO_ROOT") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import capi, synth
for name, w in (("cfg2", synth.cfg("cfg2")), ("small", synth.cfg("small")), ("20kf x 5000", synth.make_window(20, 4, 5000, seed=11, run_lo=2, run_hi=8)), ("cfg3", synth.cfg("cfg3"))):
    for mode in ("default", "host_grouping", "host_structure"):
        s = capi.Solver(hooks=True)
        if mode != "default": s.hook(mode, 1)
        s.prepare(w, pinned=True)
        for _ in range(5): s.solve_prepared(pack=False)
        ts = []
        for _ in range(40):
            t = time.perf_counter(); s.solve_prepared(pack=False); ts.append(time.perf_counter() - t)
        ts.sort()
        print(f"{name:12s} E={w.n_edges:6d} {mode:15s} min {ts[0]*1e3:.3f} median {ts[20]*1e3:.3f} ms", flush=True)
        s.close()
