"""Repeated movba_lba_run on ONE upload with the one-launch direct solver (its flags carry the launch epoch and are never reset
between launches): every run returns the bits of the first solve (run on a GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import capi, synth
for w, kw in ((synth.make_window(150, 6, 6000, 9, run_lo=2, run_hi=10), {}), (synth.pattern_cfg("hub"), {}), (synth.cfg("cfg3"), dict(direct=True))):
    s = capi.Solver(**kw)
    ref = s.solve(w)
    s.upload(w)
    outs = []
    for _ in range(4):
        s.run(); outs.append(s.download())
    ok = all(np.array_equal(o["poses"], ref["poses"]) and np.array_equal(o["chi2"], ref["chi2"]) and o["n_sync_timeouts"] == 0 and o["n_direct"] == ref["n_direct"] for o in outs)
    print("free", w.n_free, "direct trials", ref["n_direct"], "repeat-identical", ok)
    assert ok
    s.close()
print("RERUN OK")
