"""Stamp build only: per-wave start/end of k_schur(trial 3), smuggled out through the chi2 array."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
import numpy as np
from movba import synth, capi
w = synth.cfg("cfg3")
s = capi.Solver(); s.upload(w); s.run(); s.run(); r = s.download()
nw = int(sys.argv[1])
st = r["chi2"][:8192].view(np.uint64).reshape(-1, 2)[:nw].astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
a, b = (st[:, 0] - t0) * 0.01, (st[:, 1] - t0) * 0.01
print("abs first start", t0, "abs last end", st[:, 1].max())
print("waves", len(st), "start max %.2f us  end max %.2f us  dur mean %.2f max %.2f" % (a.max(), b.max(), (b - a).mean(), (b - a).max()))
order = np.argsort(-(b))[:12]
for i in order: print("  slot %4d (wg %3d wave %d) start %.2f end %.2f" % (i, i // int(sys.argv[2]), i % int(sys.argv[2]), a[i], b[i]))
h, _ = np.histogram(b, bins=10, range=(0, b.max())); print("end histogram", h.tolist())
