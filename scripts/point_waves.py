"""Stamp build only (-DMOVBA_CLOCK_STAMP): per-workgroup phase stamps of one back-substitution pass (k_point<true>), smuggled out through chi2."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
import numpy as np
from movba import synth, capi
w = synth.cfg(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
s = capi.Solver(); s.upload(w); s.run(); s.run(); r = s.download()
st = r["chi2"][20000:32768].view(np.uint64).reshape(-1, 8).astype(np.int64)
blk = st[st[:, 7] == 1]; dec = st[st[:, 7] == 2]
t0 = blk[:, 0].min()
T = (blk[:, :6] - t0) * 0.01
print("workgroups", len(blk), "first start 0, last start %.2f, last record stored %.2f us" % (T[:, 0].max(), T[:, 5].max()))
ph = np.diff(T, axis=1)
names = ["controller + edge ranges", "loads + staging", "back-substitution", "evaluation + scatter", "reduction + record"]
for k, nm in enumerate(names):
    print(f"  {nm:28s} mean {ph[:, k].mean():5.2f}  p90 {np.percentile(ph[:, k], 90):5.2f}  max {ph[:, k].max():5.2f} us")
print("  start percentiles:", " ".join("%d%%=%.2f" % (p, np.percentile(T[:, 0], p)) for p in (0, 50, 90, 100)))
print("  end percentiles:  ", " ".join("%d%%=%.2f" % (p, np.percentile(T[:, 5], p)) for p in (0, 10, 50, 90, 100)))
if len(dec):
    print("deciding wave: start %.2f end %.2f us (last record stored %.2f)" % ((dec[0, 0] - t0) * 0.01, (dec[0, 1] - t0) * 0.01, T[:, 5].max()))
