"""Stamp build only (-DMOVBA_CLOCK_STAMP): per-wave phase stamps of one k_schur launch, smuggled out through chi2."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
import numpy as np
from movba import synth, capi
w = synth.cfg("cfg3")
s = capi.Solver(); s.upload(w); s.run(); s.run(); r = s.download()
st = r["chi2"][:20000].view(np.uint64).reshape(-1, 8).astype(np.int64)
st = st[st[:, 7] == 1]
t0 = st[:, 0].min()
T = (st[:, :5] - t0) * 0.01
n = st[:, 5]; dg = st[:, 6] & 1
hw = (st[:, 6] >> 8) & 0xffffffff; xcc = (st[:, 6] >> 40) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
print("waves", len(st), "first start 0, last start %.2f, last end %.2f us" % (T[:, 0].max(), T[:, 4].max()))
ph = np.diff(T, axis=1)
for name, m in (("diag", dg == 1), ("offdiag busy", (dg == 0) & (n > 0)), ("idle", n == 0)):
    if m.any():
        print(f"{name:13s} n={m.sum():4d} entries/wave mean {n[m].mean():6.1f} max {n[m].max():4d} | prologue {ph[m,0].mean():5.2f} loop {ph[m,1].mean():5.2f} (max {ph[m,1].max():5.2f}) reduce {ph[m,2].mean():5.2f} store {ph[m,3].mean():5.2f} | start {T[m,0].mean():5.2f} end mean {T[m,4].mean():5.2f} max {T[m,4].max():5.2f}")
order = np.argsort(-T[:, 4])[:8]
for i in order: print("  late wave: entries %4d diag %d start %.2f prologue %.2f loop %.2f reduce %.2f store %.2f end %.2f" % (n[i], dg[i], T[i, 0], ph[i, 0], ph[i, 1], ph[i, 2], ph[i, 3], T[i, 4]))
ws = np.sort(T[::4, 0])      # one per workgroup (wave 0)
print("workgroup start percentiles (us):", " ".join("%d%%=%.2f" % (p, np.percentile(ws, p)) for p in (0, 10, 25, 50, 75, 80, 85, 90, 95, 100)))
print("WG end percentiles (us):", " ".join("%d%%=%.2f" % (p, np.percentile(T[:, 4], p)) for p in (0, 10, 25, 50, 75, 90, 100)))

for t in (1.0, 4.0, 7.0, 10.0, 14.0, 18.0):
    live = (T[:, 0] <= t) & (T[:, 4] > t)
    cuid = xcc[live] * 1000 + se[live] * 100 + sh[live] * 16 + cu[live]
    print("t=%5.1f us: %4d waves live on %3d distinct (xcc,se,sh,cu); per-xcc %s" % (t, live.sum(), len(np.unique(cuid)), np.bincount(xcc[live], minlength=8).tolist()))
allcu = xcc * 1000 + se * 100 + sh * 16 + cu
print("distinct CUs used overall", len(np.unique(allcu)), "se values", np.unique(se).tolist(), "sh", np.unique(sh).tolist(), "cu", np.unique(cu).tolist())
