"""Replay ONE window of tests/dev/fuzz_parity.py (same generator stream: fuzz seed, iteration index) and print both traces."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle

rng = np.random.default_rng(int(sys.argv[1]))
target = int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else target          # windows first .. target are solved on ONE handle
w = None
hist = capi.Solver() if first < target else None
for it in range(target + 1):
    K = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 10, 12, 15, 16, 17, 20, 24, 31, 40, 50, 64, 79, 81, 95, 130]))
    F = int(rng.integers(1, 6))
    P = int(rng.choice([30, 80, 200, 600, 1500, 5000]))
    lo = int(rng.integers(2, 5)); hi = int(min(K + F, lo + rng.integers(0, 12)))
    stereo = float(rng.choice([0.0, 0.0, 0.5, 1.0]))
    seed = int(rng.integers(1, 10 ** 6))
    try:
        w = synth.make_window(K, F, P, seed=seed, run_lo=lo, run_hi=max(lo, hi), stereo_frac=stereo)
    except Exception:
        continue
    variant = int(rng.integers(0, 5))
    if variant == 0 and w.n_edges > 40:
        dead = rng.choice(w.n_points, size=max(1, w.n_points // 25), replace=False)
        keep = ~np.isin(w.edge_point, dead)
        kf_alive = np.bincount(w.edge_pose[keep], minlength=w.n_poses) > 0
        if kf_alive.all():
            w.edge_pose, w.edge_point, w.obs, w.inv_sigma2 = w.edge_pose[keep], w.edge_point[keep], w.obs[keep], w.inv_sigma2[keep]
            if getattr(w, "obs_right", None) is not None: w.obs_right = w.obs_right[keep]
    elif variant == 1:
        pm = rng.permutation(w.n_edges)
        w.edge_pose, w.edge_point, w.obs, w.inv_sigma2 = w.edge_pose[pm], w.edge_point[pm], w.obs[pm], w.inv_sigma2[pm]
        if getattr(w, "obs_right", None) is not None: w.obs_right = w.obs_right[pm]
    if hist is not None and first <= it < target:
        try:
            rh = hist.solve(w)
            print(f"  [{it}] history: K={K} F={F} P={P} stereo {stereo} variant {variant} E={w.n_edges} direct_from {rh['direct_from']} status {rh['status']}")
        except capi.MovbaError as e:
            print(f"  [{it}] history: refused {e}")
print(f"K={K} F={F} P={P} run {lo}-{hi} stereo {stereo} seed {seed} variant {variant} E={w.n_edges}")
s = hist if hist is not None else capi.Solver()
ro, rg = oracle.solve(w), s.solve(w)
np.set_printoptions(linewidth=200, precision=6)
for k in ("lam", "f0", "f1", "rho", "accept"):
    print(k, "oracle", ro["trace"][k]); print(k, "gpu   ", rg["trace"][k])
print("pcg", rg["trace"]["pcg"], "direct_from", rg["direct_from"], "n_direct", rg["n_direct"], "chol_fail", rg["n_chol_fail"], "giveups", rg["n_pcg_giveups"])
print("dq", np.abs(ro["poses"][:, :4] - rg["poses"][:, :4]).max(), "dt", np.abs(ro["poses"][:, 4:] - rg["poses"][:, 4:]).max())
