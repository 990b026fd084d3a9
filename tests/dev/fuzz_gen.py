"""The window stream of tests/dev/fuzz_parity.py and fuzz_one.py: one call draws the next window from the generator."""
import numpy as np
from movba import synth


def next_window(rng):
    """-> (window or None when the draw was degenerate, description dict)"""
    K = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 10, 12, 15, 16, 17, 20, 24, 31, 40, 50, 64, 79, 81, 95, 130]))
    F = int(rng.integers(1, 6))
    P = int(rng.choice([30, 80, 200, 600, 1500, 5000]))
    lo = int(rng.integers(2, 5)); hi = int(min(K + F, lo + rng.integers(0, 12)))
    stereo = float(rng.choice([0.0, 0.0, 0.5, 1.0]))
    seed = int(rng.integers(1, 10 ** 6))
    d = dict(K=K, F=F, P=P, lo=lo, hi=hi, stereo=stereo, seed=seed, variant=-1)
    try:
        w = synth.make_window(K, F, P, seed=seed, run_lo=lo, run_hi=max(lo, hi), stereo_frac=stereo)
    except Exception:          # degenerate generator input
        return None, d
    variant = int(rng.integers(0, 6))
    d["variant"] = variant
    if variant == 0 and w.n_edges > 40:                 # map points nobody observes: drop all edges of a few points
        dead = rng.choice(w.n_points, size=max(1, w.n_points // 25), replace=False)
        keep = ~np.isin(w.edge_point, dead)
        kf_alive = np.bincount(w.edge_pose[keep], minlength=w.n_poses) > 0
        if kf_alive.all():
            w.edge_pose, w.edge_point, w.obs, w.inv_sigma2 = w.edge_pose[keep], w.edge_point[keep], w.obs[keep], w.inv_sigma2[keep]
            if getattr(w, "obs_right", None) is not None: w.obs_right = w.obs_right[keep]
    elif variant == 1:                                  # edges not grouped by map point (host structure pass, permuted arrays)
        pm = rng.permutation(w.n_edges)
        w.edge_pose, w.edge_point, w.obs, w.inv_sigma2 = w.edge_pose[pm], w.edge_point[pm], w.obs[pm], w.inv_sigma2[pm]
        if getattr(w, "obs_right", None) is not None: w.obs_right = w.obs_right[pm]
    elif variant == 5:                                  # every keyframe with a camera (and baseline) of its own
        w = synth.mixed_cameras(w, seed=seed + 1)
    return w, d
