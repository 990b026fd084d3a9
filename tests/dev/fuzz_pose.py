"""Randomised GPU-vs-oracle sweep of PoseOptimization (run on a GPU box): match counts around the LDS staging limit, outlier
fractions, thresholds, with and without the P3P hypothesis stage."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
s = capi.Solver()
bad = 0
worst = 0.0
for it in range(n_cases):
    n = int(rng.choice([4, 5, 9, 30, 64, 65, 200, 500, 511, 513, 1200, 2900, 3100, 4000]))
    of = float(rng.choice([0.0, 0.1, 0.3, 0.5]))
    hub = float(rng.choice([5.0, 8.0])); gate = hub * hub
    seed = int(rng.integers(1, 10 ** 6))
    ransac = bool(rng.integers(0, 2)) and n >= 30
    f = synth.make_frame(n=n, seed=seed, outlier_frac=of)
    if ransac:
        rs = int(rng.integers(1, 1000))
        samples = capi.ransac_samples(n, 50, rs)
        o_r = oracle.pose_ransac(f["Xw"], f["obs"], f["pose0"], f["cam"], gate, samples)
        o = oracle.pose_opt(f["Xw"], f["obs"], o_r["pose"], f["cam"], hub, gate)
        r = s.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], hub, gate, ransac_iters=50, ransac_seed=rs)
        ok = r["ransac_inliers"] == o_r["n_inliers"] and np.abs(r["ransac_pose"] - o_r["pose"]).max() < 1e-7
    else:
        o = oracle.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], hub, gate)
        r = s.pose_opt(f["Xw"], f["obs"], f["pose0"], f["cam"], hub, gate)
        ok = True
    dp = np.abs(r["pose"] - o["pose"]).max()
    mism = r["outlier"] != o["outlier"]
    # (a pose from a handful of matches is weakly constrained: rounding differences are amplified by the condition number; those
    #  frames are held to SURVEY 8(d)'s float32-map tolerance, like the weakly constrained windows of fuzz_parity.py)
    tol = 1e-6 if n < 12 else 1e-8
    ok = ok and dp < tol and (np.abs(o["chi2"][mism] - gate) <= 1e-6).all() and (mism.any() or r["n_inliers"] == o["n_inliers"])
    worst = max(worst, dp)
    if not ok:
        bad += 1
        print(f"[{it}] MISMATCH n={n} outliers {of} hub {hub} seed {seed} ransac {ransac}: dpose {dp:.2e} inliers {r['n_inliers']}/{o['n_inliers']} flag mismatches {int(mism.sum())}", flush=True)
print(f"{n_cases} frames, {bad} mismatches; worst pose difference {worst:.2e}")
sys.exit(1 if bad else 0)
