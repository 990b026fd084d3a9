"""The windows of the randomised sweep on which round 4's banded factorisation left the oracle, on the default solver choice
and forced onto k_band: distances to the oracle's poses, which solver answered, resident time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle
W = [(16, 5, 30, 2, 7, 0.0, 372277), (81, 4, 30, 2, 4, 1.0, 28338), (12, 3, 80, 2, 2, 0.0, 982937),
     (40, 1, 1500, 3, 3, 0.0, 166442), (50, 2, 600, 2, 3, 0.0, 398504), (50, 2, 600, 2, 4, 0.0, 563415),
     (50, 2, 30, 2, 6, 0.0, 101), (40, 1, 60, 2, 4, 0.0, 102), (24, 2, 40, 3, 8, 0.0, 103), (16, 1, 25, 2, 5, 0.0, 104), (9, 1, 12, 2, 4, 0.0, 106)]
for (K, F, P, lo, hi, st, seed) in W:
    w = synth.make_window(K, F, P, seed=seed, run_lo=lo, run_hi=max(lo, hi), stereo_frac=st)
    o = oracle.solve(w)
    per_kf = np.bincount(w.edge_pose, minlength=w.n_poses)[w.pose_fixed == 0]
    for label, kw in (("default", {}), ("band", dict(solver=2)), ("direct", dict(solver=1))):
        s = capi.Solver(**kw)
        try:
            r = s.solve(w)
        except capi.MovbaError as e:
            print(f"K={K} seed {seed} {label}: refused {e}"); s.close(); continue
        dq = np.abs(r["poses"][:, :4] - o["poses"][:, :4]).max(); dt = np.abs(r["poses"][:, 4:] - o["poses"][:, 4:]).max(); dp = np.abs(r["points"] - o["points"]).max()
        print(f"K={K:3d} F={F} P={P:5d} run {lo}-{hi} st {st} seed {seed:7d} min obs/kf {per_kf.min():3d} {label:8s}: n_band {r['n_band']:2d} n_direct {r['n_direct']:2d} giveups {r['n_pcg_giveups']} chol_fail {r['n_chol_fail']} "
              f"solves {r['n_solves']}/{o['n_solves']} dq {dq:.2e} dt {dt:.2e} dpt {dp:.2e} accept same {np.array_equal(r['trace']['accept'], o['trace']['accept'])}", flush=True)
        s.close()
