"""Replay ONE window of tests/dev/fuzz_parity.py (same generator stream: fuzz seed, iteration index) and print both traces."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_gen

rng = np.random.default_rng(int(sys.argv[1]))
target = int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else target          # windows first .. target are solved on ONE handle
w = None
hist = capi.Solver() if first < target else None
for it in range(target + 1):
    wn, d = fuzz_gen.next_window(rng)
    if wn is None:
        continue
    w = wn
    K, F, P, lo, hi, stereo, seed, variant = d["K"], d["F"], d["P"], d["lo"], d["hi"], d["stereo"], d["seed"], d["variant"]
    if hist is not None and first <= it < target:
        try:
            rh = hist.solve(w)
            print(f"  [{it}] history: K={K} F={F} P={P} stereo {stereo} variant {variant} E={w.n_edges} direct_from {rh['direct_from']} status {rh['status']}")
        except capi.MovbaError as e:
            print(f"  [{it}] history: refused {e}")
print(f"K={K} F={F} P={P} run {lo}-{hi} stereo {stereo} seed {seed} variant {variant} E={w.n_edges}")
s = hist if hist is not None else capi.Solver(direct=os.environ.get("MOVBA_FUZZ_DIRECT") == "1")
ro, rg = oracle.solve(w), s.solve(w)
np.set_printoptions(linewidth=200, precision=6)
for k in ("lam", "f0", "f1", "rho", "accept"):
    print(k, "oracle", ro["trace"][k]); print(k, "gpu   ", rg["trace"][k])
print("pcg", rg["trace"]["pcg"], "direct_from", rg["direct_from"], "n_direct", rg["n_direct"], "chol_fail", rg["n_chol_fail"], "giveups", rg["n_pcg_giveups"])
print("dq", np.abs(ro["poses"][:, :4] - rg["poses"][:, :4]).max(), "dt", np.abs(ro["poses"][:, 4:] - rg["poses"][:, 4:]).max(), "pt", np.abs(ro["points"] - rg["points"]).max())
per_kf = np.bincount(w.edge_pose, minlength=w.n_poses)[w.pose_fixed == 0]
print("edges per free keyframe: min", per_kf.min(), "median", int(np.median(per_kf)))
