"""Randomised GPU-vs-oracle parity sweep over window shapes (run on a GPU box): free/fixed keyframe counts around every
code-path boundary (one keyframe per wave, two rows per aggregate, VGPR overflow, generic PCG), track lengths, stereo,
unobserved map points, edges that do not arrive grouped by map point, intrinsics by keyframe.
    python tests/dev/fuzz_parity.py <seed> <windows> [direct|band|pcg]  ("direct": every window on the one-launch direct solver;
                                                                         "band": on the banded factorisation wherever its band fits one CU's LDS)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_gen
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_order_noise

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mode = sys.argv[3] if len(sys.argv) > 3 else "default"
s = (capi.Solver(direct=True) if mode == "direct" else capi.Solver(solver=2) if mode == "band" else
     capi.Solver(solver=3) if mode == "pcg" else capi.Solver())       # "pcg": never the banded factorisation (small windows reach the PCG's fresh coarse level)
worst = dict(dq=0.0, dt=0.0, pt=0.0, outl=0)
bad = 0
noisy = 0       # windows beyond their tolerance but inside three times the oracle's own spread over the reference's edge orders
for it in range(n):
    w, d = fuzz_gen.next_window(rng)
    if w is None:
        continue
    K, F, P, lo, hi, stereo, seed = d["K"], d["F"], d["P"], d["lo"], d["hi"], d["stereo"], d["seed"]
    # keyframes held by a handful of observations make the reduced system rank-deficient up to the LM damping: the PCG gives
    # up there and the direct solver takes over; they are held to SURVEY 8(d)'s float32-map tolerance (two exact solvers differ
    # by cond(S) * eps in the weak directions) and counted as mismatches like every other window when they exceed it
    per_kf = np.bincount(w.edge_pose, minlength=w.n_poses)[w.pose_fixed == 0]
    weak = per_kf.min() < 12 if len(per_kf) else True
    tol_q, tol_t, tol_p = (1e-6, 1e-6, 1e-4) if weak else (1e-8, 1e-8, 1e-6)
    # a free keyframe with fewer than three observations has no unique pose at all (six unknowns from < 6 equations): the
    # reduced matrix is singular but for the LM damping, two exact solvers land anywhere along the free directions
    # (seed 21, window 474: 130 keyframes x 200 points, one observation on some keyframes: both direct solvers 1e-5 m from the
    # oracle, costs equal to six digits).  Such windows are held to the cost trace instead of the poses.
    degenerate = len(per_kf) > 0 and per_kf.min() < 3
    if degenerate: tol_q, tol_t, tol_p = 1e-3, 1e-3, 1e-2
    ro = oracle.solve(w)
    try:
        rg = s.solve(w)
    except capi.MovbaError as e:
        print(f"[{it}] K={K} F={F} P={P} run {lo}-{hi} stereo {stereo}: GPU refused: {e}")
        continue
    fr = w.pose_fixed == 0
    dq = np.abs(ro['poses'][:, :4] - rg['poses'][:, :4]).max(); dt = np.abs(ro['poses'][:, 4:] - rg['poses'][:, 4:]).max()
    pt = np.abs(ro['points'] - rg['points']).max() if w.n_points else 0.0
    guard = np.abs(ro['chi2'] - w.chi2_gate) <= 1e-6
    outl = int(((ro['outlier'] != rg['outlier']) & ~guard).sum())
    # decisions taken on rounding noise (|F0 - F1| <= 1e-9 F0: the solve has converged to machine precision) are a guard band
    f0, f1 = ro['trace']['f0'], ro['trace']['f1']
    # ... and so are decisions at a cost 18 orders of magnitude under the initial one (a noise-free window solved to the last
    # bit: residuals of 1e-13 px, whose squares are rounding noise in absolute terms)
    nz = np.flatnonzero((np.abs(f0 - f1) <= 1e-9 * np.abs(f0)) | (f0 <= 1e-18 * f0[0]))
    k0 = int(nz[0]) if len(nz) else len(f0)
    same = np.array_equal(ro['trace']['accept'][:k0], rg['trace']['accept'][:k0]) and (k0 < len(f0) or ro['n_solves'] == rg['n_solves'])
    ok = dq < tol_q and dt < tol_t and pt < tol_p and outl == 0 and same
    if degenerate: ok = ok and np.allclose(ro['trace']['f1'][:k0], rg['trace']['f1'][:k0], rtol=1e-4)
    worst['dq'] = max(worst['dq'], dq); worst['dt'] = max(worst['dt'], dt); worst['pt'] = max(worst['pt'], pt); worst['outl'] += outl
    if ok and not weak and (dt > 1e-9 or dq > 1e-10):
        print(f"[{it}] close to tolerance: K={K} F={F} P={P} run {lo}-{hi} stereo {stereo} seed {seed}: dq {dq:.2e} dt {dt:.2e} pt {pt:.2e} pcg {rg['pcg_iters']} per trial {rg['trace']['pcg'].tolist()}", flush=True)
    if not ok and outl == 0 and same and not degenerate:
        # beyond the tolerance of its class: how far does the ORACLE move when a map point's edges are added in another order
        # (the reference's own run-to-run freedom, conftest.oracle_order_noise)?  Inside three times that spread the window
        # says nothing about the solver.
        nq, nt, npt = oracle_order_noise(oracle, w, n=4)
        if dq <= max(tol_q, 3 * nq) and dt <= max(tol_t, 3 * nt) and pt <= max(tol_p, 3 * npt):
            noisy += 1
            print(f"[{it}] ORDER-NOISE K={K} F={F} P={P} run {lo}-{hi} stereo {stereo} seed {seed}: dq {dq:.2e} dt {dt:.2e} pt {pt:.2e} against the oracle's own spread {nq:.2e} {nt:.2e} {npt:.2e} (band trials {rg['n_band']}, direct {rg['n_direct']})", flush=True)
            continue
    if it % 50 == 49:
        print(f"[{it}] ... {it + 1} windows, {bad} mismatches, {noisy} inside the oracle's edge-order spread so far", flush=True)
    if not ok:
        bad += 1
        print(f"[{it}] weak={weak} direct_from={rg['direct_from']} n_direct={rg['n_direct']} chol_fail={rg['n_chol_fail']}", flush=True)
        print(f"[{it}] MISMATCH K={K} F={F} P={P} run {lo}-{hi} stereo {stereo} seed {seed}: dq {dq:.2e} dt {dt:.2e} pt {pt:.2e} outl {outl} accept_same {same} solves {ro['n_solves']}/{rg['n_solves']} pcg {rg['pcg_iters']}", flush=True)
print(f"{n} windows, {bad} mismatches, {noisy} inside the oracle's own edge-order spread; worst dq {worst['dq']:.2e} dt {worst['dt']:.2e} pt {worst['pt']:.2e}")
sys.exit(1 if bad else 0)
