#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run from the repo root).

The reference has no tests or golden vectors for its optimizer and cannot
be run here, so these vectors are NOT reference outputs: they come from an independent
numpy restatement of the same algorithm that shares no code with oracle/lba_oracle.c —
poses as 4x4 matrices updated with scipy.linalg.expm of the twist, the FULL (6K+3P)
normal equations solved with numpy.linalg.solve (no Schur complement, no
back-substitution), Jacobians from the chain rule on matrices.  Agreement between the
two pins the oracle's Schur/back-substitution algebra, SE3 exponential and quaternion
handling against a second derivation; the LM schedule itself is g2o's published one
 in both.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
from scipy.linalg import expm

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth  # noqa: E402


def T_from_qt(qt):
    T = np.eye(4)
    T[:3, :3] = synth.R_from_quat(qt[:4] / np.linalg.norm(qt[:4]))
    T[:3, 3] = qt[4:]
    return T


def qt_from_T(T):
    return np.concatenate([synth.quat_from_R(T[:3, :3]), T[:3, 3]])


def twist(d):
    w, v = d[:3], d[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = v
    return M


def huber(c, delta):
    d2 = delta * delta
    if delta <= 0 or c <= d2:
        return c, 1.0
    s = np.sqrt(c)
    return 2 * s * delta - d2, delta / s


def errors(Ts, X, w):
    """(E,3) residuals; third component u_r - (u - bf/z) for stereo edges (g2o EdgeStereoSE3ProjectXYZ), else 0."""
    fx, fy, cx, cy, bf = edge_cameras(w)
    Xc = np.einsum("eij,ej->ei", Ts[w.edge_pose][:, :3, :3], X[w.edge_point]) + Ts[w.edge_pose][:, :3, 3]
    u = fx * Xc[:, 0] / Xc[:, 2] + cx
    e = np.zeros((w.n_edges, 3))
    e[:, 0] = w.obs[:, 0] - u
    e[:, 1] = w.obs[:, 1] - (fy * Xc[:, 1] / Xc[:, 2] + cy)
    if w.obs_right is not None:
        st = w.obs_right >= 0
        e[st, 2] = w.obs_right[st] - (u[st] - bf[st] / Xc[st, 2])
    return e, Xc


def edge_cameras(w):
    """per-edge fx, fy, cx, cy, bf: the camera of the edge's keyframe (e->pCamera = pKFi->mpCamera, src/Optimizer.cc:664;
    e->fx .. e->bf from pKFi, :690-695) when the window carries cameras by keyframe, the window's one camera otherwise"""
    E = w.n_edges
    if getattr(w, "cam_kf", None) is not None:
        ck = np.asarray(w.cam_kf)[w.edge_pose]
        fx, fy, cx, cy = ck[:, 0], ck[:, 1], ck[:, 2], ck[:, 3]
    else:
        fx, fy, cx, cy = (np.full(E, v) for v in w.cam)
    bf = np.asarray(w.bf_kf)[w.edge_pose] if getattr(w, "bf_kf", None) is not None else np.full(E, w.bf)
    return fx, fy, cx, cy, bf


def robust_cost(e, w):
    c = w.inv_sigma2 * (e ** 2).sum(1)
    return sum(huber(ci, w.huber_delta)[0] for ci in c)


def lm_dense(w, max_iters=10):
    """g2o Levenberg on the full un-Schur'd system."""
    NP, P, E = w.n_poses, w.n_points, w.n_edges
    efx, efy, ecx, ecy, ebf = edge_cameras(w)
    Ts = np.stack([T_from_qt(q) for q in w.poses])
    X = w.points.copy()
    free = np.flatnonzero(w.pose_fixed == 0)
    hidx = -np.ones(NP, int); hidx[free] = np.arange(len(free))
    n = 6 * len(free) + 3 * P
    off = 6 * len(free)
    trace = dict(lam=[], f0=[], f1=[], rho=[], accept=[])
    lam, ni, ok = 0.0, 2.0, True
    e_stored, _ = errors(Ts, X, w)
    iters = 0
    for it in range(max_iters):
        if not ok:
            break
        e, Xc = errors(Ts, X, w)
        e_stored = e
        F0 = robust_cost(e, w)
        H = np.zeros((n, n)); b = np.zeros(n)
        for k in range(E):
            ip, l = w.edge_pose[k], w.edge_point[k]
            x, y, z = Xc[k]
            fx, fy = efx[k], efy[k]
            Jpi = np.array([[fx / z, 0, -fx * x / z ** 2], [0, fy / z, -fy * y / z ** 2], [0, 0, 0]])
            if w.obs_right is not None and w.obs_right[k] >= 0:
                Jpi[2] = [fx / z, 0, -fx * x / z ** 2 + ebf[k] / z ** 2]      # d(u - bf/z)/dXc
            A = -Jpi @ Ts[ip][:3, :3]
            skew = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
            B = -Jpi @ np.hstack([-skew, np.eye(3)])
            c = w.inv_sigma2[k] * (e[k] @ e[k])
            wgt = huber(c, w.huber_delta)[1]
            om = wgt * w.inv_sigma2[k]
            r = -om * e[k]
            sl = slice(off + 3 * l, off + 3 * l + 3)
            H[sl, sl] += om * A.T @ A; b[sl] += A.T @ r
            if hidx[ip] >= 0:
                sp = slice(6 * hidx[ip], 6 * hidx[ip] + 6)
                H[sp, sp] += om * B.T @ B; b[sp] += B.T @ r
                H[sp, sl] += om * B.T @ A; H[sl, sp] += om * A.T @ B
        if it == 0:
            lam = 1e-5 * np.abs(np.diag(H)).max(); ni = 2.0
        rho, qmax = 0.0, 0
        while True:
            Ts_bk, X_bk = Ts.copy(), X.copy()
            dx = np.linalg.solve(H + lam * np.eye(n), b)
            for i in free:
                Ts[i] = expm(twist(dx[6 * hidx[i]:6 * hidx[i] + 6])) @ Ts[i]
            X = X + dx[off:].reshape(P, 3)
            e1, _ = errors(Ts, X, w)
            e_stored = e1
            F1 = robust_cost(e1, w)
            scale = float(dx @ (lam * dx + b)) + 1e-3
            rho = (F0 - F1) / scale
            trace["lam"].append(lam); trace["f0"].append(F0); trace["f1"].append(F1); trace["rho"].append(rho)
            if rho > 0 and np.isfinite(F1):
                alpha = min(1 - (2 * rho - 1) ** 3, 2 / 3)
                lam *= max(1 / 3, alpha); ni = 2.0; F0 = F1
                trace["accept"].append(1)
            else:
                lam *= ni; ni *= 2; Ts, X = Ts_bk, X_bk
                trace["accept"].append(0)
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        iters = it + 1
        if qmax == 10 or rho == 0:
            ok = False
    chi2 = w.inv_sigma2 * (e_stored ** 2).sum(1)
    _, Xc = errors(Ts, X, w)
    outlier = ((chi2 > w.chi2_gate) | ~(Xc[:, 2] > 0)).astype(np.uint8)
    return dict(poses=np.stack([qt_from_T(T) for T in Ts]), points=X, chi2=chi2, outlier=outlier,
                iters=iters, lam=lam, **{"tr_" + k: np.array(v) for k, v in trace.items()})


def save(name, w, out):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz")
    np.savez_compressed(
        path, in_poses=w.poses, in_pose_fixed=w.pose_fixed, in_points=w.points, in_edge_pose=w.edge_pose,
        in_edge_point=w.edge_point, in_obs=w.obs, in_inv_sigma2=w.inv_sigma2, in_cam=np.array(w.cam),
        in_huber_delta=w.huber_delta, in_chi2_gate=w.chi2_gate, in_max_iters=w.max_iters,
        in_obs_right=(w.obs_right if w.obs_right is not None else np.zeros(0)), in_bf=w.bf,
        **({"in_cam_kf": w.cam_kf} if getattr(w, "cam_kf", None) is not None else {}),
        **({"in_bf_kf": w.bf_kf} if getattr(w, "bf_kf", None) is not None else {}),
        **{"out_" + k: v for k, v in out.items()})
    print(name, w.meta, "iters", out["iters"], "trials", len(out["tr_lam"]),
          "accept", out["tr_accept"].tolist(), "outliers", int(out["outlier"].sum()))


def main():
    # 3 KF x 20 points, 10 KF x 200 points (SURVEY.md §8c v)
    w = synth.cfg("tiny"); save("lba_tiny", w, lm_dense(w))
    w = synth.cfg("small"); save("lba_small", w, lm_dense(w))
    # a hard start (large noise) that exercises rejected trials / lambda growth
    w = synth.make_window(4, 2, 60, seed=23, run_lo=2, run_hi=5, rot_sigma_deg=6.0, trans_sigma=0.6,
                          point_sigma=1.5, outlier_frac=0.15)
    save("lba_hard", w, lm_dense(w))
    # no robust kernel (BundleAdjustment bRobust=false, Optimizer.cc:185-190)
    w = synth.make_window(3, 1, 40, seed=31, run_lo=2, run_hi=4)
    w.huber_delta = 0.0
    save("lba_norobust", w, lm_dense(w))
    # stereo + monocular edges mixed (the reference's stereo branch, Optimizer.cc:673-705)
    w = synth.make_window(5, 2, 80, seed=41, run_lo=2, run_hi=5, stereo_frac=0.6)
    save("lba_stereo", w, lm_dense(w))
    cameras()


def cameras():
    # every keyframe with a camera (and baseline) of its own, three different ones in the window, stereo + monocular edges
    # (e->pCamera = pKFi->mpCamera, Optimizer.cc:664; e->fx .. e->bf from pKFi, :690-695)
    w = synth.mixed_cameras(synth.make_window(5, 2, 80, seed=47, run_lo=2, run_hi=5, stereo_frac=0.6), seed=48)
    save("lba_cameras", w, lm_dense(w))


if __name__ == "__main__":
    cameras() if sys.argv[1:] == ["cameras"] else main()
