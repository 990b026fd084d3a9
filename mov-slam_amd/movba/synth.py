"""Seeded synthetic covisibility windows (BASELINE.md cfg 1-3, 5).

The generator stands in for the graph `Optimizer::LocalBundleAdjustment` builds at
/root/reference/src/Optimizer.cc:464-747: K free + F fixed keyframe vertices, P map
points, one monocular edge per (point, observing keyframe), pinhole fx=fy=320,
cx=320, cy=240 on 640x480 (Examples/Monocular/TartanAir.yaml:15-26).  All values are
rounded to float32 and widened, mimicking the float->double casts at
Optimizer.cc:559, 627, 650.  Edge order: point id ascending, then keyframe id
ascending (stands in for the reference's std::map<KeyFrame*> pointer order).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

FX = FY = 320.0
CX, CY = 320.0, 240.0
WIDTH, HEIGHT = 640, 480
HUBER_DELTA = float(np.sqrt(np.float32(5.0)))  # (double)sqrtf(5.0f): Optimizer.cc:616
CHI2_GATE = 5.0                                   # static float delta, Optimizer.cc:52, 769


@dataclass
class Window:
    """Flattened local-BA window in the layout the C-ABI takes (include/movba.h)."""
    poses: np.ndarray        # (NP,7) f64  qx qy qz qw tx ty tz (Tcw), ascending keyframe id
    pose_fixed: np.ndarray   # (NP,)  u8
    points: np.ndarray       # (P,3)  f64
    edge_pose: np.ndarray    # (E,)   i32
    edge_point: np.ndarray   # (E,)   i32
    obs: np.ndarray          # (E,2)  f64
    inv_sigma2: np.ndarray   # (E,)   f64
    cam: tuple = (FX, FY, CX, CY)
    huber_delta: float = HUBER_DELTA
    chi2_gate: float = CHI2_GATE
    max_iters: int = 10
    truth_poses: np.ndarray | None = None
    truth_points: np.ndarray | None = None
    meta: dict = field(default_factory=dict)
    obs_right: np.ndarray | None = None   # (E,) f64: right-image u of stereo observations, < 0 = monocular edge
    bf: float = 0.0                       # KeyFrame::mbf (baseline * fx)
    cam_kf: np.ndarray | None = None      # (NP,4) f64: fx fy cx cy by keyframe (None: `cam` for all, every shipped configuration)
    bf_kf: np.ndarray | None = None       # (NP,)  f64: mbf by keyframe

    @property
    def n_poses(self): return int(self.poses.shape[0])
    @property
    def n_free(self): return int((self.pose_fixed == 0).sum())
    @property
    def n_points(self): return int(self.points.shape[0])
    @property
    def n_edges(self): return int(self.edge_pose.shape[0])


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def quat_from_R(R):
    """Rotation matrix -> unit quaternion (x,y,z,w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def R_from_quat(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _rodrigues(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def make_window(n_free: int, n_fixed: int, n_points: int, seed: int,
                run_lo: int = 2, run_hi: int = 10, outlier_frac: float = 0.05,
                pix_sigma: float = 0.5, rot_sigma_deg: float = 0.5, trans_sigma: float = 0.02,
                point_sigma: float = 0.05, min_obs: int = 2, stereo_frac: float = 0.0, bf: float = 80.0) -> Window:
    rng = np.random.default_rng(seed)
    NP = n_free + n_fixed
    k = np.arange(NP, dtype=np.float64)
    centres = np.stack([0.30 * k, 0.05 * np.sin(0.3 * k), 0.02 * k], axis=1)
    yaw = 0.01 * k
    Rcw = np.zeros((NP, 3, 3))
    tcw = np.zeros((NP, 3))
    for i in range(NP):
        c, s = np.cos(yaw[i]), np.sin(yaw[i])
        Rwc = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])   # yaw about the camera's y axis
        Rcw[i] = Rwc.T
        tcw[i] = -Rcw[i] @ centres[i]

    # candidate points until P of them have >= min_obs valid observations
    pts, e_pose, e_point, e_uv = [], [], [], []
    n_kept = 0
    while n_kept < n_points:
        m = max(256, (n_points - n_kept) * 2)
        anchor = rng.integers(0, NP, size=m)
        depth = rng.uniform(4.0, 30.0, size=m)
        u = rng.uniform(0.0, WIDTH, size=m)
        v = rng.uniform(0.0, HEIGHT, size=m)
        run = rng.integers(run_lo, run_hi + 1, size=m)
        off = rng.integers(0, run_hi + 1, size=m)
        for j in range(m):
            if n_kept >= n_points:
                break
            a = int(anchor[j])
            Xc = np.array([(u[j] - CX) / FX * depth[j], (v[j] - CY) / FY * depth[j], depth[j]])
            Xw = Rcw[a].T @ (Xc - tcw[a])
            first = a - int(off[j]) % int(run[j])
            ks = [kk for kk in range(first, first + int(run[j])) if 0 <= kk < NP]
            uv_ok = []
            for kk in ks:
                Y = Rcw[kk] @ Xw + tcw[kk]
                if Y[2] <= 0.1:
                    continue
                uu, vv = FX * Y[0] / Y[2] + CX, FY * Y[1] / Y[2] + CY
                if 0 <= uu < WIDTH and 0 <= vv < HEIGHT:
                    uv_ok.append((kk, uu, vv))
            if len(uv_ok) < min_obs:
                continue
            pts.append(Xw)
            for kk, uu, vv in uv_ok:
                e_pose.append(kk); e_point.append(n_kept); e_uv.append((uu, vv))
            n_kept += 1

    truth_points = np.array(pts)
    edge_pose = np.array(e_pose, dtype=np.int32)
    edge_point = np.array(e_point, dtype=np.int32)
    obs = np.array(e_uv, dtype=np.float64)
    E = len(edge_pose)
    obs = obs + rng.normal(0.0, pix_sigma, size=(E, 2))
    is_out = rng.random(E) < outlier_frac
    mag = rng.uniform(5.0, 30.0, size=E)
    ang = rng.uniform(0.0, 2 * np.pi, size=E)
    obs[is_out] += np.stack([mag * np.cos(ang), mag * np.sin(ang)], axis=1)[is_out]

    # stereo observations (Optimizer.cc:673-705): u_right = u - bf / z (+ noise) for a fraction of the edges
    obs_right = None
    if stereo_frac > 0.0:
        Xc_all = np.einsum('eij,ej->ei', Rcw[edge_pose], truth_points[edge_point]) + tcw[edge_pose]
        ur = obs[:, 0] - bf / Xc_all[:, 2] + rng.normal(0.0, pix_sigma, size=E)
        st = (rng.random(E) < stereo_frac) & (ur >= 0.0)
        obs_right = np.where(st, ur, -1.0)

    truth_poses = np.zeros((NP, 7))
    poses = np.zeros((NP, 7))
    fixed = np.zeros(NP, dtype=np.uint8)
    fixed[:n_fixed] = 1                                     # the F lowest-id keyframes, exact truth
    for i in range(NP):
        truth_poses[i, :4] = quat_from_R(Rcw[i]); truth_poses[i, 4:] = tcw[i]
        if fixed[i]:
            poses[i] = truth_poses[i]
        else:
            dR = _rodrigues(np.deg2rad(rot_sigma_deg) * rng.normal(size=3))
            poses[i, :4] = quat_from_R(dR @ Rcw[i])
            poses[i, 4:] = tcw[i] + trans_sigma * rng.normal(size=3)
    points = truth_points + point_sigma * rng.normal(size=truth_points.shape)

    poses = _f32(poses)
    poses[:, :4] /= np.linalg.norm(poses[:, :4], axis=1, keepdims=True)   # cast<double> of a unit float quat, renormalised by SE3Quat
    return Window(poses=poses, pose_fixed=fixed, points=_f32(points), edge_pose=edge_pose,
                  edge_point=edge_point, obs=_f32(obs), inv_sigma2=np.ones(E),
                  truth_poses=truth_poses, truth_points=truth_points,
                  meta=dict(seed=seed, K=n_free, F=n_fixed, P=n_points, E=E, outliers=int(is_out.sum())),
                  obs_right=None if obs_right is None else np.where(obs_right >= 0, _f32(obs_right), -1.0), bf=bf if obs_right is not None else 0.0)


def cfg(name: str, seed: int | None = None) -> Window:
    """Named BASELINE.md configurations."""
    if name == "tiny":      # golden-fixture size (3 KF x 20 points)
        return make_window(2, 1, 20, 7 if seed is None else seed, run_lo=2, run_hi=3)
    if name == "small":     # golden-fixture size (10 KF x 200 points)
        return make_window(8, 2, 200, 11 if seed is None else seed, run_lo=2, run_hi=6)
    if name == "cfg2":      # LBA 10 KF x 2k MapPoints
        return make_window(10, 2, 2000, 1002 if seed is None else seed, run_lo=2, run_hi=6)
    if name == "cfg3":      # LBA 50 KF x 20k MapPoints, Huber on
        return make_window(50, 10, 20000, 1003 if seed is None else seed, run_lo=2, run_hi=10)
    raise KeyError(name)


def make_frame(n: int = 500, seed: int = 1001, outlier_frac: float = 0.10, pix_sigma: float = 0.5):
    """cfg1: one Frame with n 2D-3D matches (PoseOptimization operand, Optimizer.cc:404-413)."""
    rng = np.random.default_rng(seed)
    depth = rng.uniform(4.0, 30.0, size=n)
    u = rng.uniform(0, WIDTH, size=n); v = rng.uniform(0, HEIGHT, size=n)
    Xc = np.stack([(u - CX) / FX * depth, (v - CY) / FY * depth, depth], axis=1)
    Rcw = _rodrigues(np.array([0.02, -0.03, 0.01])); tcw = np.array([0.3, -0.1, 0.2])
    Xw = (Xc - tcw) @ Rcw                                    # Rcw^T (Xc - t)
    obs = np.stack([u, v], axis=1) + rng.normal(0, pix_sigma, size=(n, 2))
    is_out = rng.random(n) < outlier_frac
    mag = rng.uniform(10.0, 60.0, size=n); ang = rng.uniform(0, 2 * np.pi, size=n)
    obs[is_out] += np.stack([mag * np.cos(ang), mag * np.sin(ang)], axis=1)[is_out]
    truth = np.concatenate([quat_from_R(Rcw), tcw])
    dR = _rodrigues(np.deg2rad(1.0) * rng.normal(size=3))
    pose0 = np.concatenate([quat_from_R(dR @ Rcw), tcw + 0.05 * rng.normal(size=3)])
    pose0 = _f32(pose0); pose0[:4] /= np.linalg.norm(pose0[:4])
    return dict(Xw=_f32(Xw), obs=_f32(obs), pose0=pose0, truth=truth, is_outlier=is_out,
                cam=(FX, FY, CX, CY))


# ---------------------------------------------------------------------------------------------
# Covisibility patterns beyond the contiguous-run generator above (round 3).
#
# make_window() gives every map point a contiguous run of observing keyframes, and keyframe ids follow the
# trajectory: the reduced system is banded.  The reference's local window is "every keyframe sharing >= 15 points
# with pKF" (/root/reference/src/KeyFrame.cc:227-231, 408-427; selection src/Optimizer.cc:464-477): all of them
# covisible through pKF and mostly with each other, ids in order of creation, not of position.  The patterns below
# produce such windows with consistent geometry (every observation is the projection of its map point):
#   hub      keyframes clustered around one place, all looking at the same scene: every pair shares points,
#   revisit  the trajectory comes back over itself: keyframe k and k + 25 see the same points,
#   shuffle_ids()  any window with its keyframe ids permuted (same graph, id order != spatial order).
# ---------------------------------------------------------------------------------------------
def _poses_from(centres, yaw):
    NP = len(yaw)
    Rcw = np.zeros((NP, 3, 3)); tcw = np.zeros((NP, 3))
    for i in range(NP):
        c, s = np.cos(yaw[i]), np.sin(yaw[i])
        Rwc = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        Rcw[i] = Rwc.T
        tcw[i] = -Rcw[i] @ centres[i]
    return Rcw, tcw


def make_pattern_window(pattern: str, n_free: int, n_fixed: int, n_points: int, seed: int,
                        run_lo: int = 2, run_hi: int = 10, outlier_frac: float = 0.05, pix_sigma: float = 0.5,
                        rot_sigma_deg: float = 0.5, trans_sigma: float = 0.02, point_sigma: float = 0.05,
                        min_obs: int = 2, revisit_gap: int = 25) -> Window:
    """Window with the covisibility pattern `pattern` ("hub" | "revisit"); noise model, camera, float32 rounding, edge order
    and the choice of fixed keyframes (the n_fixed lowest ids) as make_window()."""
    rng = np.random.default_rng(seed)
    NP = n_free + n_fixed
    k = np.arange(NP, dtype=np.float64)
    if pattern == "hub":
        centres = np.stack([rng.uniform(-1.5, 1.5, NP), rng.uniform(-0.3, 0.3, NP), rng.uniform(-1.0, 1.0, NP)], axis=1)
        yaw = rng.uniform(-0.15, 0.15, NP)
        order_key = None
    elif pattern == "revisit":
        # second pass over the same stretch of path, half a keyframe spacing out of step and 0.2 m to the side
        first = NP - revisit_gap if NP > revisit_gap else NP
        s = np.where(k < first, k, k - revisit_gap + 0.5)
        side = np.where(k < first, 0.0, 0.2)
        centres = np.stack([0.30 * s, 0.05 * np.sin(0.3 * s) + side, 0.02 * s], axis=1)
        yaw = 0.01 * s
        order_key = s
    else:
        raise KeyError(pattern)
    Rcw, tcw = _poses_from(centres, yaw)
    by_path = np.argsort(order_key, kind="stable") if order_key is not None else None
    path_rank = np.argsort(by_path) if by_path is not None else None

    pts, e_pose, e_point, e_uv = [], [], [], []
    n_kept = 0
    while n_kept < n_points:
        m = max(256, (n_points - n_kept) * 2)
        anchor = rng.integers(0, NP, size=m)
        depth = rng.uniform(6.0 if pattern == "hub" else 4.0, 30.0, size=m)
        u = rng.uniform(0.0, WIDTH, size=m); v = rng.uniform(0.0, HEIGHT, size=m)
        run = rng.integers(run_lo, run_hi + 1, size=m)
        off = rng.integers(0, run_hi + 1, size=m)
        Xc = np.stack([(u - CX) / FX * depth, (v - CY) / FY * depth, depth], axis=1)
        Xw = np.einsum('mji,mj->mi', Rcw[anchor], Xc - tcw[anchor])        # Rcw^T (Xc - t)
        Y = np.einsum('kij,mj->mki', Rcw, Xw) + tcw[None, :, :]             # (m, NP, 3): the candidates in every camera
        z = Y[:, :, 2]
        zs = np.where(z > 0.1, z, 1.0)
        uu = FX * Y[:, :, 0] / zs + CX; vv = FY * Y[:, :, 1] / zs + CY
        vis = (z > 0.1) & (uu >= 0) & (uu < WIDTH) & (vv >= 0) & (vv < HEIGHT)
        for j in range(m):
            if n_kept >= n_points:
                break
            if pattern == "hub":
                cand = np.flatnonzero(vis[j])
                if len(cand) < min_obs:
                    continue
                ks = np.sort(rng.choice(cand, size=min(int(run[j]), len(cand)), replace=False))
            else:
                a = int(path_rank[anchor[j]])
                first_r = a - int(off[j]) % int(run[j])
                ranks = [r for r in range(first_r, first_r + int(run[j])) if 0 <= r < NP]
                ks = np.sort([int(by_path[r]) for r in ranks if vis[j, by_path[r]]])
            if len(ks) < min_obs:
                continue
            pts.append(Xw[j])
            for kk in ks:
                e_pose.append(int(kk)); e_point.append(n_kept); e_uv.append((uu[j, kk], vv[j, kk]))
            n_kept += 1

    truth_points = np.array(pts)
    edge_pose = np.array(e_pose, dtype=np.int32); edge_point = np.array(e_point, dtype=np.int32)
    obs = np.array(e_uv, dtype=np.float64)
    E = len(edge_pose)
    obs = obs + rng.normal(0.0, pix_sigma, size=(E, 2))
    is_out = rng.random(E) < outlier_frac
    mag = rng.uniform(5.0, 30.0, size=E); ang = rng.uniform(0.0, 2 * np.pi, size=E)
    obs[is_out] += np.stack([mag * np.cos(ang), mag * np.sin(ang)], axis=1)[is_out]

    truth_poses = np.zeros((NP, 7)); poses = np.zeros((NP, 7))
    fixed = np.zeros(NP, dtype=np.uint8); fixed[:n_fixed] = 1
    for i in range(NP):
        truth_poses[i, :4] = quat_from_R(Rcw[i]); truth_poses[i, 4:] = tcw[i]
        if fixed[i]:
            poses[i] = truth_poses[i]
        else:
            dR = _rodrigues(np.deg2rad(rot_sigma_deg) * rng.normal(size=3))
            poses[i, :4] = quat_from_R(dR @ Rcw[i]); poses[i, 4:] = tcw[i] + trans_sigma * rng.normal(size=3)
    points = truth_points + point_sigma * rng.normal(size=truth_points.shape)
    poses = _f32(poses)
    poses[:, :4] /= np.linalg.norm(poses[:, :4], axis=1, keepdims=True)
    return Window(poses=poses, pose_fixed=fixed, points=_f32(points), edge_pose=edge_pose, edge_point=edge_point,
                  obs=_f32(obs), inv_sigma2=np.ones(E), truth_poses=truth_poses, truth_points=truth_points,
                  meta=dict(seed=seed, K=n_free, F=n_fixed, P=n_points, E=E, outliers=int(is_out.sum()), pattern=pattern))


def shuffle_ids(w: Window, seed: int) -> Window:
    """The same graph with the keyframe ids permuted (fixed flags, estimates and observations follow their keyframe); edges
    re-sorted into the reference's order (point ascending, then keyframe id ascending)."""
    rng = np.random.default_rng(seed)
    NP = w.n_poses
    new_of_old = rng.permutation(NP)                    # keyframe `old` becomes id new_of_old[old]
    old_of_new = np.argsort(new_of_old)
    ep = new_of_old[w.edge_pose].astype(np.int32)
    order = np.lexsort((ep, w.edge_point))
    meta = dict(w.meta); meta["pattern"] = (meta.get("pattern", "run") + "+shuffled"); meta["id_perm"] = new_of_old
    return Window(poses=w.poses[old_of_new], pose_fixed=w.pose_fixed[old_of_new], points=w.points,
                  edge_pose=ep[order], edge_point=w.edge_point[order], obs=w.obs[order], inv_sigma2=w.inv_sigma2[order],
                  cam=w.cam, huber_delta=w.huber_delta, chi2_gate=w.chi2_gate, max_iters=w.max_iters,
                  truth_poses=None if w.truth_poses is None else w.truth_poses[old_of_new], truth_points=w.truth_points,
                  meta=meta, obs_right=None if w.obs_right is None else w.obs_right[order], bf=w.bf)


def mixed_cameras(w: Window, seed: int, models=((320.0, 320.0, 320.0, 240.0), (400.0, 410.0, 330.0, 250.0), (281.5, 279.25, 311.0, 236.5)),
                  bf_scale=(1.0, 1.25, 0.8)) -> Window:
    """The same window seen through DIFFERENT cameras: every keyframe draws one of `models` (and a baseline factor), its
    observations are mapped pixel by pixel - u' = fx' (u - cx) / fx + cx', u_r' = u' - (bf' / bf) (u - u_r) - so geometry,
    noise and outliers carry over.  The reference gives every edge its keyframe's camera (src/Optimizer.cc:664, 690-695);
    `cam` / `bf` of the result are deliberately useless (the first model / 0) so that a path reading them shows up."""
    import dataclasses
    rng = np.random.default_rng(seed)
    NP = w.n_poses
    pick = rng.integers(0, len(models), NP)
    pick[: len(models)] = np.arange(len(models))[: min(len(models), NP)]            # every model occurs
    cam_kf = _f32(np.asarray(models)[pick])
    fx, fy, cx, cy = w.cam
    k = cam_kf[w.edge_pose]
    obs = np.stack([k[:, 0] * (w.obs[:, 0] - cx) / fx + k[:, 2], k[:, 1] * (w.obs[:, 1] - cy) / fy + k[:, 3]], axis=1)
    out = dataclasses.replace(w, obs=_f32(obs), cam=tuple(float(v) for v in cam_kf[0]), cam_kf=cam_kf)
    if w.obs_right is not None:
        bf_kf = _f32(w.bf * np.asarray(bf_scale)[pick])
        st = w.obs_right >= 0
        ur = np.where(st, obs[:, 0] - (bf_kf[w.edge_pose] / w.bf) * (w.obs[:, 0] - w.obs_right), -1.0)
        # (a disparity that leaves the image on the left is no stereo observation any more)
        ur = np.where(st & (ur >= 0), ur, -1.0)
        out = dataclasses.replace(out, obs_right=np.where(ur >= 0, _f32(ur), -1.0), bf=0.0, bf_kf=bf_kf)
    return out


def pattern_cfg(name: str, seed: int | None = None) -> Window:
    """cfg3-sized windows (50 + 10 keyframes x 20 000 map points) of the other covisibility patterns."""
    if name == "hub":
        return make_pattern_window("hub", 50, 10, 20000, 3001 if seed is None else seed)
    if name == "revisit":
        return make_pattern_window("revisit", 50, 10, 20000, 3002 if seed is None else seed, run_lo=2, run_hi=10)
    if name == "shuffled":
        return shuffle_ids(cfg("cfg3"), 3003 if seed is None else seed)
    raise KeyError(name)
