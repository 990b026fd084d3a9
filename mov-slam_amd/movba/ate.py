"""Trajectory writer and ATE metric for BASELINE config 4 (TartanAir end-to-end), restated.

* `kf_trajectory_rows` follows System::saveKeyFrameTrajectoryKITTI
  (/root/reference/src/System.cc:722-776): keyframes sorted by id, every pose expressed relative to the
  first keyframe, one row `mnFrameId r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz` of Twc.
* `ate_tartanair` follows the reference's evaluation protocol
  (evaluation/tartanair_eval/evaluation/tartanair_evaluator.py:19-71): row 0 of the estimate is skipped,
  the ground-truth row is picked by frame id, the estimate goes from the KITTI camera frame to TartanAir's
  NED frame (trajectory_transform.py:57-75), and the ATE is the RMSE of the translation after Horn's
  closed-form alignment with the ESTIMATE scaled to the ground truth (evaluate_ate_scale.py:50-103,
  evaluator_base.py:26-53; scale=True is the monocular track).
The full cfg4 run needs the reference's whole stack and the TartanAir images (absent here); what this
module pins is the metric itself: tests/test_ate.py reproduces the reference evaluator's 6.7469 on the
trajectory pair the reference commits.
"""
from __future__ import annotations

import numpy as np

from . import synth


def kf_trajectory_rows(poses_tcw: np.ndarray, frame_ids) -> np.ndarray:
    """poses_tcw: (N,7) qx qy qz qw tx ty tz (Tcw), in keyframe-id order -> (N,13) rows."""
    def T(p):
        M = np.eye(4); M[:3, :3] = synth.R_from_quat(p[:4] / np.linalg.norm(p[:4])); M[:3, 3] = p[4:]; return M
    Tow = np.linalg.inv(T(poses_tcw[0]))
    rows = []
    for p, fid in zip(poses_tcw, frame_ids):
        Twc = np.linalg.inv(T(p) @ Tow)
        rows.append(np.concatenate([[float(fid)], Twc[:3, :].reshape(-1)]))
    return np.array(rows)


def kitti_to_tartan_xyz(est_3x4: np.ndarray) -> np.ndarray:
    """Position part of kitti2tartan: T t with T = [[0,0,1],[1,0,0],[0,1,0]] (camera -> NED)."""
    t = est_3x4[:, [3, 7, 11]]
    return np.stack([t[:, 2], t[:, 0], t[:, 1]], axis=1)


def horn_align(model: np.ndarray, data: np.ndarray, calc_scale: bool):
    """model, data: (3,n).  Returns rot, trans, per-point error, scale (estimate scaled to the model)."""
    mz = model - model.mean(1, keepdims=True)
    dz = data - data.mean(1, keepdims=True)
    W = mz @ dz.T
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    rot = U @ S @ Vh
    if calc_scale:
        rotmodel = rot @ mz
        dots = float((dz * rotmodel).sum())
        norms = float((mz * mz).sum())
        s = norms / dots
    else:
        s = 1.0
    trans = s * data.mean(1, keepdims=True) - rot @ model.mean(1, keepdims=True)
    err = rot @ model + trans - s * data
    return rot, trans, np.sqrt((err * err).sum(0)), s


def ate_tartanair(gt_traj: np.ndarray, est_rows: np.ndarray, scale: bool = True) -> dict:
    """gt_traj: (M,7) TartanAir pose_left rows (x y z qx qy qz qw); est_rows: (N,13) keyframe trajectory."""
    est = est_rows[1:]                                   # "ignore frame 0"
    gt = gt_traj[est[:, 0].astype(int)]
    est_xyz = kitti_to_tartan_xyz(est[:, 1:])
    _, _, err, s = horn_align(gt[:, :3].T, est_xyz.T, scale)
    return dict(ate=float(np.sqrt(err @ err / len(err))), scale=float(s), n=int(len(err)))
