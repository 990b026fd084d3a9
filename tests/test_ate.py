"""BASELINE config 4's metric: the trajectory writer and the TartanAir ATE protocol, restated in
mov-slam_amd/movba/ate.py and pinned on the trajectory pair the reference commits."""
import os

import numpy as np

from conftest import GOLDEN
from movba import ate, capture, synth


def test_ate_reproduces_the_reference_evaluators_number_on_its_committed_sample():
    """evaluation/tartanair_eval/evaluation/{pose_est,pose_gt}.txt through the reference's own
    tartanair_evaluator.py give ate_score 6.7469, ATE scale 9.5516 over 591 keyframes."""
    d = np.load(os.path.join(GOLDEN, "tartanair_sample_trajectories.npz"))      # tests/golden/make_tartanair_fixture.py
    r = ate.ate_tartanair(d["pose_gt"], d["pose_est"], scale=True)
    # expected_* are what the reference's own evaluator returned on the pair when the fixture was generated
    assert r["n"] == int(d["expected_n"]) == 591
    assert abs(r["ate"] - float(d["expected_ate"])) < 1e-9 and abs(float(d["expected_ate"]) - 6.7469) < 5e-5
    assert abs(r["scale"] - float(d["expected_scale"])) < 1e-9 and abs(float(d["expected_scale"]) - 9.5516) < 5e-5


def test_ate_is_zero_for_a_scaled_rotated_copy_and_positive_with_noise():
    rng = np.random.default_rng(0)
    n = 50
    gt = np.zeros((n, 7)); gt[:, :3] = np.cumsum(rng.normal(size=(n, 3)), 0); gt[:, 6] = 1
    # estimate = ground truth in the camera frame convention, scaled by 0.25: x_cam = y_ned, y_cam = z_ned, z_cam = x_ned
    est = np.zeros((n, 13)); est[:, 0] = np.arange(n)
    R = np.eye(3)
    for i in range(n):
        t_cam = 0.25 * np.array([gt[i, 1], gt[i, 2], gt[i, 0]])
        est[i, 1:] = np.hstack([R, t_cam[:, None]]).reshape(-1)
    r = ate.ate_tartanair(gt, est, scale=True)
    assert r["ate"] < 1e-9 and abs(r["scale"] - 4.0) < 1e-9
    est[:, [4, 8, 12]] += 0.01 * rng.normal(size=(n, 3))
    assert ate.ate_tartanair(gt, est, scale=True)["ate"] > 1e-3


def test_keyframe_trajectory_rows_follow_the_reference_writer():
    """System::saveKeyFrameTrajectoryKITTI: Twc relative to the first keyframe, frame id first."""
    w = synth.cfg("tiny")
    rows = ate.kf_trajectory_rows(w.truth_poses, frame_ids=[10, 20, 30])
    assert rows.shape == (3, 13) and rows[:, 0].tolist() == [10, 20, 30]
    np.testing.assert_allclose(rows[0, 1:].reshape(3, 4), np.hstack([np.eye(3), np.zeros((3, 1))]), atol=1e-12)
    # second row: camera centre of keyframe 1 expressed in keyframe 0's frame
    T0 = np.eye(4); T0[:3, :3] = synth.R_from_quat(w.truth_poses[0, :4]); T0[:3, 3] = w.truth_poses[0, 4:]
    T1 = np.eye(4); T1[:3, :3] = synth.R_from_quat(w.truth_poses[1, :4]); T1[:3, 3] = w.truth_poses[1, 4:]
    np.testing.assert_allclose(rows[1, 1:].reshape(3, 4), (T0 @ np.linalg.inv(T1))[:3], atol=1e-12)


def test_window_capture_round_trip(tmp_path):
    w = synth.cfg("small")
    p = str(tmp_path / "lba_000000.mbw")
    capture.save_window(p, w)
    v = capture.load_window(p)
    for f in ("poses", "pose_fixed", "points", "edge_pose", "edge_point", "obs", "inv_sigma2"):
        assert np.array_equal(getattr(w, f), getattr(v, f)), f
    assert v.cam == w.cam and v.huber_delta == w.huber_delta and v.chi2_gate == w.chi2_gate and v.max_iters == w.max_iters
