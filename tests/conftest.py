"""Shared fixtures.  `-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbols.
`-m gpu`: parity of the HIP path (through the C-ABI) against the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (Window built from the fixture's inputs, dict of expected outputs)."""
    from movba import synth
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    w = synth.Window(poses=g["in_poses"], pose_fixed=g["in_pose_fixed"], points=g["in_points"],
                     edge_pose=g["in_edge_pose"], edge_point=g["in_edge_point"], obs=g["in_obs"],
                     inv_sigma2=g["in_inv_sigma2"], cam=tuple(g["in_cam"]),
                     huber_delta=float(g["in_huber_delta"]), chi2_gate=float(g["in_chi2_gate"]),
                     max_iters=int(g["in_max_iters"]))
    if "in_obs_right" in g.files and g["in_obs_right"].size:
        w.obs_right, w.bf = g["in_obs_right"], float(g["in_bf"])
    if "in_cam_kf" in g.files:
        w.cam_kf = g["in_cam_kf"]
    if "in_bf_kf" in g.files:
        w.bf_kf = g["in_bf_kf"]
    out = {k[4:]: g[k] for k in g.files if k.startswith("out_")}
    return w, out


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def built_lib():
    """libmovba.so must exist for every test that touches the C-ABI (cross-compiled on CPU)."""
    from movba import capi
    if not os.path.exists(capi.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mov-slam_amd", "csrc"), "-s"])
    return capi


@pytest.fixture(scope="session")
def solver(built_lib):
    s = built_lib.Solver()          # raises loudly if the HIP library or the device is missing
    yield s
    s.close()


def quat_angle(q1, q2):
    """geodesic angle between unit quaternions (x,y,z,w), rows; accurate down to 1e-16
    (|q1 - q2| = 2 sin(theta/4); arccos of the dot product would floor at 3e-8)."""
    d = np.minimum(np.linalg.norm(q1 - q2, axis=-1), np.linalg.norm(q1 + q2, axis=-1))
    return 4 * np.arcsin(np.clip(d / 2, 0, 1))


def oracle_order_noise(oracle_mod, w, n=3):
    """How far the oracle's own result moves when ONLY the order of a map point's edges changes: the reference adds a point's
    edges in the iteration order of a std::map keyed by KeyFrame POINTERS (MapPoint::GetObservations(), src/Optimizer.cc:629-700),
    which differs from run to run, so every such order is the reference's arithmetic.  Well-conditioned windows move by 1e-13;
    windows whose keyframes hang on a few short tracks move by 1e-8 ... 1e-6 m, and no solver can be held closer to ONE of those
    orders than they are to each other.  -> (rotation [rad], translation [m], point [m]) maxima over n seeded permutations."""
    import copy
    o = oracle_mod.solve(w)
    rot = trans = point = 0.0
    for t in range(n):
        pm = np.random.default_rng(7919 + t).permutation(w.n_edges)
        pm = pm[np.argsort(w.edge_point[pm], kind="stable")]            # still grouped by point, shuffled inside a group
        w2 = copy.copy(w)
        w2.edge_pose, w2.edge_point, w2.obs, w2.inv_sigma2 = w.edge_pose[pm], w.edge_point[pm], w.obs[pm], w.inv_sigma2[pm]
        if getattr(w, "obs_right", None) is not None: w2.obs_right = w.obs_right[pm]
        o2 = oracle_mod.solve(w2)
        rot = max(rot, float(quat_angle(o2["poses"][:, :4], o["poses"][:, :4]).max()))
        trans = max(trans, float(np.abs(o2["poses"][:, 4:] - o["poses"][:, 4:]).max()))
        point = max(point, float(np.abs(o2["points"] - o["points"]).max()) if w.n_points else 0.0)
    return rot, trans, point
