"""ctypes wrapper over oracle/liblba_oracle.so.

TEST INFRASTRUCTURE ONLY (see lba_oracle.h): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg — never by the product
package mov-slam_amd/.  PARITY UNPINNED: the reference holds no fixtures for this path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liblba_oracle.so")
MAX_TRACE = 128

_d = C.POINTER(C.c_double)
_i = C.POINTER(C.c_int32)
_u = C.POINTER(C.c_uint8)


class _Problem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_points", C.c_int), ("n_edges", C.c_int),
                ("poses", _d), ("pose_fixed", _u), ("points", _d),
                ("edge_pose", _i), ("edge_point", _i), ("obs", _d), ("inv_sigma2", _d),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber_delta", C.c_double), ("chi2_gate", C.c_double),
                ("max_iters", C.c_int), ("stale_error_quirk", C.c_int), ("max_trials", C.c_int), ("stop", _u),
                ("obs_right", _d), ("bf", C.c_double), ("cam_kf", _d), ("bf_kf", _d)]


class _Result(C.Structure):
    _fields_ = [("poses", _d), ("points", _d), ("chi2", _d), ("outlier", _u),
                ("iters_done", C.c_int), ("n_solves", C.c_int), ("n_outliers", C.c_int),
                ("lambda_", C.c_double), ("cost", C.c_double), ("cost0", C.c_double),
                ("status", C.c_int), ("n_trace", C.c_int),
                ("tr_lambda", C.c_double * MAX_TRACE), ("tr_f0", C.c_double * MAX_TRACE),
                ("tr_f1", C.c_double * MAX_TRACE), ("tr_rho", C.c_double * MAX_TRACE),
                ("tr_accept", C.c_int * MAX_TRACE)]


class _PoseProblem(C.Structure):
    _fields_ = [("n", C.c_int), ("Xw", _d), ("obs", _d), ("inv_sigma2", _d),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("pose0", C.c_double * 7), ("huber_delta", C.c_double), ("chi2_gate", C.c_double),
                ("rounds", C.c_int), ("its_per_round", C.c_int)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lba_oracle.c")
    omp = os.path.join(_HERE, "liblba_oracle_omp.so")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src) or not os.path.exists(omp) \
            or os.path.getmtime(omp) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


_lib = None
_lib_omp = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.lba_oracle_solve.restype = C.c_int
        _lib.lba_oracle_linearize.restype = C.c_int
        _lib.lba_oracle_pose_opt.restype = C.c_int
    return _lib


def lib_omp():
    """The OpenMP build of the same source (all host cores): timing context only, never the parity reference."""
    global _lib_omp
    if _lib_omp is None:
        build()
        _lib_omp = C.CDLL(os.path.join(_HERE, "liblba_oracle_omp.so"))
        _lib_omp.lba_oracle_solve.restype = C.c_int
        _lib_omp.lba_oracle_set_threads.restype = C.c_int
    return _lib_omp


def host_core_share() -> int:
    """CPUs this process may really use: the cgroup quota when there is one, else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(float(quota) / float(period)))
    except (OSError, ValueError):
        pass
    return len(os.sched_getaffinity(0))


def set_omp_threads(n: int) -> int:
    return lib_omp().lba_oracle_set_threads(int(n))


def _p(a, t):
    return a.ctypes.data_as(t)


def _problem(w, stale_error_quirk=True, stop=None, max_iters=None, max_trials=0):
    keep = dict(
        poses=np.ascontiguousarray(w.poses, np.float64), fixed=np.ascontiguousarray(w.pose_fixed, np.uint8),
        points=np.ascontiguousarray(w.points, np.float64), ep=np.ascontiguousarray(w.edge_pose, np.int32),
        el=np.ascontiguousarray(w.edge_point, np.int32), obs=np.ascontiguousarray(w.obs, np.float64),
        isg=np.ascontiguousarray(w.inv_sigma2, np.float64))
    pb = _Problem()
    pb.n_poses, pb.n_points, pb.n_edges = w.n_poses, w.n_points, w.n_edges
    pb.poses = _p(keep["poses"], _d); pb.pose_fixed = _p(keep["fixed"], _u); pb.points = _p(keep["points"], _d)
    pb.edge_pose = _p(keep["ep"], _i); pb.edge_point = _p(keep["el"], _i)
    pb.obs = _p(keep["obs"], _d); pb.inv_sigma2 = _p(keep["isg"], _d)
    pb.fx, pb.fy, pb.cx, pb.cy = w.cam
    pb.huber_delta, pb.chi2_gate = w.huber_delta, w.chi2_gate
    pb.max_iters = w.max_iters if max_iters is None else max_iters
    pb.stale_error_quirk = 1 if stale_error_quirk else 0
    pb.max_trials = max_trials
    if stop is not None:
        keep["stop"] = stop
        pb.stop = _p(stop, _u)
    if getattr(w, "obs_right", None) is not None:
        keep["obs_right"] = np.ascontiguousarray(w.obs_right, np.float64)
        pb.obs_right = _p(keep["obs_right"], _d)
        pb.bf = float(w.bf)
    if getattr(w, "cam_kf", None) is not None:
        keep["cam_kf"] = np.ascontiguousarray(w.cam_kf, np.float64)
        pb.cam_kf = _p(keep["cam_kf"], _d)
    if getattr(w, "bf_kf", None) is not None:
        keep["bf_kf"] = np.ascontiguousarray(w.bf_kf, np.float64)
        pb.bf_kf = _p(keep["bf_kf"], _d)
    return pb, keep


def solve(w, stale_error_quirk=True, stop=None, max_iters=None, max_trials=0, omp=False) -> dict:
    """Optimizer::LocalBundleAdjustment's solve + outlier gate on a flattened window.
    omp=True runs the OpenMP build (all host cores, OMP_NUM_THREADS): timing context only."""
    pb, keep = _problem(w, stale_error_quirk, stop, max_iters, max_trials)
    poses = np.zeros((w.n_poses, 7)); points = np.zeros((w.n_points, 3))
    chi2 = np.zeros(w.n_edges); outlier = np.zeros(w.n_edges, np.uint8)
    res = _Result()
    res.poses = _p(poses, _d); res.points = _p(points, _d); res.chi2 = _p(chi2, _d); res.outlier = _p(outlier, _u)
    status = (lib_omp() if omp else lib()).lba_oracle_solve(C.byref(pb), C.byref(res))
    n = res.n_trace
    return dict(status=status, poses=poses, points=points, chi2=chi2, outlier=outlier,
                iters_done=res.iters_done, n_solves=res.n_solves, n_outliers=res.n_outliers,
                lam=res.lambda_, cost=res.cost, cost0=res.cost0,
                trace=dict(lam=np.array(res.tr_lambda[:n]), f0=np.array(res.tr_f0[:n]),
                           f1=np.array(res.tr_f1[:n]), rho=np.array(res.tr_rho[:n]),
                           accept=np.array(res.tr_accept[:n])))


def linearize(w, lam: float) -> dict:
    pb, keep = _problem(w)
    nf = int((np.asarray(w.pose_fixed) == 0).sum())
    n = 6 * nf
    Hpp = np.zeros((nf, 6, 6)); bp = np.zeros((nf, 6)); Hll = np.zeros((w.n_points, 3, 3)); bl = np.zeros((w.n_points, 3))
    S = np.zeros((n, n)); bS = np.zeros(n); fidx = np.zeros(w.n_poses, np.int32); F0 = C.c_double()
    nfree = lib().lba_oracle_linearize(C.byref(pb), C.c_double(lam), _p(Hpp, _d), _p(bp, _d), _p(Hll, _d), _p(bl, _d),
                                       _p(S, _d), _p(bS, _d), _p(fidx, _i), C.byref(F0))
    m = 6 * nfree
    return dict(nfree=nfree, Hpp=Hpp[:nfree], bp=bp[:nfree], Hll=Hll, bl=bl,
                S=S.ravel()[:m * m].reshape(m, m).copy(), bS=bS[:m], free_index=fidx, F0=F0.value)


def pose_opt(Xw, obs, pose0, cam, huber_delta, chi2_gate, rounds=4, its=10, inv_sigma2=None) -> dict:
    n = len(Xw)
    Xw = np.ascontiguousarray(Xw, np.float64); obs = np.ascontiguousarray(obs, np.float64)
    isg = np.ones(n) if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float64)
    pb = _PoseProblem()
    pb.n = n; pb.Xw = _p(Xw, _d); pb.obs = _p(obs, _d); pb.inv_sigma2 = _p(isg, _d)
    pb.fx, pb.fy, pb.cx, pb.cy = cam
    pb.pose0 = (C.c_double * 7)(*pose0)
    pb.huber_delta, pb.chi2_gate, pb.rounds, pb.its_per_round = huber_delta, chi2_gate, rounds, its
    pose = np.zeros(7); outl = np.zeros(n, np.uint8); chi2 = np.zeros(n)
    ninl = lib().lba_oracle_pose_opt(C.byref(pb), _p(pose, _d), _p(outl, _u), _p(chi2, _d))
    return dict(n_inliers=ninl, pose=pose, outlier=outl, chi2=chi2)


def pose_ransac(Xw, obs, pose0, cam, chi2_gate, samples, inv_sigma2=None, confidence=0.0, lo_its=0) -> dict:
    """Hypothesis stage of PoseOptimization: best P3P pose over the given minimal samples ((H, 3) match indices); with
    cv::solvePnPRansac's stopping rule (confidence) and one local-optimisation step (lo_its LM iterations on the inliers)."""
    n = len(Xw)
    Xw = np.ascontiguousarray(Xw, np.float64); obs = np.ascontiguousarray(obs, np.float64)
    isg = np.ones(n) if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float64)
    samples = np.ascontiguousarray(samples, np.int32)
    pb = _PoseProblem()
    pb.n = n; pb.Xw = _p(Xw, _d); pb.obs = _p(obs, _d); pb.inv_sigma2 = _p(isg, _d)
    pb.fx, pb.fy, pb.cx, pb.cy = cam
    pb.pose0 = (C.c_double * 7)(*pose0)
    pb.huber_delta, pb.chi2_gate, pb.rounds, pb.its_per_round = 0.0, chi2_gate, 0, 0
    pose = np.zeros(7); info = np.zeros(3, np.int32)
    L = lib(); L.lba_oracle_pose_ransac_lo.restype = C.c_int
    cnt = L.lba_oracle_pose_ransac_lo(C.byref(pb), C.c_int(len(samples)), _p(samples, _i), C.c_double(confidence), C.c_int(lo_its),
                                      _p(pose, _d), _p(info, _i))
    return dict(n_inliers=cnt, pose=pose, samples_used=int(info[0]), lo_accepted=int(info[1]), lo_inliers=int(info[2]))


def se3_exp(u):
    out = np.zeros(7); lib().lba_oracle_se3_exp(_p(np.ascontiguousarray(u, np.float64), _d), _p(out, _d)); return out


def se3_mul(a, b):
    out = np.zeros(7)
    lib().lba_oracle_se3_mul(_p(np.ascontiguousarray(a, np.float64), _d), _p(np.ascontiguousarray(b, np.float64), _d), _p(out, _d))
    return out


def se3_map(qt, X):
    out = np.zeros(3)
    lib().lba_oracle_se3_map(_p(np.ascontiguousarray(qt, np.float64), _d), _p(np.ascontiguousarray(X, np.float64), _d), _p(out, _d))
    return out


def edge(qt, X, obs, cam):
    err = np.zeros(2); Jp = np.zeros((2, 3)); Jc = np.zeros((2, 6))
    a = [np.ascontiguousarray(v, np.float64) for v in (qt, X, obs, cam)]
    lib().lba_oracle_edge(_p(a[0], _d), _p(a[1], _d), _p(a[2], _d), _p(a[3], _d), _p(err, _d), _p(Jp, _d), _p(Jc, _d))
    return err, Jp, Jc


def huber(chi2, delta):
    rho = np.zeros(3); lib().lba_oracle_huber(C.c_double(chi2), C.c_double(delta), _p(rho, _d)); return rho
