"""ctypes binding of libmovba.so (include/movba.h) — the Python-side stand-in for the
C++ adapter (mov-slam_amd/host/Optimizer.cc) used by tests/ and bench.py.

There is no CPU fallback: if the shared library is missing, or no HIP device is
visible, every entry point raises (MovbaError) instead of computing anything here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOVBA_LIB selects another build of the same library (diagnostic builds, e.g. -DMOVBA_CLOCK_STAMP)
LIB_PATH = os.environ.get("MOVBA_LIB") or os.path.join(os.path.dirname(_HERE), "libmovba.so")
# the TEST build of the same sources (-DMOVBA_TEST_HOOKS: per-handle switches through movba_test_hook; tests only)
HOOKS_LIB_PATH = os.path.join(os.path.dirname(_HERE), "libmovba_hooks.so")

MAX_TRACE = 128
NKERNELS = 6
OK, STOPPED, NO_FIXED, EMPTY, ERR_ARG, ERR_HIP, ERR_STATE, ERR_DEVICE_WAIT, ERR_TOO_LARGE = 0, 1, 2, 3, -1, -2, -3, -4, -5
FLAG_STALE_ERROR_QUIRK = 1

_d = C.POINTER(C.c_double)
_i = C.POINTER(C.c_int32)
_u = C.POINTER(C.c_uint8)


class MovbaError(RuntimeError):
    pass


class LbaDesc(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("poses", _d), ("pose_fixed", _u), ("points", _d), ("edge_pose", _i), ("edge_point", _i),
                ("obs", _d), ("inv_sigma2", _d),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber_delta", C.c_double), ("chi2_gate", C.c_double),
                ("max_iters", C.c_int32), ("max_trials", C.c_int32), ("flags", C.c_uint32), ("stop", _u),
                ("obs_right", _d), ("bf", C.c_double), ("cam_kf", _d), ("bf_kf", _d)]


class LbaResult(C.Structure):
    _fields_ = [("poses", _d), ("points", _d), ("chi2", _d), ("outlier", _u),
                ("status", C.c_int32), ("iters_done", C.c_int32), ("n_solves", C.c_int32),
                ("n_outliers", C.c_int32), ("pcg_iters", C.c_int32), ("last_rejected", C.c_int32),
                ("lambda_", C.c_double), ("cost0", C.c_double), ("cost", C.c_double),
                ("n_trace", C.c_int32),
                ("tr_lambda", C.c_double * MAX_TRACE), ("tr_f0", C.c_double * MAX_TRACE),
                ("tr_f1", C.c_double * MAX_TRACE), ("tr_rho", C.c_double * MAX_TRACE),
                ("tr_accept", C.c_int32 * MAX_TRACE), ("tr_pcg_iters", C.c_int32 * MAX_TRACE),
                ("n_direct", C.c_int32), ("direct_from", C.c_int32), ("n_chol_fail", C.c_int32), ("n_pcg_giveups", C.c_int32),
                ("n_sync_timeouts", C.c_int32), ("n_band", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("pcg_rel_tol", C.c_double), ("pcg_max_iters", C.c_int32), ("run_ahead", C.c_int32),
                ("profile", C.c_int32), ("pcg_coarse", C.c_int32), ("host_wait", C.c_int32), ("pcg_spill", C.c_int32), ("solver", C.c_int32), ("reorder", C.c_int32), ("pad_o", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("name", C.c_char_p * NKERNELS), ("ms", C.c_double * NKERNELS), ("launches", C.c_int64 * NKERNELS),
                ("upload_ms", C.c_double), ("structure_ms", C.c_double), ("download_ms", C.c_double)]


class StructureInfo(C.Structure):
    _fields_ = [("n_free", C.c_int32), ("n_pairs", C.c_int32), ("n_entries", C.c_int64), ("n_items", C.c_int32),
                ("max_degree", C.c_int32), ("already_grouped", C.c_int32), ("pcg_on_chip", C.c_int32),
                ("pcg_overflow", C.c_int32), ("pcg_max_wave_entries", C.c_int32), ("n_row_entries", C.c_int32),
                ("n_sched_slots", C.c_int32), ("sched_items", C.c_int32), ("sched_max_permille", C.c_int32), ("slots_ok", C.c_int32),
                ("reordered", C.c_int32), ("pad_s", C.c_int32)]


class PoseDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("Xw", _d), ("obs", _d), ("inv_sigma2", _d),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("pose0", C.c_double * 7), ("huber_delta", C.c_double), ("chi2_gate", C.c_double),
                ("rounds", C.c_int32), ("its_per_round", C.c_int32), ("ransac_iters", C.c_int32), ("ransac_seed", C.c_uint32),
                ("confidence", C.c_double), ("lo_iters", C.c_int32), ("pad_p", C.c_int32)]


class PoseResult(C.Structure):
    _fields_ = [("pose", C.c_double * 7), ("outlier", _u), ("chi2", _d), ("n_inliers", C.c_int32), ("status", C.c_int32),
                ("ransac_inliers", C.c_int32), ("lm_iters", C.c_int32), ("ransac_pose", C.c_double * 7),
                ("ransac_samples_used", C.c_int32), ("lo_accepted", C.c_int32), ("lo_inliers", C.c_int32), ("pad_q", C.c_int32)]


EXPORTS = ["movba_version", "movba_status_string", "movba_create", "movba_destroy", "movba_lba_solve",
           "movba_lba_upload", "movba_lba_reset", "movba_lba_run", "movba_lba_download",
           "movba_lba_export_poses_device", "movba_lba_set_pose_export", "movba_get_profile", "movba_reset_profile",
           "movba_structure_probe", "movba_pose_opt", "movba_set_profile_mask", "movba_lba_run_batch", "movba_pose_ransac_samples",
           "movba_host_alloc", "movba_host_free", "movba_dense_plan_probe"]

_libs = {False: None, True: None}


def _one_hip_runtime():
    """A process must hold ONE copy of the HIP runtime.  libmovba.so needs `libamdhip64.so.7` (found in /opt/rocm); torch's
    libtorch_hip.so needs `libamdhip64.so` (found in torch/lib by its rpath: another FILE with the same soname, 7).  Loaded
    first, torch's copy satisfies libmovba's soname; loaded second, it is mapped beside ROCm's copy, and the second runtime
    finds no device ("no ROCm-capable device is detected": profiles/r03p_hip_runtime_copies.log).  So when torch is installed
    its copy is mapped (RTLD_GLOBAL) before libmovba.so — without importing torch — and both bind to that one whichever is
    imported first; without torch (the C++ adapter inside MoV-SLAM) ROCm's own copy is the only one there is."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib(hooks: bool = False):
    """Load libmovba.so (hooks: the test build, libmovba_hooks.so); raise loudly when the HIP extension has not been built."""
    if _libs[hooks] is None:
        path = HOOKS_LIB_PATH if hooks else LIB_PATH
        if not os.path.exists(path):
            raise MovbaError(f"{path} not found: build it with `make -C mov-slam_amd/csrc` "
                             "(or __graft_entry__.build()); there is no CPU fallback")
        _one_hip_runtime()
        L = C.CDLL(path)
        if hooks:
            L.movba_test_hook.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
        L.movba_status_string.restype = C.c_char_p
        L.movba_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(Options)]
        L.movba_destroy.argtypes = [C.c_void_p]
        L.movba_destroy.restype = None
        for fn in ("movba_lba_upload",):
            getattr(L, fn).argtypes = [C.c_void_p, C.POINTER(LbaDesc)]
        L.movba_lba_solve.argtypes = [C.c_void_p, C.POINTER(LbaDesc), C.POINTER(LbaResult)]
        L.movba_lba_run.argtypes = [C.c_void_p]
        L.movba_lba_reset.argtypes = [C.c_void_p]
        L.movba_lba_download.argtypes = [C.c_void_p, C.POINTER(LbaResult)]
        L.movba_lba_export_poses_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.movba_lba_set_pose_export.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.movba_get_profile.argtypes = [C.c_void_p, C.POINTER(Profile)]
        L.movba_reset_profile.argtypes = [C.c_void_p]
        L.movba_set_profile_mask.argtypes = [C.c_void_p, C.c_int32]
        L.movba_structure_probe.argtypes = [C.POINTER(LbaDesc), C.POINTER(StructureInfo), _i, _i]
        L.movba_pose_opt.argtypes = [C.c_void_p, C.POINTER(PoseDesc), C.POINTER(PoseResult)]
        L.movba_lba_run_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
        L.movba_pose_ransac_samples.argtypes = [C.c_int32, C.c_int32, C.c_uint32, _i]
        L.movba_host_alloc.argtypes = [C.c_size_t]
        L.movba_host_alloc.restype = C.c_void_p
        L.movba_dense_plan_probe.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i, _i, C.c_int32, _i, C.c_int32]
        L.movba_host_free.argtypes = [C.c_void_p]
        L.movba_host_free.restype = None
        _libs[hooks] = L
    return _libs[hooks]


def status_string(s: int) -> str:
    return lib().movba_status_string(s).decode()


def _p(a, t):
    return a.ctypes.data_as(t)


def make_desc(w, flags=FLAG_STALE_ERROR_QUIRK, stop=None, max_iters=None, max_trials=0):
    """Flattened window (movba.synth.Window or anything with the same fields) -> movba_lba_desc."""
    keep = dict(
        poses=np.ascontiguousarray(w.poses, np.float64), fixed=np.ascontiguousarray(w.pose_fixed, np.uint8),
        points=np.ascontiguousarray(w.points, np.float64), ep=np.ascontiguousarray(w.edge_pose, np.int32),
        el=np.ascontiguousarray(w.edge_point, np.int32), obs=np.ascontiguousarray(w.obs, np.float64),
        isg=np.ascontiguousarray(w.inv_sigma2, np.float64))
    d = LbaDesc()
    d.n_poses, d.n_points, d.n_edges = len(keep["poses"]), len(keep["points"]), len(keep["ep"])
    d.poses = _p(keep["poses"], _d); d.pose_fixed = _p(keep["fixed"], _u); d.points = _p(keep["points"], _d)
    d.edge_pose = _p(keep["ep"], _i); d.edge_point = _p(keep["el"], _i)
    d.obs = _p(keep["obs"], _d); d.inv_sigma2 = _p(keep["isg"], _d)
    d.fx, d.fy, d.cx, d.cy = w.cam
    d.huber_delta, d.chi2_gate = w.huber_delta, w.chi2_gate
    d.max_iters = w.max_iters if max_iters is None else max_iters
    d.flags = flags
    d.max_trials = max_trials
    if stop is not None:
        keep["stop"] = stop
        d.stop = _p(stop, _u)
    if getattr(w, "obs_right", None) is not None:
        keep["obs_right"] = np.ascontiguousarray(w.obs_right, np.float64)
        d.obs_right = _p(keep["obs_right"], _d)
        d.bf = float(w.bf)
    if getattr(w, "cam_kf", None) is not None:           # intrinsics by keyframe (n_poses x 4)
        keep["cam_kf"] = np.ascontiguousarray(w.cam_kf, np.float64)
        assert keep["cam_kf"].shape == (d.n_poses, 4)
        d.cam_kf = _p(keep["cam_kf"], _d)
    if getattr(w, "bf_kf", None) is not None:
        keep["bf_kf"] = np.ascontiguousarray(w.bf_kf, np.float64)
        assert keep["bf_kf"].shape == (d.n_poses,)
        d.bf_kf = _p(keep["bf_kf"], _d)
    return d, keep


def structure_probe(w):
    d, keep = make_desc(w)
    info = StructureInfo()
    perm = np.zeros(d.n_edges, np.int32); fidx = np.zeros(d.n_poses, np.int32)
    rc = lib().movba_structure_probe(C.byref(d), C.byref(info), _p(perm, _i), _p(fidx, _i))
    if rc < 0:
        raise MovbaError(f"movba_structure_probe: {status_string(rc)}")
    return dict(status=rc, n_free=info.n_free, n_pairs=info.n_pairs, n_entries=info.n_entries, n_items=info.n_items,
                max_degree=info.max_degree, already_grouped=bool(info.already_grouped), pcg_on_chip=bool(info.pcg_on_chip),
                pcg_overflow=bool(info.pcg_overflow), pcg_max_wave_entries=info.pcg_max_wave_entries,
                n_row_entries=info.n_row_entries, n_sched_slots=info.n_sched_slots, sched_items=info.sched_items,
                sched_max_permille=info.sched_max_permille, slots_ok=bool(info.slots_ok), reordered=bool(info.reordered), perm=perm, free_index=fidx)


def dense_plan(nt: int, max_groups: int = 0, max_slots: int = 0):
    """Static schedule of the one-launch direct solver for nt block columns (host only): dict(ok, G, slots, task_ptr, tasks)
    with tasks as rows (op, slot, I, K, k, pad0, pad1)."""
    info = np.zeros(4, np.int32)
    lib().movba_dense_plan_probe(nt, max_groups, max_slots, _p(info, _i), None, 0, None, 0)
    if not info[0]:
        return dict(ok=False, G=0, slots=0, task_ptr=np.zeros(1, np.int32), tasks=np.zeros((0, 7), np.int32), tasks8=np.zeros((0, 8), np.int32))
    tp = np.zeros(info[1] + 1, np.int32); tk = np.zeros((info[3], 8), np.int32)
    lib().movba_dense_plan_probe(nt, max_groups, max_slots, _p(info, _i), _p(tp, _i), len(tp), _p(tk, _i), info[3])
    return dict(ok=True, G=int(info[1]), slots=int(info[2]), task_ptr=tp, tasks=tk[:, :7], tasks8=tk)


def ransac_samples(n: int, n_hyp: int, seed: int) -> np.ndarray:
    """The minimal samples movba_pose_opt draws (host only): (n_hyp, 3) match indices."""
    out = np.zeros((n_hyp, 3), np.int32)
    rc = lib().movba_pose_ransac_samples(n, n_hyp, seed, _p(out, _i))
    if rc != OK:
        raise MovbaError(f"movba_pose_ransac_samples: {status_string(rc)}")
    return out


def run_batch(solvers) -> int:
    """movba_lba_run_batch over the resident windows of `solvers` (created on one stream); download each as usual."""
    arr = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
    rc = solvers[0]._L.movba_lba_run_batch(arr, len(solvers))
    if rc < 0:
        raise MovbaError(f"movba_lba_run_batch: {status_string(rc)}")
    return rc


class Solver:
    """One handle = one device + one stream (movba_create / movba_destroy)."""

    def __init__(self, device: int = 0, stream: int | None = None, pcg_rel_tol: float = 0.0,
                 pcg_max_iters: int = 0, run_ahead: int = 0, profile=False, pcg_coarse: bool = True, host_wait: int = 0,
                 pcg_spill: bool = False, direct: bool = False, reorder: bool = True, solver: int | None = None, hooks: bool = False):
        self._h = C.c_void_p()
        self._L = lib(hooks)                # (hooks: the test build, whose handles take movba_test_hook)
        self._hooks = hooks
        self._pinned_blocks = []
        opt = Options(pcg_rel_tol, pcg_max_iters, run_ahead, (0x3f if profile is True else int(profile)), 0 if pcg_coarse else -1, host_wait, 1 if pcg_spill else 0, (int(solver) if solver is not None else (1 if direct else 0)), 0 if reorder else -1, 0)
        rc = self._L.movba_create(C.byref(self._h), device, C.c_void_p(stream) if stream else None, C.byref(opt))
        if rc != OK:
            self._h = C.c_void_p()
            raise MovbaError(f"movba_create failed: {status_string(rc)} (no CPU fallback)")
        self._keep = None

    def hook(self, name: str, value: int):
        """movba_test_hook (test build only: Solver(hooks=True)): host_structure, entries_unpacked, no_sorted_structure,
        pcg_packed, helper_delay_us, wait_ticks, band_park_trial, device_cus"""
        if not self._hooks:
            raise MovbaError("test hooks exist in libmovba_hooks.so only: Solver(hooks=True)")
        rc = self._L.movba_test_hook(self._h, name.encode(), int(value))
        if rc != OK:
            raise MovbaError(f"movba_test_hook({name}): {status_string(rc)}")

    def close(self):
        if self._h:
            self._L.movba_destroy(self._h)
            self._h = C.c_void_p()
        if getattr(self, "_pinned_blocks", None):
            self._prep = None
            for p in self._pinned_blocks:
                self._L.movba_host_free(p)
            self._pinned_blocks = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- results ---------------------------------------------------------------
    def _pinned(self, shape, dtype=np.float64):
        """array in movba_host_alloc memory (released by close()); float64 unless said otherwise"""
        n = int(np.prod(shape))
        item = np.dtype(dtype).itemsize
        p = self._L.movba_host_alloc(max(item * n, 8))
        if not p:
            raise MovbaError("movba_host_alloc failed")
        self._pinned_blocks.append(p)
        a = np.frombuffer((C.c_uint8 * max(item * n, 8)).from_address(p), dtype=dtype, count=n).reshape(shape)
        a[...] = 0
        return a

    def _pin_inputs(self, d, keep):
        """The descriptor's input arrays moved into movba_host_alloc memory (what a C++ caller that flattens its window into
        buffers of the library's allocator hands over: the adapter's): the device reads the index arrays where they lie and the
        copy engine takes the observations and estimates straight out of them, nothing is staged."""
        for key, field, ptr in (("poses", "poses", _d), ("fixed", "pose_fixed", _u), ("points", "points", _d), ("ep", "edge_pose", _i),
                                ("el", "edge_point", _i), ("obs", "obs", _d), ("isg", "inv_sigma2", _d), ("obs_right", "obs_right", _d)):
            if key not in keep:
                continue
            a = self._pinned(keep[key].shape, keep[key].dtype)
            a[...] = keep[key]
            keep[key] = a
            setattr(d, field, _p(a, ptr))

    def _alloc_result(self, d, pinned=False, chi2=True):
        mk = self._pinned if pinned else np.zeros
        out = dict(poses=mk((d.n_poses, 7)), points=mk((d.n_points, 3)),
                   chi2=mk((d.n_edges,)) if chi2 else np.zeros(0), outlier=np.zeros(d.n_edges, np.uint8))
        r = LbaResult()
        r.poses = _p(out["poses"], _d); r.points = _p(out["points"], _d)
        if chi2:
            r.chi2 = _p(out["chi2"], _d)                  # (left NULL: the per-edge chi2 is neither exported nor copied out)
        r.outlier = _p(out["outlier"], _u)
        return r, out

    @staticmethod
    def _pack(r, out, rc):
        n = r.n_trace
        out.update(status=rc, iters_done=r.iters_done, n_solves=r.n_solves, n_outliers=r.n_outliers,
                   pcg_iters=r.pcg_iters, last_rejected=r.last_rejected, lam=r.lambda_, cost0=r.cost0, cost=r.cost,
                   n_direct=r.n_direct, direct_from=r.direct_from, n_chol_fail=r.n_chol_fail, n_pcg_giveups=r.n_pcg_giveups,
                   n_sync_timeouts=r.n_sync_timeouts, n_band=r.n_band,
                   trace=dict(lam=np.array(r.tr_lambda[:n]), f0=np.array(r.tr_f0[:n]), f1=np.array(r.tr_f1[:n]),
                              rho=np.array(r.tr_rho[:n]), accept=np.array(r.tr_accept[:n]),
                              pcg=np.array(r.tr_pcg_iters[:n])))
        return out

    def solve(self, w, flags=FLAG_STALE_ERROR_QUIRK, stop=None, max_iters=None, max_trials=0) -> dict:
        d, keep = make_desc(w, flags, stop, max_iters, max_trials)
        r, out = self._alloc_result(d)
        rc = self._L.movba_lba_solve(self._h, C.byref(d), C.byref(r))
        if rc < 0:
            raise MovbaError(f"movba_lba_solve: {status_string(rc)}")
        self._keep = (d, keep)             # (download() may be called again on the solved window)
        if rc != OK:                       # silent early return: nothing written, echo the inputs
            out["poses"][:] = keep["poses"]; out["points"][:] = keep["points"]
        return self._pack(r, out, rc)

    def prepare(self, w, flags=FLAG_STALE_ERROR_QUIRK, stop=None, max_iters=None, max_trials=0, pinned=False, chi2=True):
        """Descriptor and result buffers built once for repeated solve_prepared() calls: what a C++ caller that keeps its
        flattened arrays and result buffers does (nothing is allocated or converted per call).  pinned=True: the buffers are
        blocks of movba_host_alloc memory that live until close() (results handed out earlier stay valid): prepare a window
        once and keep it - `self._prep` may be saved and put back to alternate between prepared windows - rather than
        preparing it again for every call."""
        d, keep = make_desc(w, flags, stop, max_iters, max_trials)
        if pinned:
            self._pin_inputs(d, keep)                     # ... and input arrays the device reads where they lie
        r, out = self._alloc_result(d, pinned, chi2)      # pinned: result arrays the solve's last kernel writes into directly
        self._keep = (d, keep)
        self._prep = (d, keep, r, out)

    def solve_prepared(self, pack=True):
        """movba_lba_solve on the prepared buffers; pack=False returns only the status (results stay in the buffers)."""
        d, keep, r, out = self._prep
        rc = self._L.movba_lba_solve(self._h, C.byref(d), C.byref(r))
        if rc < 0:
            raise MovbaError(f"movba_lba_solve: {status_string(rc)}")
        if not pack:
            return rc
        if rc != OK:
            out["poses"][:] = keep["poses"]; out["points"][:] = keep["points"]
        return self._pack(r, dict(out), rc)

    def upload_prepared(self):
        """movba_lba_upload of the prepared descriptor (phased calls on buffers built once: batched runs)"""
        d, keep, r, out = self._prep
        rc = self._L.movba_lba_upload(self._h, C.byref(d))
        if rc < 0:
            raise MovbaError(f"movba_lba_upload: {status_string(rc)}")
        self._keep = (d, keep)
        return rc

    def download_prepared(self, pack=False):
        """movba_lba_download into the prepared result buffers"""
        d, keep, r, out = self._prep
        rc = self._L.movba_lba_download(self._h, C.byref(r))
        if rc < 0:
            raise MovbaError(f"movba_lba_download: {status_string(rc)}")
        return self._pack(r, dict(out), rc) if pack else rc

    def upload(self, w, flags=FLAG_STALE_ERROR_QUIRK, stop=None, max_iters=None, max_trials=0):
        d, keep = make_desc(w, flags, stop, max_iters, max_trials)
        rc = self._L.movba_lba_upload(self._h, C.byref(d))
        if rc < 0:
            raise MovbaError(f"movba_lba_upload: {status_string(rc)}")
        self._keep = (d, keep)
        return rc

    def run(self) -> int:
        rc = self._L.movba_lba_run(self._h)
        if rc < 0:
            raise MovbaError(f"movba_lba_run: {status_string(rc)}")
        return rc

    def download(self) -> dict:
        d, keep = self._keep
        r, out = self._alloc_result(d)
        rc = self._L.movba_lba_download(self._h, C.byref(r))
        if rc < 0:
            raise MovbaError(f"movba_lba_download: {status_string(rc)}")
        if rc != OK:
            out["poses"][:] = keep["poses"]; out["points"][:] = keep["points"]
        return self._pack(r, out, rc)

    def export_poses_device(self, dst_ptr: int, nbytes: int):
        rc = self._L.movba_lba_export_poses_device(self._h, C.c_void_p(dst_ptr), nbytes)
        if rc != OK:
            raise MovbaError(f"movba_lba_export_poses_device: {status_string(rc)}")

    def set_pose_export(self, dst_ptr: int, nbytes: int):
        """Register a device buffer that every later run() leaves the optimised poses in (0 unregisters)."""
        rc = self._L.movba_lba_set_pose_export(self._h, C.c_void_p(dst_ptr) if dst_ptr else None, nbytes)
        if rc != OK:
            raise MovbaError(f"movba_lba_set_pose_export: {status_string(rc)}")

    def profile(self) -> dict:
        p = Profile()
        self._L.movba_get_profile(self._h, C.byref(p))
        return dict(kernels={p.name[k].decode(): dict(ms=p.ms[k], launches=p.launches[k]) for k in range(NKERNELS)},
                    upload_ms=p.upload_ms, structure_ms=p.structure_ms, download_ms=p.download_ms)

    def reset_profile(self):
        self._L.movba_reset_profile(self._h)

    def set_profile_mask(self, mask: int):
        self._L.movba_set_profile_mask(self._h, mask)

    def pose_opt(self, Xw, obs, pose0, cam, huber_delta, chi2_gate, rounds=4, its=10, inv_sigma2=None, ransac_iters=0, ransac_seed=1,
                 confidence=0.0, lo_iters=0) -> dict:
        Xw = np.ascontiguousarray(Xw, np.float64); obs = np.ascontiguousarray(obs, np.float64)
        n = len(Xw)
        d = PoseDesc()
        d.n = n; d.Xw = _p(Xw, _d); d.obs = _p(obs, _d)
        isg = None
        if inv_sigma2 is not None:
            isg = np.ascontiguousarray(inv_sigma2, np.float64); d.inv_sigma2 = _p(isg, _d)
        d.fx, d.fy, d.cx, d.cy = cam
        d.pose0 = (C.c_double * 7)(*pose0)
        d.huber_delta, d.chi2_gate, d.rounds, d.its_per_round = huber_delta, chi2_gate, rounds, its
        d.ransac_iters, d.ransac_seed = ransac_iters, ransac_seed
        d.confidence, d.lo_iters = confidence, lo_iters
        outl = np.zeros(n, np.uint8); chi2 = np.zeros(n)
        r = PoseResult(); r.outlier = _p(outl, _u); r.chi2 = _p(chi2, _d)
        rc = self._L.movba_pose_opt(self._h, C.byref(d), C.byref(r))
        if rc < 0:
            raise MovbaError(f"movba_pose_opt: {status_string(rc)}")
        return dict(status=rc, n_inliers=r.n_inliers, pose=np.array(r.pose[:]), outlier=outl, chi2=chi2,
                    ransac_inliers=r.ransac_inliers, ransac_pose=np.array(r.ransac_pose[:]), lm_iters=r.lm_iters,
                    ransac_samples_used=r.ransac_samples_used, lo_accepted=r.lo_accepted, lo_inliers=r.lo_inliers)
