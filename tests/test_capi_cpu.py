"""C-ABI surface and host-side logic that need no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from movba import synth


def test_library_exports_every_symbol_declared_in_header(built_lib):
    hdr = open(os.path.join(ROOT, "include", "movba.h")).read()
    declared = sorted(set(re.findall(r"\b(movba_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 14
    lib = built_lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(built_lib.EXPORTS) == declared
    assert lib.movba_version() == 5
    # the product library has no test hooks; the test build of the same sources has the same C-ABI plus movba_test_hook
    assert not hasattr(lib, "movba_test_hook")
    hooks = built_lib.lib(hooks=True)
    for name in declared:
        assert hasattr(hooks, name), name
    assert hasattr(hooks, "movba_test_hook") and hooks.movba_version() == 5
    assert built_lib.status_string(0) == "ok" and built_lib.status_string(-2) == "HIP runtime error"


def test_product_library_reads_only_its_documented_process_switches(built_lib):
    """Every environment variable the product library knows is one of the diagnostic switches api.cpp documents, read once per
    process; nothing a test switches lives in it (those are hooks of libmovba_hooks.so)."""
    blob = open(built_lib.LIB_PATH, "rb").read()
    names = set(m.decode() for m in re.findall(rb"MOVBA_[A-Z_0-9]{3,}", blob))
    allowed = {"MOVBA_WATCHDOG_MS", "MOVBA_TIME_UPLOAD", "MOVBA_TIME_SOLVE", "MOVBA_DENSE_STAMPS", "MOVBA_DENSE_MULTILAUNCH", "MOVBA_BAND", "MOVBA_BATCH_GROUPS"}
    assert names <= allowed, names - allowed
    for hook in (b"band_park_trial", b"helper_delay_us", b"host_structure", b"movba_test_hook"):
        assert hook not in blob and hook in open(built_lib.HOOKS_LIB_PATH, "rb").read()
    src = open(os.path.join(ROOT, "mov-slam_amd", "csrc", "api.cpp")).read()
    assert src.count("std::getenv(") == len(allowed)          # all of them inside process_switches()


def test_ctypes_struct_sizes_match_header_layout(built_lib):
    # int32 x3 (+pad) | 7 pointers | 6 doubles | 2 int32 + u32 (+pad) | pointer
    assert ctypes.sizeof(built_lib.LbaDesc) == 16 + 7 * 8 + 6 * 8 + 16 + 8 + 16 + 16
    assert ctypes.sizeof(built_lib.Options) == 48
    assert ctypes.sizeof(built_lib.StructureInfo) == 72


def _np_structure(w):
    free = (w.pose_fixed == 0)
    active = np.zeros(w.n_poses, bool); active[w.edge_pose] = True
    hidx = -np.ones(w.n_poses, int); idx = np.flatnonzero(free & active); hidx[idx] = np.arange(len(idx))
    d_free = np.bincount(w.edge_point[hidx[w.edge_pose] >= 0], minlength=w.n_points)
    n_entries = int((d_free * (d_free + 1) // 2).sum())
    pairs = set()
    order = np.argsort(w.edge_point, kind="stable")
    for l in np.unique(w.edge_point):
        hs = sorted(h for h in hidx[w.edge_pose[w.edge_point == l]] if h >= 0)
        for a in range(len(hs)):
            for b in range(a, len(hs)):
                pairs.add((hs[a], hs[b]))
    pairs |= {(i, i) for i in range(len(idx))}
    return hidx, n_entries, len(pairs), order


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_structure_probe_matches_numpy(built_lib, name):
    w = synth.cfg(name)
    s = built_lib.structure_probe(w)
    hidx, n_entries, n_pairs, order = _np_structure(w)
    assert s["status"] == 0 and s["already_grouped"]
    assert np.array_equal(s["free_index"], hidx)
    assert s["n_free"] == (hidx >= 0).sum() and s["n_entries"] == n_entries and s["n_pairs"] == n_pairs
    assert np.array_equal(s["perm"], np.arange(w.n_edges))
    assert s["max_degree"] == np.bincount(w.edge_point).max()


@pytest.mark.parametrize("shape", [(3, 1, 20, 2, 3), (10, 2, 2000, 2, 6), (50, 10, 20000, 2, 10), (6, 2, 6000, 8, 8), (40, 4, 3000, 10, 30)])
def test_schur_launch_schedule_and_pose_major_slots(built_lib, shape):
    """Host plan of the schur pass: every work item sits in exactly one slot of the 8 XCD segments, the segments carry
    about the same estimated work, and the pose-major edge slots are a bijection onto the free keyframes' edges."""
    K, F, P, lo, hi = shape
    w = synth.make_window(K, F, P, seed=5, run_lo=lo, run_hi=hi)
    s = built_lib.structure_probe(w)
    assert s["status"] == 0 and s["slots_ok"]
    assert s["sched_items"] == s["n_items"] and s["n_sched_slots"] % 8 == 0 and s["n_sched_slots"] >= s["n_items"]
    if s["n_items"] >= 64:          # enough items to balance: the heaviest XCD segment is within 25 % of the mean
        assert s["sched_max_permille"] <= 1250, s["sched_max_permille"]
    # diagonal pairs are cut finer than off-diagonal ones, never below one item per pair
    assert s["n_items"] >= s["n_pairs"]


def _envelope(w, hidx):
    """sum over block rows of (row - first linked column) of the reduced matrix's pattern in the numbering hidx"""
    nf = int(hidx.max()) + 1
    A = np.zeros((nf, nf), bool)
    order = np.argsort(w.edge_point, kind="stable")
    ep, el = hidx[w.edge_pose[order]], w.edge_point[order]
    start = np.searchsorted(el, np.arange(w.n_points + 1))
    for l in range(w.n_points):
        hs = ep[start[l]:start[l + 1]]; hs = hs[hs >= 0]
        A[np.ix_(hs, hs)] = True
    return int(sum(r - int(np.argmax(A[r, :r + 1])) for r in range(nf)))


def test_keyframes_are_renumbered_by_covisibility_when_their_ids_do_not_follow_the_graph(built_lib):
    """The reference numbers pose vertices by KeyFrame::mnId (Optimizer.cc:557-566).  A window numbered along its path keeps
    that numbering; the same graph with shuffled ids, or a path that comes back over itself, is renumbered (reverse
    Cuthill-McKee on the pair graph) so that covisible keyframes are neighbours again."""
    w = synth.make_window(24, 4, 1500, seed=3, run_lo=2, run_hi=6)
    s0 = built_lib.structure_probe(w)
    assert not s0["reordered"]
    env0 = _envelope(w, s0["free_index"])
    for ws in (synth.shuffle_ids(w, 5), synth.make_pattern_window("revisit", 24, 4, 1500, seed=3, run_lo=2, run_hi=6, revisit_gap=12)):
        s1 = built_lib.structure_probe(ws)
        f = s1["free_index"]
        assert s1["reordered"] and sorted(f[f >= 0]) == list(range(s1["n_free"])) and (f[ws.pose_fixed == 1] == -1).all()
        natural = -np.ones(ws.n_poses, int); idx = np.flatnonzero(ws.pose_fixed == 0); natural[idx] = np.arange(len(idx))
        assert _envelope(ws, f) <= 1.5 * env0 and _envelope(ws, f) < 0.6 * _envelope(ws, natural)
        assert s1["slots_ok"] and s1["sched_items"] == s1["n_items"]
    hub = synth.make_pattern_window("hub", 12, 2, 400, seed=4)
    assert not built_lib.structure_probe(hub)["reordered"]          # every pair linked: nothing to gain


def test_structure_probe_groups_shuffled_edges_stably(built_lib):
    w = synth.cfg("small")
    rng = np.random.default_rng(3)
    p = rng.permutation(w.n_edges)
    w.edge_pose, w.edge_point, w.obs, w.inv_sigma2 = w.edge_pose[p], w.edge_point[p], w.obs[p], w.inv_sigma2[p]
    s = built_lib.structure_probe(w)
    assert not s["already_grouped"]
    assert np.array_equal(s["perm"], np.argsort(w.edge_point, kind="stable"))


def test_structure_probe_inactive_free_pose_and_fixed_only_points(built_lib):
    w = synth.cfg("tiny")
    # append a free keyframe nobody observes: it has no hessian index (edge-less vertices are inactive)
    w.poses = np.vstack([w.poses, w.poses[-1]]); w.pose_fixed = np.append(w.pose_fixed, 0).astype(np.uint8)
    s = built_lib.structure_probe(w)
    assert s["free_index"][-1] == -1 and s["n_free"] == 2


def test_structure_probe_rejects_bad_indices_and_duplicates(built_lib):
    w = synth.cfg("tiny")
    bad = synth.cfg("tiny"); bad.edge_pose = bad.edge_pose.copy(); bad.edge_pose[0] = 99
    with pytest.raises(built_lib.MovbaError):
        built_lib.structure_probe(bad)
    dup = synth.cfg("tiny")
    free_edge = int(np.flatnonzero(dup.pose_fixed[dup.edge_pose] == 0)[0])
    for f in ("edge_pose", "edge_point", "obs", "inv_sigma2"):
        a = getattr(dup, f); setattr(dup, f, np.concatenate([a, a[free_edge:free_edge + 1]]))
    with pytest.raises(built_lib.MovbaError):       # same keyframe observing the same point twice
        built_lib.structure_probe(dup)
    empty = synth.cfg("tiny")
    for f in ("edge_pose", "edge_point", "obs", "inv_sigma2"):
        setattr(empty, f, getattr(empty, f)[:0])
    assert built_lib.structure_probe(empty)["status"] == built_lib.EMPTY


def test_windows_beyond_the_direct_solvers_capacity_are_refused_by_name(built_lib):
    """MOVBA_MAX_FREE_KEYFRAMES (2 700): past it the multi-launch direct solver's back substitution no longer fits its LDS;
    the window is refused with MOVBA_ERR_TOO_LARGE at the structure pass instead of failing in a launch."""
    K = 2701
    poses = np.zeros((K + 1, 7)); poses[:, 3] = 1.0
    fixed = np.zeros(K + 1, np.uint8); fixed[0] = 1
    ep = np.stack([np.zeros(K, np.int32), np.arange(1, K + 1, dtype=np.int32)], 1).reshape(-1)      # every point: the fixed keyframe + one free
    el = np.repeat(np.arange(K, dtype=np.int32), 2)
    w = synth.Window(poses=poses, pose_fixed=fixed, points=np.ones((K, 3)), edge_pose=ep, edge_point=el, obs=np.zeros((2 * K, 2)), inv_sigma2=np.ones(2 * K))
    with pytest.raises(built_lib.MovbaError, match="too large"):
        built_lib.structure_probe(w)
    assert built_lib.ERR_TOO_LARGE == -5 and "too large" in built_lib.status_string(-5)
    w2 = synth.Window(poses=poses[:K], pose_fixed=fixed[:K], points=np.ones((K - 1, 3)), edge_pose=ep[:2 * (K - 1)], edge_point=el[:2 * (K - 1)],
                      obs=np.zeros((2 * (K - 1), 2)), inv_sigma2=np.ones(2 * (K - 1)))
    assert built_lib.structure_probe(w2)["n_free"] == 2700


def test_create_fails_loudly_without_a_gpu(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built_lib.MovbaError, match="no CPU fallback"):
        built_lib.Solver()


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under mov-slam_amd/ may import, link or call it."""
    pkg = os.path.join(ROOT, "mov-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".cc", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "lba_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_graft_entry_build_runs_clean():
    """The driver's build check (__graft_entry__.build): every native piece compiles and the library matches the header."""
    import __graft_entry__ as g
    g.build()
