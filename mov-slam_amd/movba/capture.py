"""Window capture / replay files.

The C++ adapter (mov-slam_amd/host/Optimizer.cc) writes every flattened LocalBundleAdjustment window it
solves to $MOVBA_DUMP_DIR/lba_<n>.mbw when that variable is set, so that windows recorded inside a real
MoV-SLAM run can be replayed on a GPU box (tests/dev/replay_windows.py) without the reference's stack.
Layout (little endian): magic 'MOVBAWIN', u32 version=1, i32 NP, P, E, max_iters, u32 flags,
f64 cam[4], huber_delta, chi2_gate, then pose_fixed u8[NP] (padded to 8), poses f64[NP*7],
points f64[P*3], edge_pose i32[E], edge_point i32[E], obs f64[E*2], inv_sigma2 f64[E]; when the top bit of
flags is set a stereo trailer follows: f64 bf, obs_right f64[E] (< 0 = monocular edge).
"""
from __future__ import annotations

import struct

import numpy as np

from . import synth

MAGIC = b"MOVBAWIN"


def save_window(path: str, w, flags: int = 1):
    with open(path, "wb") as f:
        f.write(MAGIC)
        stereo = getattr(w, "obs_right", None) is not None
        f.write(struct.pack("<I4iI", 1, w.n_poses, w.n_points, w.n_edges, w.max_iters, flags | (0x80000000 if stereo else 0)))
        f.write(struct.pack("<6d", *w.cam, w.huber_delta, w.chi2_gate))
        fixed = np.ascontiguousarray(w.pose_fixed, np.uint8).tobytes()
        f.write(fixed + b"\0" * (-len(fixed) % 8))
        for arr, dt in ((w.poses, np.float64), (w.points, np.float64), (w.edge_pose, np.int32), (w.edge_point, np.int32),
                        (w.obs, np.float64), (w.inv_sigma2, np.float64)):
            f.write(np.ascontiguousarray(arr, dt).tobytes())
        if stereo:
            f.write(struct.pack("<d", w.bf)); f.write(np.ascontiguousarray(w.obs_right, np.float64).tobytes())


def load_window(path: str):
    b = open(path, "rb").read()
    if b[:8] != MAGIC:
        raise ValueError(f"{path}: not a MOVBAWIN file")
    ver, NP, P, E, max_iters, flags = struct.unpack_from("<I4iI", b, 8)
    if ver != 1:
        raise ValueError(f"{path}: unsupported version {ver}")
    off = 8 + 24
    cam = struct.unpack_from("<4d", b, off); huber, gate = struct.unpack_from("<2d", b, off + 32); off += 48
    fixed = np.frombuffer(b, np.uint8, NP, off).copy(); off += NP + (-NP % 8)

    def take(dt, n):
        nonlocal off
        a = np.frombuffer(b, dt, n, off).copy(); off += a.nbytes; return a
    poses = take(np.float64, 7 * NP).reshape(NP, 7); points = take(np.float64, 3 * P).reshape(P, 3)
    ep = take(np.int32, E); el = take(np.int32, E)
    obs = take(np.float64, 2 * E).reshape(E, 2); isg = take(np.float64, E)
    w = synth.Window(poses=poses, pose_fixed=fixed, points=points, edge_pose=ep, edge_point=el, obs=obs, inv_sigma2=isg,
                     cam=tuple(cam), huber_delta=huber, chi2_gate=gate, max_iters=max_iters, meta=dict(flags=flags & 0x7fffffff))
    if flags & 0x80000000:
        w.bf = struct.unpack_from("<d", b, off)[0]; off += 8
        w.obs_right = take(np.float64, E)
    return w
