#!/usr/bin/env python3
"""Replay captured windows (*.mbw) on the GPU and compare every one with the CPU oracle.

    python tests/dev/replay_check.py <dir>
"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
from movba import capi, capture
from oracle import oracle

s = capi.Solver()
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.mbw"))):
    w = capture.load_window(f)
    r, o = s.solve(w), oracle.solve(w)
    print(f"{os.path.basename(f)}: KF {w.n_poses} MP {w.n_points} E {w.n_edges}  |pose - oracle| {np.abs(r['poses'] - o['poses']).max():.2e}"
          f"  outlier mismatches {(r['outlier'] != o['outlier']).sum()}")
