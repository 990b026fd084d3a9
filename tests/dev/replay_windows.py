#!/usr/bin/env python3
"""Replay captured LocalBundleAdjustment windows (*.mbw, written by the adapter under MOVBA_DUMP_DIR) on the GPU.

    python scripts/replay_windows.py <dir>

(tests/dev/replay_check.py additionally compares every window with the CPU oracle.)
"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import capi, capture

files = sorted(glob.glob(os.path.join(sys.argv[1], "*.mbw")))
s = capi.Solver()
tot = 0.0
for f in files:
    w = capture.load_window(f)
    t = time.perf_counter(); r = s.solve(w); dt = time.perf_counter() - t; tot += dt
    print(f"{os.path.basename(f)}: KF {w.n_poses} MP {w.n_points} E {w.n_edges}  {1e3*dt:7.2f} ms  trials {r['n_solves']}  outliers {r['n_outliers']}  status {r['status']}")
print(f"{len(files)} windows, {1e3*tot:.1f} ms total")
