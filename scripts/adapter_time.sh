#!/bin/bash
# host-side phases of Optimizer::LocalBundleAdjustment over the mock map classes (cfg3), fourth call of the process on a fresh
# copy of the map (run through gpurun from the repo root; LAPS=1 TAILN=10 for the laps inside extraction and write-back)
python - <<'PY'
import sys, struct, os
import numpy as np
sys.path.insert(0, "mov-slam_amd")
from movba import synth
w = synth.cfg("cfg3")
with open("/tmp/w3.bin", "wb") as fh:
    fh.write(struct.pack("4i", w.n_poses, w.n_points, w.n_edges, 0))
    for arr, dt in ((w.pose_fixed, np.uint8), (w.poses, np.float64), (w.points, np.float64), (w.edge_pose, np.int32), (w.edge_point, np.int32), (w.obs, np.float64)):
        fh.write(np.ascontiguousarray(arr, dt).tobytes())
PY
make -C mov-slam_amd/host -s
MOVBA_ADAPTER_REPS=4 MOVBA_ADAPTER_LAPS=${LAPS:-} MOVBA_ADAPTER_TIMING=1 mov-slam_amd/host/adapter_test lba /tmp/w3.bin /tmp/o3.bin 2>&1 | grep "adapter" | tail -${TAILN:-2}
echo "with MapPoint::ForEachObservation (-DMOVBA_MAPPOINT_HAS_FOR_EACH_OBSERVATION):"
MOVBA_ADAPTER_REPS=4 MOVBA_ADAPTER_LAPS=${LAPS:-} MOVBA_ADAPTER_TIMING=1 mov-slam_amd/host/adapter_test_foreach lba /tmp/w3.bin /tmp/o3f.bin 2>&1 | grep "adapter" | tail -${TAILN:-2}
