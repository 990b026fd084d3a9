"""movba_lba_upload alone, repeated (target of rocprofv3 --kernel-trace --stats for the structure-pass kernels; nothing is
solved, so diagnostic builds whose structure pass is incomplete can be timed safely)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd"))
from movba import synth, capi
w = synth.cfg(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
s = capi.Solver()
for i in range(20):
    try:
        s.upload(w)
    except capi.MovbaError as e:      # (a diagnostic build without its pair masks fails the consistency check: the kernels ran)
        pass
print("uploads done")
