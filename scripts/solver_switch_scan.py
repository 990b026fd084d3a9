"""Where the one-launch direct solver overtakes the on-chip PCG: cfg3-like windows at growing keyframe counts and track lengths (run_hi), the PCG forced
(pcg_spill) beside the direct solver (run on a GPU box).  Prints the gather entries of S against the 1 024 the PCG
workgroup holds in registers - the quantity movba_lba_upload's choice is made on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
from movba import capi, synth


def timed(s, w, n=5):
    s.solve(w)
    t = time.perf_counter()
    for _ in range(n):
        r = s.solve(w)
    return 1e3 * (time.perf_counter() - t) / n, r


pcg = capi.Solver(pcg_spill=True, profile=True); direct = capi.Solver(direct=True, profile=True)
for K, span in ((50, 10), (56, 10), (62, 10), (68, 10), (74, 10), (80, 10), (40, 14), (50, 14), (56, 14), (62, 14), (40, 20), (50, 20)):
    w = synth.make_window(K, 10, 400 * K, 4000 + K, run_lo=2, run_hi=span)
    info = capi.structure_probe(w)
    pcg.reset_profile(); direct.reset_profile()
    tp, rp = timed(pcg, w); td, rd = timed(direct, w)
    kp, kd = pcg.profile()["kernels"], direct.profile()["kernels"]
    print(f"K={K} run_hi={span} E={w.n_edges} row entries {info['n_row_entries']} (overflow {info['pcg_overflow']}) solves {rp['n_solves']}/{rd['n_solves']}  "
          f"pcg {tp:.3f} ms (solve-kernels {kp['k_pcg']['ms'] / 6:.3f})  direct {td:.3f} ms (solve-kernels {kd['k_pcg']['ms'] / 6:.3f})", flush=True)
