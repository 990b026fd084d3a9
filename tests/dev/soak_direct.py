"""Soak of the one-launch direct solver's hand-offs under uneven load (run on a GPU box): windows of 7, 10, 19 and 50 block
columns solved over and over on one handle while a second thread keeps the chip busy with batched PCG solves and a third
with another direct-solver window (their launches are chained by the gate).  Every solve must return the bits of the first
one, with no wait given up.   python tests/dev/soak_direct.py [seconds]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from movba import capi, synth, shard

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
stop = threading.Event()
errs = []


def noise_batch():
    st = torch.cuda.Stream(device=0)
    ws = [synth.make_window(50, 10, 20000, shard.window_seed(i), run_lo=2, run_hi=10) for i in range(4)]
    ss = [capi.Solver(device=0, stream=st.cuda_stream) for _ in ws]
    for s, w in zip(ss, ws): s.upload(w)
    while not stop.is_set():
        capi.run_batch(ss)
    for s in ss: s.close()


def noise_direct():
    w = synth.pattern_cfg("hub")
    s = capi.Solver()
    ref = s.solve(w)
    n = 0
    while not stop.is_set():
        r = s.solve(w); n += 1
        if r["n_sync_timeouts"] or not np.array_equal(r["poses"], ref["poses"]): errs.append("noise_direct changed")
    s.close()
    print("hub solves beside:", n, flush=True)


ts = [threading.Thread(target=noise_batch), threading.Thread(target=noise_direct)]
for t in ts: t.start()
cases = [("cfg3", synth.cfg("cfg3"), dict(direct=True)), ("80kf", synth.make_window(80, 10, 40000, 5, run_lo=2, run_hi=10), {}),
         ("150kf", synth.make_window(150, 6, 6000, 9, run_lo=2, run_hi=10), {}), ("400kf", synth.make_window(400, 8, 12000, 9, run_lo=2, run_hi=12), {})]
cases[3][1].max_iters = 3
t_end = time.time() + secs
for name, w, kw in cases:
    s = capi.Solver(**kw)
    ref = s.solve(w)
    assert ref["n_direct"] == ref["n_solves"]
    n, t1 = 0, time.time() + secs / len(cases)
    while time.time() < t1:
        r = s.solve(w); n += 1
        if r["status"] != 0 or r["n_sync_timeouts"]:
            errs.append(f"{name}: status {r['status']} timeouts {r['n_sync_timeouts']}"); break
        if not (np.array_equal(r["poses"], ref["poses"]) and np.array_equal(r["points"], ref["points"]) and np.array_equal(r["chi2"], ref["chi2"])):
            errs.append(f"{name}: solve {n} differs from the first"); break
    print(f"{name}: {n} solves x {ref['n_solves']} direct launches, identical", flush=True)
    s.close()
stop.set()
for t in ts: t.join()
print("SOAK FAILED: " + "; ".join(errs[:5]) if errs else "SOAK OK")
sys.exit(1 if errs else 0)
