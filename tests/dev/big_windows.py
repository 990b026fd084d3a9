import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(ROOT, 'mov-slam_amd')); sys.path.insert(0, ROOT)
import numpy as np
from movba import synth, capi
from oracle import oracle
for (K,F,P,hi) in ((80,10,40000,10),(64,8,60000,12),(30,5,30000,20)):
    w = synth.make_window(K,F,P,seed=77,run_lo=2,run_hi=hi)
    t=time.time(); ro=oracle.solve(w); tc=time.time()-t
    s=capi.Solver(); rg=s.solve(w); s.upload(w)
    ts=[]
    for _ in range(5):
        t=time.perf_counter(); s.run(); ts.append(time.perf_counter()-t)
    print(f"K={K} F={F} P={P} E={w.n_edges}: cpu {tc*1e3:.0f} ms gpu run {min(ts)*1e3:.3f} ms pcg {rg['pcg_iters']} solves {ro['n_solves']}/{rg['n_solves']} dq {np.abs(ro['poses'][:,:4]-rg['poses'][:,:4]).max():.2e} dt {np.abs(ro['poses'][:,4:]-rg['poses'][:,4:]).max():.2e} outl {int((ro['outlier']!=rg['outlier']).sum())} probe {capi.structure_probe(w)['pcg_overflow']}")
