import sys, time, os
sys.path.insert(0,'mov-slam_amd'); sys.path.insert(0,'.')
from movba import synth
from oracle import oracle
w=synth.cfg('cfg3')
oracle.solve(w, omp=True)
ts=[]
for _ in range(3):
    t=time.time(); oracle.solve(w, omp=True); ts.append(time.time()-t)
print(os.environ.get('OMP_NUM_THREADS'), os.environ.get('OMP_WAIT_POLICY'), 'affinity', len(os.sched_getaffinity(0)), 'omp ms', [round(x*1e3) for x in ts])
