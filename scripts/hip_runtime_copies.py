"""Which copies of the HIP runtime a process maps when libmovba.so and torch are both loaded, in either order (the
solver fixture's torch.cuda.init() workaround blamed "torch's own copy of the HIP runtime").  Run on a GPU box; prints the
mapped libamdhip64 paths and whether torch sees the device after libmovba has used it."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "mov-slam_amd"))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "libhsa-runtime" in l})
if order == "movba_first":
    from movba import capi, synth
    s = capi.Solver(); r = s.solve(synth.cfg("small")); print("movba solved", r["n_solves"]); print("maps after movba:", maps())
    import torch
    print("torch.cuda.is_available:", torch.cuda.is_available())
    st = torch.cuda.Stream(); print("torch stream ok", st.cuda_stream != 0)
    s2 = capi.Solver(stream=st.cuda_stream); print("movba on torch stream:", s2.solve(synth.cfg("small"))["n_solves"])
else:
    import torch
    print("torch.cuda.is_available:", torch.cuda.is_available()); st = torch.cuda.Stream()
    from movba import capi, synth
    s = capi.Solver(stream=st.cuda_stream); print("movba solved", s.solve(synth.cfg("small"))["n_solves"])
print("maps at end:", maps())
''' % ROOT
for order in ("movba_first", "torch_first"):
    p = subprocess.run([sys.executable, "-c", child, order], capture_output=True, text=True)
    print("==", order, "rc", p.returncode); print(p.stdout); print(p.stderr[-1500:])
