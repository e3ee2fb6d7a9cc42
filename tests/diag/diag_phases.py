"""Diagnostic (not a test): build a -DBMPC_PROFILE library and print per-phase cycle shares."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "boundplanner_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "libboundmpc_prof.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
fl = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DBMPC_PROFILE"]
objs = []
for nt in (64, 128, 256):
    o = os.path.join(ROOT, "gpurun_out", f"k{nt}.o")
    subprocess.check_call(["hipcc", *fl, f"-DBMPC_NT={nt}", "-c", os.path.join(CS, "bmpc_kernels.hip"), "-o", o]); objs.append(o)
o = os.path.join(ROOT, "gpurun_out", "capi.o")
subprocess.check_call(["hipcc", *fl, "-c", os.path.join(CS, "bmpc_capi.hip"), "-o", o])
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, o, *objs])
from boundplanner_amd import solver, scenes
solver.LIB_PATH = out
N, B = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 768
be = solver.HipBoundMPC(N)
batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
buf = (ctypes.c_double * 16)()
be.lib.bmpc_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
be.lib.bmpc_debug_phase_cycles(be._h, buf)
names = ['B_eval', 'B_rows', 'B_group', 'B_chain', 'B_points', 'B_curv', 'B_direct', 'B_T', 'B_couple', 'B_adj', 'B_factor', '-', 'forward', 'trial', '-', '-']
v = np.array(list(buf)); tot = v.sum()
its = r["iters"].sum() * (N - 1)
print("kernel ms", be.last_kernel_ms(), "iters mean", r["iters"].mean())
for n, x in zip(names, v):
    if x > 0: print(f"{n:10s} {100*x/tot:5.1f}%   {x/its:9.0f} cycles/stage-iteration")
print("total cycles/stage-iteration", tot / its)
