"""Diagnostic: which instrumentation point perturbs the results?"""
import os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "boundplanner_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
FL = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
base = []
for nt in (128, 256):
    o = os.path.join(OUT, f"kb_{nt}.o")
    subprocess.check_call(["hipcc", *FL, f"-DBMPC_NT={nt}", "-c", os.path.join(CS, "bmpc_kernels.hip"), "-o", o]); base.append(o)
oc = os.path.join(OUT, "cb.o")
subprocess.check_call(["hipcc", *FL, "-c", os.path.join(CS, "bmpc_capi.hip"), "-o", oc])
from boundplanner_amd import solver, scenes
N, B = 20, 256
batch = None
for bit in [None, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13]:
    defs = [] if bit is None else ["-DBMPC_PROFILE", f"-DBMPC_PROF_MASK={1 << bit}"]
    o = os.path.join(OUT, f"kv_{bit}.o"); so = os.path.join(OUT, f"libv_{bit}.so")
    subprocess.check_call(["hipcc", *FL, *defs, "-DBMPC_NT=64", "-c", os.path.join(CS, "bmpc_kernels.hip"), "-o", o])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, oc, o, *base])
    solver._lib = None; solver.LIB_PATH = so
    be = solver.HipBoundMPC(N)
    if batch is None: batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    print("point", bit, "iters mean", r["iters"].mean(), flush=True)
