"""Diagnostic: does the -DBMPC_PROFILE build give the same answers as the product build?"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from boundplanner_amd import solver, scenes
N, B = 20, 256
be = solver.HipBoundMPC(N)
batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
r1 = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
r1b = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
print("product build: iters mean", r1["iters"].mean(), "rerun identical", np.array_equal(r1["x"], r1b["x"]))
import ctypes
lib2 = ctypes.CDLL(os.path.join(ROOT, "gpurun_out", "libboundmpc_prof.so"))
solver._lib = None
solver.LIB_PATH = os.path.join(ROOT, "gpurun_out", "libboundmpc_prof.so")
be2 = solver.HipBoundMPC(N)
r2 = be2.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
r2b = be2.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
print("profile build: iters mean", r2["iters"].mean(), "rerun identical", np.array_equal(r2["x"], r2b["x"]))
print("max |dx| between builds", np.abs(r1["x"] - r2["x"]).max(), "iters differ on", (r1["iters"] != r2["iters"]).sum())
