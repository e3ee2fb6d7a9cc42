"""Diagnostic (not a test): per-phase cycle shares of the Riccati kernel from a -DBMPC_PROFILE build
(BMPC_LIB must point at it)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from boundplanner_amd import solver, scenes
N, B = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
be = solver.HipBoundMPC(N)
batch = scenes.make_batch(B, N, 8192, be.fk, randomize_sets=True)
r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
buf = (ctypes.c_double * 16)()
be.lib.bmpc_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
be.lib.bmpc_debug_phase_cycles(be._h, buf)
names = ['load+scatter', 'lam-curv', 'T', 'couple C3 + gradients', 'adj+chol+gains', 'schur', 'forward(all stages)', 'couple C1 (Et, Y, vt0)', 'couple C2 (structured)']
v = np.array(list(buf)); its = (r["iters"].sum() + B) * (N - 1)
print("kernel ms", be.last_kernel_ms(), "iters mean", r["iters"].mean())
for n, x in zip(names, v):
    print(f"{n:22s} {100*x/v.sum():5.1f}%   {x/its:9.0f} cycles/stage-sweep")
print("total", v.sum() / its)
