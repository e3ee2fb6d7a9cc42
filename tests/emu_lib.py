"""ctypes access to the CPU thread emulation of the HIP solver (tests/emu) -- debugging aid,
TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "emu_main.cpp")
LIB = os.path.join(ROOT, "tests", "emu", "libbmpc_emu.so")
_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    deps = [SRC] + [os.path.join(ROOT, "boundplanner_amd", "csrc", f) for f in
                    ("bmpc_device.hpp", "bmpc_solver.hpp", "bmpc_robot.hpp")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-o", LIB, SRC])


def solve(N, x0, lbx, ubx, p, dt=0.1, tol=1e-5, max_iter=100, hess=0, hess_switch=0.1, mu_init=0.1,
          kappa_mu=0.1, theta_mu=2.0, kappa_eps=1000.0):
    build()
    lib = ctypes.CDLL(LIB)
    n_w = 44 * N + 6
    lbx = np.where(np.isinf(lbx), -1e20, lbx); ubx = np.where(np.isinf(ubx), 1e20, ubx)
    x0, lbx, ubx, p = (np.ascontiguousarray(a, float) for a in (x0, lbx, ubx, p))
    x = np.zeros(n_w)
    f = ctypes.c_double(); it = ctypes.c_int(); st = ctypes.c_int(); viol = ctypes.c_double()
    P = lambda a: a.ctypes.data_as(_dp)
    D = ctypes.c_double
    rc = lib.emu_solve(N, D(dt), D(tol), max_iter, hess, D(hess_switch), D(mu_init), D(kappa_mu), D(theta_mu),
                       D(kappa_eps), P(x0), P(lbx), P(ubx), P(p), P(x), ctypes.byref(f), ctypes.byref(it),
                       ctypes.byref(st), ctypes.byref(viol))
    assert rc == 0
    return dict(x=x, f=f.value, iters=it.value, status=st.value, viol=viol.value)
