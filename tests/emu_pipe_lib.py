"""ctypes access to the CPU thread emulation of the batch-synchronous HIP pipeline (tests/emu/emu_pipe.cpp)
-- debugging aid, TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "emu_pipe.cpp")
LIB = os.path.join(ROOT, "tests", "emu", "libbmpc_emupipe.so")
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False, lib=LIB, defs=()):
    cs = os.path.join(ROOT, "boundplanner_amd", "csrc")
    deps = [SRC] + [os.path.join(cs, f) for f in os.listdir(cs) if f.endswith(".hpp")]
    if force or not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", *defs,
                               "-o", lib, SRC])
    return lib


# build variants of the device source (kept behind build knobs; the product build uses the defaults)
VARIANTS = {"trial4": ("-DEMU_TRIAL_NW=4",)}       # k_trial as a workgroup of four wavefronts, one part of the row walk each


def solve_batch(N, x0, lbx, ubx, p, dt=0.1, tol=1e-5, max_iter=100, hess=2, hess_switch=1.0, mu_init=0.1,
                kappa_mu=0.1, theta_mu=2.0, kappa_eps=1000.0, want_g=False, verbose=0, want_lam=False, slots=0,
                mu_floor_k=1e4, dw0=1e-4, inertia_err=1e-2, inertia=2, stall_n=8, gn_backoff=2, slack_reset=1, ls_alpha_mem=0.0, trial_repeats=9,
                variant=None):
    if variant is None:
        lib = ctypes.CDLL(build())
    else:
        lib = ctypes.CDLL(build(lib=LIB.replace(".so", f"_{variant}.so"), defs=VARIANTS[variant]))
    n_w, n_g = 44 * N + 6, 147 * (N - 1) + 21
    lbx = np.where(np.isinf(lbx), -1e20, lbx); ubx = np.where(np.isinf(ubx), 1e20, ubx)
    x0, lbx, ubx, p = (np.ascontiguousarray(np.atleast_2d(a), float) for a in (x0, lbx, ubx, p))
    B = x0.shape[0]
    x = np.zeros((B, n_w)); g = np.zeros((B, n_g)) if want_g else None
    f = np.zeros(B); viol = np.zeros(B); it = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
    lam_g = np.full((B, n_g), np.nan) if want_lam else None
    lam_x = np.full((B, n_w), np.nan) if want_lam else None
    P = lambda a: a.ctypes.data_as(_dp) if a is not None else None
    D = ctypes.c_double
    steps = lib.emu_pipe_solve(N, D(dt), D(tol), max_iter, hess, D(hess_switch), D(mu_init), D(kappa_mu), D(theta_mu),
                               D(kappa_eps), B, P(x0), P(lbx), P(ubx), P(p), P(x), P(g), P(f),
                               it.ctypes.data_as(_ip), st.ctypes.data_as(_ip), P(viol), verbose, P(lam_g), P(lam_x), slots,
                               D(mu_floor_k), D(dw0), D(inertia_err), inertia, stall_n, gn_backoff, slack_reset, D(ls_alpha_mem), trial_repeats)
    return dict(x=x, g=g, f=f, iters=it, status=st, viol=viol, steps=steps, lam_g=lam_g, lam_x=lam_x)
