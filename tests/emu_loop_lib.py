"""ctypes access to the CPU build of the closed-loop step logic (tests/emu/emu_loop.cpp) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "emu_loop.cpp")
LIB = os.path.join(ROOT, "tests", "emu", "libbmpc_emuloop.so")
_dp = ctypes.POINTER(ctypes.c_double)
P = lambda a: a.ctypes.data_as(_dp) if a is not None else None
_lib = None


def lib():
    global _lib
    if _lib is None:
        cs = os.path.join(ROOT, "boundplanner_amd", "csrc")
        deps = [SRC] + [os.path.join(cs, f) for f in ("bmpc_loop.hpp", "bmpc_device.hpp", "bmpc_robot.hpp")]
        if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
            subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", LIB, SRC])
        _lib = ctypes.CDLL(LIB)
    return _lib


def layout():
    from boundplanner_amd.device_loop import read_layout
    L = lib()
    return read_layout(L.emu_loop_field, L.emu_loop_state_doubles)


def prepare(N, S, prev):
    n_w = 44 * N + 6
    x0, lbx, ubx, p = np.zeros(n_w), np.zeros(n_w), np.zeros(n_w), np.zeros(875)
    lib().emu_loop_prepare(N, P(S), P(prev), P(x0), P(lbx), P(ubx), P(p))
    return x0, lbx, ubx, p


def prepare_obs(N, S, prev, obs_sets, obs_points_sets):
    from boundplanner_amd.device_loop import pack_obstacles
    A, b, nrows, V, nv = pack_obstacles(obs_sets, obs_points_sets)
    ip = ctypes.POINTER(ctypes.c_int)
    n_w = 44 * N + 6
    x0, lbx, ubx, p = np.zeros(n_w), np.zeros(n_w), np.zeros(n_w), np.zeros(875)
    lib().emu_loop_prepare_obs(N, P(S), P(prev), P(x0), P(lbx), P(ubx), P(p), len(obs_sets), P(A), P(b), nrows.ctypes.data_as(ip),
                               P(V), nv.ctypes.data_as(ip))
    return x0, lbx, ubx, p


def finish(N, dt, S, x, prev, status, viol, iters=0, par=None):
    """-> log row; with `par` (the step's parameter vector) -> (log row, MPCData record of the step)."""
    log = np.zeros(lib().emu_loop_logw())
    rec = np.zeros(lib().emu_loop_record_doubles(N)) if par is not None else None
    lib().emu_loop_finish(N, ctypes.c_double(dt), P(S), P(np.ascontiguousarray(x, float)), P(prev), int(status),
                          ctypes.c_double(viol), int(iters), P(log), P(rec) if rec is not None else None,
                          P(np.ascontiguousarray(par, float)) if par is not None else None)
    return log if par is None else (log, rec)


def so3(v, M):
    R, v2, e = np.zeros((3, 3)), np.zeros(3), np.zeros(3)
    lib().emu_so3(P(np.ascontiguousarray(v, float)), P(np.ascontiguousarray(M, float)), P(R), P(v2), P(e))
    return R, v2, e
