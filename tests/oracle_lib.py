"""ctypes access to the CPU oracle (oracle/libbmpc_oracle.so) -- TEST INFRASTRUCTURE ONLY.
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libbmpc_oracle.so")
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


class Opts(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int), ("dt", ctypes.c_double), ("tol", ctypes.c_double),
                ("max_iter", ctypes.c_int), ("verbose", ctypes.c_int), ("hess", ctypes.c_int), ("mu_strategy", ctypes.c_int), ("hess_switch", ctypes.c_double), ("mu_init", ctypes.c_double), ("kappa_mu", ctypes.c_double), ("theta_mu", ctypes.c_double), ("kappa_eps", ctypes.c_double),
                ("inertia", ctypes.c_int), ("dw0", ctypes.c_double), ("inertia_err", ctypes.c_double), ("stall_n", ctypes.c_int),
                ("gn_backoff", ctypes.c_int), ("slack_reset", ctypes.c_int), ("ls_alpha_mem", ctypes.c_double), ("mu_floor_k", ctypes.c_double),
                ("soc", ctypes.c_int), ("soc_after", ctypes.c_int), ("pi_shoot", ctypes.c_int)]


def _P(a):
    return a.ctypes.data_as(_dp)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib = ctypes.CDLL(LIB)
        _lib.bmpc_oracle_eval.restype = ctypes.c_int
        _lib.bmpc_oracle_solve.restype = ctypes.c_int
    return _lib


def set_robot(table=None):
    """table: boundplanner_amd.robots table (dict) or None for the iiwa14 -- process-wide."""
    if table is None:
        lib().bmpc_oracle_set_robot(None, None, None, None, None)
        return
    a = lambda k: np.ascontiguousarray(table[k], float)
    xyz, rpy, ee, eer, l4 = a("joint_xyz"), a("joint_rpy"), a("ee_xyz"), a("ee_rpy"), a("link4_col_xyz")
    lib().bmpc_oracle_set_robot(_P(xyz), _P(rpy), _P(ee), _P(eer), _P(l4))


def fk_batch(q, dq=None):
    q = np.ascontiguousarray(q, float).reshape(-1, 7)
    dq = np.zeros_like(q) if dq is None else np.ascontiguousarray(dq, float).reshape(-1, 7)
    B = q.shape[0]
    out = dict(ee_pos=np.zeros((B, 3)), ee_rot=np.zeros((B, 3, 3)), col_pts=np.zeros((B, 6, 3)),
               jac=np.zeros((B, 6, 7)), dvdq=np.zeros((B, 6, 7)))
    L = lib()
    for b in range(B):
        L.bmpc_oracle_fk(_P(q[b]), _P(dq[b]), _P(out["ee_pos"][b]), _P(out["ee_rot"][b]),
                         _P(out["col_pts"][b]), _P(out["jac"][b]), _P(out["dvdq"][b]))
    return out


def nlp_eval(N, w, p, dt=0.1, jac=True):
    n_w, n_g = 44 * N + 6, 147 * (N - 1) + 21
    w = np.ascontiguousarray(w, float); p = np.ascontiguousarray(p, float)
    f = ctypes.c_double()
    g = np.zeros(n_g); gr = np.zeros(n_w)
    J = np.zeros((n_g, n_w)) if jac else None
    rc = lib().bmpc_oracle_eval(N, ctypes.c_double(dt), _P(w), _P(p), ctypes.byref(f), _P(g), _P(gr),
                                _P(J) if jac else None)
    assert rc == 0
    return f.value, g, gr, J


def gbounds(N):
    n_g = 147 * (N - 1) + 21
    lb = np.zeros(n_g); ub = np.zeros(n_g)
    lib().bmpc_oracle_gbounds(N, _P(lb), _P(ub))
    return lb, ub


def solve(N, x0, lbx, ubx, p, dt=0.1, tol=1e-5, max_iter=100, verbose=0, hess=2, mu_strategy=1, hess_switch=1.0, mu_init=0.1, kappa_mu=0.1, theta_mu=2.0, kappa_eps=1000.0, inertia=2, dw0=1e-4, inertia_err=1e-2, stall_n=8, mu_floor_k=1e4, gn_backoff=2, slack_reset=1, ls_alpha_mem=0.0, soc=0, soc_after=0, pi_shoot=0):
    n_w, n_g = 44 * N + 6, 147 * (N - 1) + 21
    o = Opts(N, dt, tol, max_iter, verbose, hess, mu_strategy, hess_switch, mu_init, kappa_mu, theta_mu, kappa_eps, inertia, dw0, inertia_err, stall_n, gn_backoff, slack_reset, ls_alpha_mem, mu_floor_k, soc, soc_after, pi_shoot)
    lbx = np.where(np.isinf(lbx), -1e20, lbx); ubx = np.where(np.isinf(ubx), 1e20, ubx)
    x0, lbx, ubx, p = (np.ascontiguousarray(a, float) for a in (x0, lbx, ubx, p))
    x = np.zeros(n_w); g = np.zeros(n_g); lg = np.zeros(n_g); lx = np.zeros(n_w)
    f = ctypes.c_double(); it = ctypes.c_int(); st = ctypes.c_int(); viol = ctypes.c_double()
    lib().bmpc_oracle_solve(ctypes.byref(o), _P(x0), _P(lbx), _P(ubx), _P(p), _P(x), _P(g), _P(lg), _P(lx),
                            ctypes.byref(f), ctypes.byref(it), ctypes.byref(st), ctypes.byref(viol))
    return dict(x=x, g=g, lam_g=lg, lam_x=lx, f=f.value, iters=it.value, status=st.value, viol=viol.value)


def solve_batch(N, x0, lbx, ubx, p, dt=0.1, tol=1e-5, max_iter=100, nthreads=0, hess=2, mu_strategy=1, hess_switch=1.0, mu_init=0.1, kappa_mu=0.1, theta_mu=2.0, kappa_eps=1000.0, inertia=2, dw0=1e-4, inertia_err=1e-2, stall_n=8, mu_floor_k=1e4, gn_backoff=2, slack_reset=1, ls_alpha_mem=0.0, soc=0, soc_after=0, pi_shoot=0):
    B = x0.shape[0]
    o = Opts(N, dt, tol, max_iter, 0, hess, mu_strategy, hess_switch, mu_init, kappa_mu, theta_mu, kappa_eps, inertia, dw0, inertia_err, stall_n, gn_backoff, slack_reset, ls_alpha_mem, mu_floor_k, soc, soc_after, pi_shoot)
    lbx = np.where(np.isinf(lbx), -1e20, lbx); ubx = np.where(np.isinf(ubx), 1e20, ubx)
    x0, lbx, ubx, p = (np.ascontiguousarray(a, float) for a in (x0, lbx, ubx, p))
    x = np.zeros_like(x0); f = np.zeros(B); viol = np.zeros(B)
    it = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
    lib().bmpc_oracle_solve_batch(ctypes.byref(o), B, _P(x0), _P(lbx), _P(ubx), _P(p), _P(x), _P(f),
                                  it.ctypes.data_as(_ip), st.ctypes.data_as(_ip), _P(viol), nthreads)
    return dict(x=x, f=f, iters=it, status=st, viol=viol)


INFO_FIELDS = ("iters", "status", "mu", "alpha", "alpha_dual", "alpha_ftb", "delta_w", "hess_next", "retries", "backtracks", "err_prev", "stall")


def solve_batch_info(N, x0, lbx, ubx, p, nthreads=0, **kw):
    """solve_batch + info [B][12]: the decisions of the last iteration of every solve (bmpc_oracle_solve_batch_info)."""
    B = x0.shape[0]
    d = dict(dt=0.1, tol=1e-5, max_iter=100, hess=2, mu_strategy=1, hess_switch=1.0, mu_init=0.1, kappa_mu=0.1, theta_mu=2.0, kappa_eps=1000.0,
             inertia=2, dw0=1e-4, inertia_err=1e-2, stall_n=8, mu_floor_k=1e4, gn_backoff=2, slack_reset=1, ls_alpha_mem=0.0, soc=0, soc_after=0, pi_shoot=0)
    d.update(kw)
    o = Opts(N, d["dt"], d["tol"], d["max_iter"], 0, d["hess"], d["mu_strategy"], d["hess_switch"], d["mu_init"], d["kappa_mu"], d["theta_mu"],
             d["kappa_eps"], d["inertia"], d["dw0"], d["inertia_err"], d["stall_n"], d["gn_backoff"], d["slack_reset"], d["ls_alpha_mem"],
             d["mu_floor_k"], d["soc"], d["soc_after"], d["pi_shoot"])
    lbx = np.where(np.isinf(lbx), -1e20, lbx); ubx = np.where(np.isinf(ubx), 1e20, ubx)
    x0, lbx, ubx, p = (np.ascontiguousarray(a, float) for a in (x0, lbx, ubx, p))
    x = np.zeros_like(x0); f = np.zeros(B); viol = np.zeros(B); info = np.zeros((B, 12))
    it = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
    lib().bmpc_oracle_solve_batch_info(ctypes.byref(o), B, _P(x0), _P(lbx), _P(ubx), _P(p), _P(x), _P(f),
                                       it.ctypes.data_as(_ip), st.ctypes.data_as(_ip), _P(viol), _P(info), nthreads)
    return dict(x=x, f=f, iters=it, status=st, viol=viol, info=info)
