"""Evaluate the REFERENCE's own NLP formulation numerically (this container only).

Imports /root/reference/bound_planner/BoundMPC/casadi_ocp_formulation.py unmodified, with the
numeric `casadi` shim and stub `pinocchio`/`cdd`/`cvxpy` modules first on sys.path, and
evaluates f(w,p), g(w,p) at concrete points; complex-step gives exact derivatives.

Never shipped / never imported by the product or by GPU tests: it only produces the
fixtures under tests/golden/*.npz (see gen_golden.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(0, HERE)
sys.path.insert(1, REF)
os.chdir(REF)  # RobotModel.CA_SAVE_PATH is relative (RobotModel.py:9)

import casadi as ca  # noqa: E402  (the shim)

from bound_planner.BoundMPC import casadi_ocp_formulation as ocp  # noqa: E402

NR_SEGS = 4
NJ = 7


def n_w(N):
    return 44 * N + 6


def n_g(N):
    return 147 * (N - 1) + 21


N_P = 875


def _provider(N, w, p):
    """Map SX.sym creations (name, shape, occurrence) -> values sliced from (w, p)."""
    w = np.asarray(w)
    p = np.asarray(p)
    S = NR_SEGS
    o = {}
    off = 0

    def take(n):
        nonlocal off
        v = p[off : off + n]
        off += n
        return v

    # parameter layout, casadi_ocp_formulation.py:383-415
    o["split_idx"] = take(S + 1).reshape(S + 1, 1)
    o["slacks0"] = take(6).reshape(6, 1)
    o["i_omega_ref_0"] = take(3).reshape(3, 1)
    o["initial lie space error"] = take(3 * S).reshape((3, S), order="F")
    o["initial lie space error par"] = take(3 * S).reshape((3, S), order="F")
    o["initial lie space error orth1"] = take(3 * S).reshape((3, S), order="F")
    o["initial lie space error orth2"] = take(3 * S).reshape((3, S), order="F")
    o["x phi_desired"] = take(3).reshape(3, 1)
    o["path parameter switch"] = take(S + 1).reshape(S + 1, 1)
    o["right jacobian at initial error"] = take(9).reshape((3, 3), order="F")
    o["left jacobian at initial error"] = take(9).reshape((3, 3), order="F")
    o["linear ref position"] = take(6 * S).reshape((S, 6), order="F")
    o["linear ref velocity"] = take(6 * S).reshape((S, 6), order="F")
    o["norm of orientation reference"] = take(3 * S).reshape((S, 3), order="F")
    o["orthogonal error basis 1"] = take(3 * S).reshape((S, 3), order="F")
    o["orthogonal error basis 2"] = take(3 * S).reshape((S, 3), order="F")
    o["orthogonal error basis 1r"] = take(3 * S).reshape((S, 3), order="F")
    o["orthogonal error basis 2r"] = take(3 * S).reshape((S, 3), order="F")
    o["error bounds orientation"] = take(6 * S).reshape((S, 6), order="F")
    o["cost weights"] = take(11).reshape(11, 1)
    o["max path parameter"] = take(1).reshape(1, 1)
    o["v1"] = take(3 * S).reshape((S, 3), order="F")
    o["v2"] = take(3 * S).reshape((S, 3), order="F")
    o["v3"] = take(3 * S).reshape((S, 3), order="F")
    o["q desired"] = take(NJ).reshape(NJ, 1)
    a_set = [take(45).reshape((15, 3), order="F") for _ in range(S)]
    o["b_set"] = take(15 * S).reshape((S, 15), order="F")
    a_set_j = [take(45).reshape((15, 3), order="F") for _ in range(6)]
    o["b_set_joints"] = take(90).reshape((6, 15), order="F")
    assert off == N_P

    # decision vector layout, casadi_ocp_formulation.py:89-101
    wo = 0

    def takew(n):
        nonlocal wo
        v = w[wo : wo + n]
        wo += n
        return v

    dec = {}
    dec["q"] = takew(NJ * N).reshape((N, NJ), order="F")
    dec["dq"] = takew(NJ * N).reshape((N, NJ), order="F")
    dec["ddq"] = takew(NJ * N).reshape((N, NJ), order="F")
    dec["u"] = takew(NJ * N).reshape((N, NJ), order="F")
    dec["p"] = takew(6 * N).reshape((N, 6), order="F")
    dec["v"] = takew(6 * N).reshape((N, 6), order="F")
    dslacks = takew(6).reshape(6, 1)
    srot = [takew(N).reshape(N, 1) for _ in range(4)]  # rslacks, drslacks, pslacks, dpslacks
    assert wo == n_w(N)

    counters = {"s sets": 0, "s rot": 0}

    def prov(idx, name, shape, kk):
        if name == "a_set":
            return a_set[kk]
        if name == "a_set_joints":
            return a_set_j[kk]
        if name == "s sets":  # creation order: dslacks, slacks0 (casadi_ocp_formulation.py:82-83)
            c = counters["s sets"]
            counters["s sets"] += 1
            return dslacks if c == 0 else o["slacks0"]
        if name == "s rot":
            c = counters["s rot"]
            counters["s rot"] += 1
            return srot[c]
        if name == "split_idx":
            return o["split_idx"]
        if name in dec and shape == dec[name].shape:
            return dec[name]
        if name in o and shape == o[name].shape:
            return o[name]
        # RobotModel.setup_ik_problem symbols ("q" 7x1, "p desired", "r desired"): irrelevant
        return None

    return prov


def eval_fg(N, w, p, dt=0.1):
    """f (scalar) and g (n_g,) of the reference NLP at (w, p); complex inputs allowed."""
    cplx = np.iscomplexobj(w) or np.iscomplexobj(p)
    ca.DTYPE[0] = complex if cplx else float
    ca.SYM_LOG.clear()
    ca.PROVIDER[0] = _provider(N, np.asarray(w, dtype=ca.DTYPE[0]), np.asarray(p, dtype=ca.DTYPE[0]))
    solver, lbg, ubg = ocp.setup_optimization_problem(N, NJ, NR_SEGS, dt, {})
    prob = ca.CAPTURED["nlpsol"]
    f = prob["f"].a.reshape(-1)[0]
    g = prob["g"].a.reshape(-1, order="F")
    x = prob["x"].a.reshape(-1, order="F")
    pp = prob["p"].a.reshape(-1, order="F")
    assert np.allclose(x, w) and np.allclose(pp, p), "symbol feed mismatch"
    assert g.size == n_g(N)
    return f, g, np.array(lbg, dtype=float), np.array(ubg, dtype=float)


def eval_derivs(N, w, p, cols=None, h=1e-30, dt=0.1):
    """Complex-step grad f and columns of J_g (all columns when cols is None)."""
    w = np.asarray(w, dtype=float)
    nw = w.size
    cols = range(nw) if cols is None else cols
    grad = np.zeros(nw)
    jac = np.zeros((n_g(N), nw))
    for j in cols:
        wc = w.astype(complex)
        wc[j] += 1j * h
        f, g, _, _ = eval_fg(N, wc, p, dt)
        grad[j] = f.imag / h
        jac[:, j] = g.imag / h
    return grad, jac


def eval_dir(N, w, p, r, h=1e-30, dt=0.1):
    """Directional derivatives (grad f . r, J_g r) by one complex step."""
    wc = np.asarray(w, dtype=complex) + 1j * h * np.asarray(r)
    f, g, _, _ = eval_fg(N, wc, p, dt)
    return f.imag / h, g.imag / h


if __name__ == "__main__":
    N = 6
    rng = np.random.default_rng(0)
    w = rng.normal(size=n_w(N)) * 0.1
    p = rng.normal(size=N_P) * 0.1
    p[0:5] = [0, N, N, N, N]
    f, g, lbg, ubg = eval_fg(N, w, p)
    print("f", f, "g", g.shape, g[:5])
