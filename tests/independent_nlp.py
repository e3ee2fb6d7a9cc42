"""Independent NLP solve of one BoundMPC instance with scipy's SLSQP (a sequential-QP method that
shares no code and no algorithmic family with the interior-point oracle / HIP path) on the PINNED
full-space NLP: objective, constraints and their derivatives are the golden-validated
`bmpc_oracle_eval` (tests/test_oracle_nlp.py pins it to the reference's own formulation,
/root/reference/bound_planner/BoundMPC/casadi_ocp_formulation.py:13-421).  SURVEY.md 8(c) bridge (ii).
TEST INFRASTRUCTURE ONLY."""
import numpy as np
from scipy.optimize import minimize

import oracle_lib as O


def slsqp_solve(N, x0, lbx, ubx, p, maxiter=400):
    lbg, ubg = O.gbounds(N)
    eq = lbg == ubg
    up = (~eq) & (ubg < 1e19)
    dn = (~eq) & (lbg > -1e19)
    cache = {}

    def ev(x):
        k = x.tobytes()
        if k not in cache:
            cache.clear()
            cache[k] = O.nlp_eval(N, x, p)
        return cache[k]

    cons = [{"type": "eq", "fun": lambda x: ev(x)[1][eq] - lbg[eq], "jac": lambda x: ev(x)[3][eq]},
            {"type": "ineq", "fun": lambda x: ubg[up] - ev(x)[1][up], "jac": lambda x: -ev(x)[3][up]}]
    if dn.any():
        cons.append({"type": "ineq", "fun": lambda x: ev(x)[1][dn] - lbg[dn], "jac": lambda x: ev(x)[3][dn]})
    lo = np.where(lbx < -1e19, -np.inf, lbx)
    hi = np.where(ubx > 1e19, np.inf, ubx)
    s = minimize(lambda x: ev(x)[0], np.clip(x0, lo, hi), jac=lambda x: ev(x)[2], bounds=list(zip(lo, hi)),
                 constraints=cons, method="SLSQP", options=dict(maxiter=maxiter, ftol=1e-12))
    return s


def _structure(N):
    """Colour groups for finite-difference Hessians: in the reference's variable-major layout
    (casadi_ocp_formulation.py:89-101) variable (field f, stage k) sits at f*N + k for the 40 joint / task fields, the
    six global dslacks at 40N..40N+5, and the four per-stage slack fields behind them.  Every nonlinear term of the NLP
    couples variables of ONE stage only (dynamics, trapezoids and slack integrators are linear), so perturbing one field
    at all stages at once recovers its Hessian columns exactly."""
    groups = [np.arange(f * N, (f + 1) * N) for f in range(40)]
    groups += [np.array([40 * N + i]) for i in range(6)]
    groups += [np.arange(40 * N + 6 + f * N, 40 * N + 6 + (f + 1) * N) for f in range(4)]
    stage = np.full(44 * N + 6, -1)
    for g in groups:
        if len(g) == N:
            stage[g] = np.arange(N)
    return groups, stage


def trust_constr_solve(N, x0, lbx, ubx, p, maxiter=3000, verbose=0):
    """scipy `trust-constr` (Byrd-Hribar-Nocedal trust-region interior point; scipy's own implementation) on the same
    pinned full-space NLP: sparse analytic Jacobian, Lagrangian Hessian by coloured forward differences of the pinned
    analytic gradients."""
    import scipy.sparse as sp
    from scipy.optimize import Bounds, NonlinearConstraint
    lbg, ubg = O.gbounds(N)
    lbg = np.where(lbg < -1e19, -np.inf, lbg); ubg = np.where(ubg > 1e19, np.inf, ubg)
    lo = np.where(lbx < -1e19, -np.inf, lbx); hi = np.where(ubx > 1e19, np.inf, ubx)
    n = x0.size
    groups, stage = _structure(N)
    cache = {}

    def ev(x):
        k = x.tobytes()
        if k not in cache:
            if len(cache) > 4:
                cache.clear()
            cache[k] = O.nlp_eval(N, x, p)
        return cache[k]

    def fd_hess(gradfun, x):
        g0 = gradfun(x)
        rows, cols, vals = [], [], []
        eps = 1e-7
        for grp in groups:
            h = eps * np.maximum(1.0, np.abs(x[grp]))
            xp = x.copy(); xp[grp] += h
            dg = gradfun(xp) - g0
            nz = np.nonzero(dg)[0]
            if len(grp) == 1:
                rows += list(nz); cols += [grp[0]] * len(nz); vals += list(dg[nz] / h[0])
            else:
                # row r belongs to stage stage[r]: its entry is the derivative w.r.t. this field at the same stage
                for r in nz:
                    k = stage[r]
                    if k >= 0:
                        rows.append(r); cols.append(grp[k]); vals.append(dg[r] / h[k])
                    else:           # a global variable's row: sum over stages cannot be separated -- must vanish
                        assert abs(dg[r]) < 1e-9 * (1 + np.abs(g0[r])), "global slack couples nonlinearly"
        H = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
        return 0.5 * (H + H.T)

    con = NonlinearConstraint(lambda x: ev(x)[1], lbg, ubg, jac=lambda x: sp.csr_matrix(ev(x)[3]),
                              hess=lambda x, v: fd_hess(lambda y: ev(y)[3].T @ v, x))
    s = minimize(lambda x: ev(x)[0], np.clip(x0, lo, hi), jac=lambda x: ev(x)[2], hess=lambda x: fd_hess(lambda y: ev(y)[2], x),
                 bounds=Bounds(lo, hi), constraints=[con], method="trust-constr",
                 options=dict(maxiter=maxiter, xtol=1e-12, gtol=1e-9, barrier_tol=1e-10, initial_barrier_parameter=0.1,
                              sparse_jacobian=True, verbose=verbose))
    return s
