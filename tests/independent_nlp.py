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
