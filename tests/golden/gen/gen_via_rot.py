"""Generate tests/golden/via_rot.npz: objective and constraint values of the reference's via-point / rotation NLP
(/root/reference/bound_planner/utils/optimization_functions.py:227-387, via_point_rot_optimization_problem) at sample points.

Runs, in the build container only, the reference's UNMODIFIED problem construction under the numeric `casadi` stand-in of
tests/golden/gen/stubs: every SX.sym is a concrete array supplied below, so building the problem evaluates f and g at that point; the
two symbolic calls on the way -- ca.jacobian(p_max_ee, phi_max) and the ca.Function f_max called at phi_max, 0 and 1 -- are served by
re-evaluating the recorded expression (complex step for the derivative).  The fixture is data only; no reference source is copied.

    python tests/golden/gen/gen_via_rot.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.abspath(os.path.join(HERE, ".."))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(1, REF)
os.chdir(REF)

import casadi as ca  # noqa: E402  (the stand-in)
from bound_planner.utils import optimization_functions as OF  # noqa: E402


def sample(rng, nr_via, S):
    """values for every symbol of the problem, by name"""
    def sets(n):
        A = np.zeros((n, S, 3)); b = np.zeros((S, n))
        for i in range(n):
            m = rng.integers(3, S + 1)                   # m real rows, the rest padding (a = 0, b = 1) as normalize_set_size leaves it
            a = rng.normal(size=(m, 3)); a /= np.linalg.norm(a, axis=1)[:, None]
            A[i, :m] = a; b[:m, i] = rng.uniform(0.2, 0.8, m); b[m:, i] = 1.0
        return A, b
    Ai, bi = sets(nr_via); Av, bv = sets(nr_via + 1)
    om = rng.normal(size=3); om /= np.linalg.norm(om)
    vals = {"a set inter": Ai, "b set inter": bi, "a set via": Av, "b set via": bv, "w size via": rng.uniform(0.5, 2.0, nr_via + 1),
            "p start": rng.uniform(-0.3, 0.3, 3), "p end": rng.uniform(-0.3, 0.3, 3), "l ee": rng.uniform(-0.2, 0.2, 3), "omega": om,
            "omega norm": rng.uniform(0.3, 2.5), "omega_prev": 0.0}
    for i in range(nr_via):
        vals[f"p_via {i}"] = rng.uniform(-0.3, 0.3, 3)
        vals[f"omega_via {i}"] = rng.uniform(0.0, 1.0)
        for j in range(S):
            vals[f"phi_max {i} {j}"] = rng.uniform(0.0, 1.0)
    return vals


def evaluate(nr_via, S, vals):
    def provider(idx, name, shape, kk):
        v = vals[name]
        if kk is not None:
            v = v[kk]
        return np.asarray(v, float).reshape(shape)
    ca.SYM_LOG.clear(); ca.CAPTURED.clear()
    ca.PROVIDER[0] = provider
    _, lbu, ubu, lbg, ubg = OF.via_point_rot_optimization_problem(nr_via, S)
    ca.PROVIDER[0] = None
    prob = ca.CAPTURED["nlpsol"]
    full = lambda m: np.real(np.asarray(m.a)).reshape(-1, order="F")
    return full(prob["x"]), full(prob["p"]), float(np.real(prob["f"].a).ravel()[0]), full(prob["g"]), np.array(lbg, float), np.array(ubg, float)


out = {}
rng = np.random.default_rng(227)
for tag, (nr_via, S, n) in {"a": (2, 6, 6), "b": (3, 5, 4), "c": (1, 8, 4)}.items():
    X, P, F, G = [], [], [], []
    for _ in range(n):
        x, p, f, g, lbg, ubg = evaluate(nr_via, S, sample(rng, nr_via, S))
        X.append(x); P.append(p); F.append(f); G.append(g)
    out.update({f"{tag}_nr_via": nr_via, f"{tag}_S": S, f"{tag}_x": np.array(X), f"{tag}_p": np.array(P), f"{tag}_f": np.array(F),
                f"{tag}_g": np.array(G), f"{tag}_lbg": lbg, f"{tag}_ubg": ubg})
    print(tag, nr_via, S, "x", X[0].shape, "p", P[0].shape, "g", G[0].shape, "stationarity rows active:",
          int(sum((g_.reshape(-1)[:] != 0).sum() for g_ in G)), "f", F[:2])
np.savez(os.path.join(OUT, "via_rot.npz"), **out)
print("via_rot.npz written")
