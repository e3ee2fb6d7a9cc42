"""Small exact solvers behind the plan phase.

The reference's planner delegates these sub-problems to third-party packages that are not in this image:
  * qpOASES / OSQP through CasADi's qpsol -- point projections onto polytopes (utils/optimization_functions.py:107-137,
    BoundPlanner/ConvexSetFinder.py:10-49), the "end effector fits" feasibility problem (:140-185);
  * cvxpy + Clarabel -- the maximum-volume inscribed ellipsoid SOCPs (ConvexSetFinder.py:512-640);
  * pycddlib -- vertex enumeration and redundancy removal (utils/util_functions.py:68-90);
  * IPOPT -- the via-point / rotation NLP (utils/optimization_functions.py:227-387).
Here every one of them is a few dozen lines of numpy (scipy.optimize.linprog, which the reference itself uses for its
intersection test, serves the LPs; scipy's SLSQP the one NLP): the problems have three to sixteen unknowns.  Host code:
planning runs once per (re)plan, not in the per-step path.
"""
import itertools

import numpy as np
from scipy.optimize import linprog, minimize


# ------------------------------------------------------------------------------------------------------------
# Euclidean projection onto a polytope: min |x - y|^2  s.t.  A x <= b   (3 unknowns)
# ------------------------------------------------------------------------------------------------------------
def project_polytope(A, b, y):
    """Exact: the optimum has at most three active rows; all active sets of size 0..3 are tried (vectorised) and the KKT
    point (primal feasible, multipliers >= 0) of smallest distance is returned."""
    A = np.asarray(A, float); b = np.asarray(b, float); y = np.asarray(y, float)
    nrm = np.linalg.norm(A, axis=1)
    keep = nrm > 0
    A, b = A[keep] / nrm[keep, None], b[keep] / nrm[keep]      # unit rows: the singularity / feasibility tolerances are scale-free
    if A.shape[0] == 0 or np.all(A @ y - b <= 0):
        return y.copy()
    m = A.shape[0]
    best, best_d = None, np.inf
    tol = 1e-10
    for k in (1, 2, 3):
        if m < k:
            break
        idx = np.array(list(itertools.combinations(range(m), k)))          # [n][k]
        Ak, bk = A[idx], b[idx]                                             # [n][k][3], [n][k]
        G = Ak @ Ak.transpose(0, 2, 1)                                      # Gram matrices [n][k][k]
        r = Ak @ y - bk                                                     # [n][k]
        det = np.linalg.det(G)
        ok = np.abs(det) > 1e-14
        if not ok.any():
            continue
        lam = np.zeros_like(r)
        lam[ok] = np.linalg.solve(G[ok], r[ok][..., None])[..., 0]          # multipliers of x = y - A_k^T lam
        x = y[None, :] - np.einsum("nk,nkc->nc", lam, Ak)
        feas = ((x @ A.T - b[None, :]).max(axis=1) <= tol) & (lam >= -tol).all(axis=1) & ok
        if feas.any():
            d = np.linalg.norm(x - y[None, :], axis=1)
            d[~feas] = np.inf
            j = int(np.argmin(d))
            if d[j] < best_d - 1e-14:
                best, best_d = x[j], d[j]
        if best is not None:            # a KKT point of a strictly convex QP is THE solution
            break
    if best is None:
        raise RuntimeError("project_polytope: no KKT point found (empty polytope?)")
    return best


# ------------------------------------------------------------------------------------------------------------
# polytope bookkeeping (cdd in the reference)
# ------------------------------------------------------------------------------------------------------------
def polytope_vertices(A, b, tol=1e-9):
    """Vertices of the bounded polytope {x: A x <= b} in R^3: intersections of three faces that satisfy the rest."""
    A = np.asarray(A, float); b = np.asarray(b, float)
    keep = np.abs(A).sum(axis=1) > 0
    A, b = A[keep], b[keep]
    idx = np.array(list(itertools.combinations(range(A.shape[0]), 3)))
    M = A[idx]
    ok = np.abs(np.linalg.det(M)) > 1e-12
    x = np.linalg.solve(M[ok], b[idx][ok][..., None])[..., 0]
    x = x[(x @ A.T - b[None, :]).max(axis=1) <= tol]
    out = []
    for v in x:
        if not any(np.linalg.norm(v - w) < 1e-9 for w in out):
            out.append(v)
    return np.array(out)


def reduce_ineqs(A, b):
    """Redundancy removal: a row is dropped when it cannot be active given the others (an LP per row, rows visited in
    order, the set shrinking as it goes so that of two identical rows one survives).  All-zero rows go."""
    A = np.asarray(A, float); b = np.asarray(b, float).ravel()
    rows = [i for i in range(A.shape[0]) if np.abs(A[i]).sum() > 0]
    i = 0
    while i < len(rows):
        r = rows[i]
        others = [j for j in rows if j != r]
        if others:
            res = linprog(-A[r], A_ub=np.vstack((A[others], A[r][None, :])), b_ub=np.concatenate((b[others], [b[r] + 1.0])),
                          bounds=(None, None), method="highs")
            if res.status == 0 and -res.fun <= b[r] + 1e-9:
                rows.pop(i)
                continue
        i += 1
    return [A[rows].copy(), b[rows].copy()]


def feasible_point(A, b):
    """Any point of {x: A x <= b} (linprog with a zero objective, as BoundPlanner.py:774-787 does); (x, success)."""
    res = linprog(np.zeros(3), A_ub=A, b_ub=b, bounds=(None, None))
    return res.x, bool(res.success)


def chebyshev_center(A, b):
    A = np.asarray(A, float); b = np.asarray(b, float)
    n = np.linalg.norm(A, axis=1)
    keep = n > 0
    res = linprog([0, 0, 0, -1.0], A_ub=np.hstack((A[keep], n[keep, None])), b_ub=b[keep], bounds=[(None, None)] * 3 + [(0, None)],
                  method="highs")
    if res.status != 0:
        raise RuntimeError("chebyshev_center: empty polytope")
    return res.x[:3], res.x[3]


# ------------------------------------------------------------------------------------------------------------
# maximum-volume inscribed ellipsoid, in the reference's parameterisation (ConvexSetFinder.py:512-560, 790-836):
#   ellipsoid {c + L u: |u| <= 1}, L lower triangular;   |L^T a_i| <= b_i - a_i . c   for every row;
#   maximise kappa with kappa^2 <= x9 x10, x9^2 <= L00 L11, x10^2 <= L11 L22, i.e. (L00 L11^2 L22)^(1/4)
# ------------------------------------------------------------------------------------------------------------
_TRIL = np.tril_indices(3)


def mvie(A, b, fixed_mid=None, tol=1e-10):
    """Returns (q = L L^T, centre).  Log-barrier Newton method on the 9 (6 with a fixed centre) unknowns: minimise
    -(log L00 / 4 + log L11 / 2 + log L22 / 4) - (1/t) sum_i log((b_i - a_i.c)^2 - |L^T a_i|^2) for t -> infinity."""
    A = np.asarray(A, float); b = np.asarray(b, float)
    keep = np.abs(A).sum(axis=1) > 0
    A, b = A[keep], b[keep]
    m = A.shape[0]
    free_mid = fixed_mid is None
    if free_mid:
        c0, r0 = chebyshev_center(A, b)
    else:
        c0 = np.asarray(fixed_mid, float)
        r0 = np.min((b - A @ c0) / np.linalg.norm(A, axis=1))
        if r0 <= 0:
            raise RuntimeError("mvie: the fixed centre is not inside the polytope")
    nx = 9 if free_mid else 6
    x = np.zeros(nx)
    x[[0, 2, 5]] = 0.5 * r0
    if free_mid:
        x[6:9] = c0
    # row i: s = d_i + cvec_i . x,  v = M_i x  (3 x nx)
    Mi = np.zeros((m, 3, nx))
    Mi[:, 0, 0] = A[:, 0]; Mi[:, 0, 1] = A[:, 1]; Mi[:, 0, 3] = A[:, 2]
    Mi[:, 1, 2] = A[:, 1]; Mi[:, 1, 4] = A[:, 2]
    Mi[:, 2, 5] = A[:, 2]
    cv = np.zeros((m, nx))
    if free_mid:
        cv[:, 6:9] = -A
        d = b.copy()
    else:
        d = b - A @ c0
    MtM = np.einsum("mki,mkj->mij", Mi, Mi)
    w = np.array([0.25, 0.5, 0.25])
    di = np.array([0, 2, 5])

    def parts(x):
        s = d + cv @ x
        v = np.einsum("mki,i->mk", Mi, x)
        return s, s * s - (v * v).sum(axis=1)

    def value(x, t):
        s, psi = parts(x)
        if (s <= 0).any() or (psi <= 0).any() or (x[di] <= 0).any():
            return np.inf
        return -t * (w * np.log(x[di])).sum() - np.log(psi).sum()

    t = 1.0
    for _outer in range(60):
        for _newton in range(60):
            s, psi = parts(x)
            gpsi = 2 * s[:, None] * cv - 2 * np.einsum("mij,j->mi", MtM, x)            # [m][nx]
            g = -(gpsi / psi[:, None]).sum(axis=0)
            H = np.einsum("mi,mj->ij", gpsi / psi[:, None], gpsi / psi[:, None]) \
                - ((2 * np.einsum("mi,mj->mij", cv, cv) - 2 * MtM) / psi[:, None, None]).sum(axis=0)
            g[di] -= t * w / x[di]
            H[di, di] += t * w / x[di] ** 2
            dx = -np.linalg.solve(H, g)
            dec = -g @ dx
            if dec < 1e-22 * max(1.0, t):
                break
            a, f0 = 1.0, value(x, t)
            while value(x + a * dx, t) > f0 - 1e-4 * a * dec and a > 1e-14:
                a *= 0.5
            x = x + a * dx
            if dec < 1e-18 * max(1.0, t):
                break
        if 2.0 * m / t < tol:
            break
        t *= 8.0
    L = np.zeros((3, 3))
    L[_TRIL] = x[:6]
    return L @ L.T, (x[6:9].copy() if free_mid else c0.copy())


# ------------------------------------------------------------------------------------------------------------
# "does the end effector fit": a point p with A p <= b and A (p + l) <= b   (optimization_functions.py:140-185)
# ------------------------------------------------------------------------------------------------------------
def fits(A, b, l_ee):
    res = linprog(np.zeros(3), A_ub=np.vstack((A, A)), b_ub=np.concatenate((b, b - A @ l_ee)), bounds=(None, None))
    return bool(res.success)


# ------------------------------------------------------------------------------------------------------------
# via points with rotation (optimization_functions.py:227-387)
# ------------------------------------------------------------------------------------------------------------
def rodrigues(axis, angle):
    k = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * k + (1 - np.cos(angle)) * (k @ k)


def _sweep_row(a, p_prev, p, om_prev, om, l_ee, omega, omega_norm):
    """h(phi) = a . (p_prev + phi (p - p_prev) + Rod(omega, omega_norm (om_prev + phi (om - om_prev))) l_ee) on [0, 1]:
    its interior stationary point when h' changes sign (else None) -- the point the reference's phi_max variables sit at."""
    v = p - p_prev
    dom = omega_norm * (om - om_prev)
    k = np.cross(omega, l_ee); kk = np.cross(omega, k)

    def dh(phi):
        ang = omega_norm * (om_prev + phi * (om - om_prev))
        return a @ v + dom * (np.cos(ang) * (a @ k) + np.sin(ang) * (a @ kk))
    d0, d1 = dh(0.0), dh(1.0)
    if d0 * d1 >= 0:
        return None
    lo, hi = 0.0, 1.0
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if dh(mid) * d0 > 0:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def via_rot_reference_fg(nr_via, max_set_size, x, params):
    """Objective and constraint vector of the reference's via-point / rotation NLP in ITS variables and ordering
    (/root/reference/bound_planner/utils/optimization_functions.py:227-387): x = per via point [p (3), omega (1), phi_max (S)],
    g = per via point [A_inter p - b (S); per row j of its via set: (d/dphi of the swept row at phi_max_j if that derivative changes
    sign on [0, 1], else 0), (swept row at phi_max_j) - b_j; A_inter p_ee - b (S)], then the two mid points of the last segment
    (2 S).  Pinned against the reference's own construction of the problem (tests/golden/via_rot.npz); via_rot_problem below solves
    the same problem with the phi_max variables eliminated."""
    S = max_set_size
    params = np.asarray(params, float); x = np.asarray(x, float)
    p_start, p_end, l_ee, omega = params[0:3], params[3:6], params[6:9], params[9:12]
    omega_norm = params[12]
    w = params[13:13 + nr_via + 1]
    o = 13 + nr_via + 1
    inter, via = [], []
    for _ in range(nr_via):
        inter.append((params[o:o + 3 * S].reshape(3, S).T, params[o + 3 * S:o + 4 * S])); o += 4 * S
    for _ in range(nr_via + 1):
        via.append((params[o:o + 3 * S].reshape(3, S).T, params[o + 3 * S:o + 4 * S])); o += 4 * S
    step = 4 + S
    k = np.cross(omega, l_ee); kk = np.cross(omega, k)
    J, g, pp, op = 0.0, [], p_start, 0.0
    for i in range(nr_via):
        P, O, phis = x[step * i:step * i + 3], x[step * i + 3], x[step * i + 4:step * (i + 1)]
        J += w[i] * ((P - pp) @ (P - pp) + (O - op) ** 2)
        Ai, bi = inter[i]
        g.append(Ai @ P - bi)
        Av, bv = via[i]
        v, dom = P - pp, omega_norm * (O - op)
        rows = np.zeros(2 * S)
        for j in range(S):
            def dh(phi, a=Av[j]):
                ang = omega_norm * (op + phi * (O - op))
                return a @ v + dom * (np.cos(ang) * (a @ k) + np.sin(ang) * (a @ kk))
            rows[2 * j] = dh(phis[j]) if dh(0.0) * dh(1.0) < 0 else 0.0
            pm = pp + phis[j] * v + rodrigues(omega, omega_norm * (op + phis[j] * (O - op))) @ l_ee
            rows[2 * j + 1] = Av[j] @ pm - bv[j]
        g.append(rows)
        g.append(Ai @ (P + rodrigues(omega, omega_norm * O) @ l_ee) - bi)
        pp, op = P, O
    J += w[-1] * ((p_end - pp) @ (p_end - pp) + (1 - op) ** 2)
    Av, bv = via[-1]
    for pos in (0.25, 0.5):
        pm = pp + pos * (p_end - pp) + rodrigues(omega, omega_norm * (op + pos * (1 - op))) @ l_ee
        g.append(Av @ pm - bv)
    return J, np.concatenate(g)


def via_rot_problem(nr_via, max_set_size, x0, params):
    """The reference's NLP with its phi_max variables eliminated (each sits at the stationary point of its row's sweep when
    there is one; otherwise its two constraints are vacuous).  x0, params and the returned x are in the reference's layouts
    (per via point: p (3), omega (1), phi_max (max_set_size)).  Returns (x, success)."""
    S = max_set_size
    params = np.asarray(params, float)
    p_start, p_end, l_ee, omega = params[0:3], params[3:6], params[6:9], params[9:12]
    omega_norm = params[12]
    w = params[13:13 + nr_via + 1]
    o = 13 + nr_via + 1
    inter, via = [], []
    for _ in range(nr_via):
        inter.append((params[o:o + 3 * S].reshape(3, S).T, params[o + 3 * S:o + 4 * S])); o += 4 * S
    for _ in range(nr_via + 1):
        via.append((params[o:o + 3 * S].reshape(3, S).T, params[o + 3 * S:o + 4 * S])); o += 4 * S
    step = 4 + S
    z0 = np.concatenate([np.asarray(x0, float)[step * i:step * i + 4] for i in range(nr_via)])

    def unpack(z):
        return [z[4 * i:4 * i + 3] for i in range(nr_via)], [z[4 * i + 3] for i in range(nr_via)]

    def cost(z):
        P, O = unpack(z)
        J, pp, op = 0.0, p_start, 0.0
        for i in range(nr_via):
            J += w[i] * ((P[i] - pp) @ (P[i] - pp) + (O[i] - op) ** 2)
            pp, op = P[i], O[i]
        return J + w[-1] * ((p_end - pp) @ (p_end - pp) + (1 - op) ** 2)

    def cons(z):            # >= 0
        P, O = unpack(z)
        out, pp, op = [], p_start, 0.0
        for i in range(nr_via):
            Ai, bi = inter[i]
            out.append(bi - Ai @ P[i])
            out.append(bi - Ai @ (P[i] + rodrigues(omega, omega_norm * O[i]) @ l_ee))
            Av, bv = via[i]
            sw = np.full(S, 1.0)
            for j in range(S):
                if np.abs(Av[j]).sum() == 0:
                    continue
                phi = _sweep_row(Av[j], pp, P[i], op, O[i], l_ee, omega, omega_norm)
                if phi is not None:
                    pm = pp + phi * (P[i] - pp) + rodrigues(omega, omega_norm * (op + phi * (O[i] - op))) @ l_ee
                    sw[j] = bv[j] - Av[j] @ pm
            out.append(sw)
            pp, op = P[i], O[i]
        Av, bv = via[-1]
        for pos in (0.25, 0.5):
            pm = pp + pos * (p_end - pp) + rodrigues(omega, omega_norm * (op + pos * (1 - op))) @ l_ee
            out.append(bv - Av @ pm)
        return np.concatenate(out)

    bounds = []
    for _ in range(nr_via):
        bounds += [(None, None)] * 3 + [(0.0, 1.0)]
    res = minimize(cost, z0, method="SLSQP", bounds=bounds, constraints=[{"type": "ineq", "fun": cons}],
                   options={"maxiter": 300, "ftol": 1e-12})
    z = res.x
    ok = bool(res.success) and cons(z).min() > -1e-6
    P, O = unpack(z)
    x = np.zeros(step * nr_via)
    pp, op = p_start, 0.0
    for i in range(nr_via):
        x[step * i:step * i + 3] = P[i]; x[step * i + 3] = O[i]
        for j in range(S):
            phi = None if np.abs(via[i][0][j]).sum() == 0 else _sweep_row(via[i][0][j], pp, P[i], op, O[i], l_ee, omega, omega_norm)
            x[step * i + 4 + j] = 0.5 if phi is None else phi
        pp, op = P[i], O[i]
    return x, ok
