"""Per-step convex collision-avoidance sets for the 6 collision points.

Restates ConvexSetFinder.find_set_collision_avoidance(pl, pf, limit_space=True, e_max)
(/root/reference/bound_planner/BoundPlanner/ConvexSetFinder.py:309-375) together with
init_halfspaces_point (:400-421) and compute_set_projs_line (:491-510).

The reference solves, per obstacle polytope {x: A x <= b - 0.001}, the QP
    min_{x, phi in [0,1]} |p0 + phi (p1 - p0) - x|^2   s.t.  A x <= b - 0.001
with qpOASES through CasADi (third-party, not in this image).  Here the same strictly convex
QP in (x, phi) is solved by a small dual active-set free method: projected alternating
minimisation is NOT used because it is only linearly convergent; instead the closest pair is
found by Dykstra-free exact minimisation over phi (1-D convex, golden section on the distance
from the segment point to the polytope, each distance being an exact polytope projection by
active-set enumeration for boxes / a primal active-set QP for general polytopes).
"""
import numpy as np


def init_halfspaces_point(p, e_max=0.3):
    """Axis-aligned box of half-width e_max around p as 6 halfspaces, ordered +x,-x,+y,-y,+z,-z."""
    a, b = [], []
    for i in range(3):
        e = np.eye(3)[i]
        a.append(e.copy()); b.append(p[i] + e_max)
        a.append(-e); b.append(-p[i] + e_max)
    return a, b


def _project_polytope(A, b, y, iters=60):
    """Euclidean projection of y onto {x: A x <= b} by a primal active-set method (tiny sizes)."""
    x = y.copy()
    viol = A @ x - b
    if np.all(viol <= 1e-12):
        return x
    # Dual coordinate ascent (Hildreth) on lambda >= 0: x = y - A^T lambda; exact for strictly
    # convex projection QPs and adequate for <= 15 rows.
    lam = np.zeros(A.shape[0])
    AAt = A @ A.T
    diag = np.maximum(np.diag(AAt), 1e-16)
    Ay = A @ y
    for _ in range(iters * 20):
        max_change = 0.0
        for i in range(A.shape[0]):
            r = Ay[i] - AAt[i] @ lam - b[i]
            new = max(0.0, lam[i] + r / diag[i])
            max_change = max(max_change, abs(new - lam[i]))
            lam[i] = new
        if max_change < 1e-13:
            break
    return y - A.T @ lam


def closest_pair_segment_polytope(A, b, p0, p1):
    """argmin over x in polytope, phi in [0,1] of |p0 + phi (p1-p0) - x|.  Returns (x, phi)."""
    d = p1 - p0

    def dist(phi):
        y = p0 + phi * d
        x = _project_polytope(A, b, y)
        return np.linalg.norm(y - x), x

    if np.linalg.norm(d) < 1e-12:
        return dist(0.0)[1], 0.0
    lo, hi = 0.0, 1.0
    gr = (np.sqrt(5.0) - 1.0) / 2.0
    c, e = hi - gr * (hi - lo), lo + gr * (hi - lo)
    fc, fe = dist(c)[0], dist(e)[0]
    for _ in range(80):
        if fc < fe:
            hi, e, fe = e, c, fc
            c = hi - gr * (hi - lo)
            fc = dist(c)[0]
        else:
            lo, c, fc = c, e, fe
            e = lo + gr * (hi - lo)
            fe = dist(e)[0]
    phi = 0.5 * (lo + hi)
    cands = [(dist(ph)[0], ph) for ph in (0.0, 1.0, phi)]
    phi = min(cands)[1]
    return dist(phi)[1], phi


def find_set_collision_avoidance(obs_sets, obs_points_sets, p0, p1, e_max=0.3):
    """Greedy nearest-first separating halfspaces between segment [p0,p1] and the obstacles.

    obs_sets: list of [A, b]; obs_points_sets: list of vertex arrays (n_v x 3).
    Returns (A (n x 3), b (n), collision flag)."""
    a_set, b_set = init_halfspaces_point(p0, e_max)
    collision = False
    remain = list(range(len(obs_sets)))
    pts, closest, dists = {}, {}, {}
    for i in remain:
        A, b = obs_sets[i]
        x, phi = closest_pair_segment_polytope(np.asarray(A), np.asarray(b) - 0.001, p0, p1)
        pts[i] = x
        closest[i] = p0 + phi * (p1 - p0)
        dists[i] = np.linalg.norm(x - closest[i])
    while remain:
        idx = min(remain, key=lambda i: dists[i])
        cp = pts[idx]
        a = cp - closest[idx]
        na = np.linalg.norm(a)
        if na < 1e-6:
            collision = True
            a = cp - p0
            na = np.linalg.norm(a)
            if na < 1e-6:
                a = p1 - p0
                na = np.linalg.norm(a)
        a = a / na
        bh = a @ cp - 0.001
        drop = [idx]
        for i in remain:
            if i != idx and np.min(obs_points_sets[i] @ a - bh) >= -1e-4:
                drop.append(i)
        remain = [i for i in remain if i not in drop]
        a_set.append(a)
        b_set.append(bh)
    return np.array(a_set), np.array(b_set), collision
