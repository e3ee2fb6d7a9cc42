"""Row a10 of SURVEY.md section 8: the per-step collision-avoidance sets
(ConvexSetFinder.find_set_collision_avoidance, /root/reference/bound_planner/BoundPlanner/ConvexSetFinder.py:309-375,
with the segment <-> polytope closest-pair QP of :52-99, 491-510 that the reference hands to qpOASES).

qpOASES is not available here; the QP  min |p0 + phi (p1 - p0) - x|^2  s.t.  A x <= b - 0.001, 0 <= phi <= 1
is strictly convex in the separation vector, so any correct solver returns the same distance and the same
halfspace normal.  The restatement is checked against an independent solve (scipy SLSQP) and against the
defining properties of the greedy halfspace selection."""
import numpy as np
from scipy.optimize import minimize

from boundplanner_amd.collision_sets import (closest_pair_segment_polytope, find_set_collision_avoidance,
                                             init_halfspaces_point)


def _box(lo, hi):
    A = np.vstack((np.eye(3), -np.eye(3)))
    b = np.concatenate((hi, -lo))
    V = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    return A, b, V


def _qp_reference(A, b, p0, p1):
    f = lambda u: np.sum((p0 + u[3] * (p1 - p0) - u[:3]) ** 2)
    cons = [{"type": "ineq", "fun": lambda u: b - A @ u[:3]}]
    best = None
    for phi0 in (0.0, 0.5, 1.0):
        x0 = np.concatenate((np.linalg.lstsq(A, b - 0.05, rcond=None)[0], [phi0]))
        r = minimize(f, x0, constraints=cons, bounds=[(None, None)] * 3 + [(0, 1)], method="SLSQP",
                     options={"ftol": 1e-14, "maxiter": 300})
        if best is None or r.fun < best.fun:
            best = r
    return best.x[:3], best.x[3], np.sqrt(best.fun)


def test_closest_pair_matches_independent_qp():
    rng = np.random.default_rng(3)
    for _ in range(40):
        lo = rng.uniform(-0.5, 0.3, 3); hi = lo + rng.uniform(0.05, 0.4, 3)
        A, b, _ = _box(lo, hi)
        p0, p1 = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        x, phi = closest_pair_segment_polytope(A, b - 0.001, p0, p1)
        xr, phir, dr = _qp_reference(A, b - 0.001, p0, p1)
        d = np.linalg.norm(p0 + phi * (p1 - p0) - x)
        assert np.all(A @ x <= b - 0.001 + 1e-9) and -1e-12 <= phi <= 1 + 1e-12
        assert abs(d - dr) < 1e-6, (d, dr)
        if dr > 1e-3:       # separated: the unit normal of the halfspace is unique
            n = (x - (p0 + phi * (p1 - p0))) / d
            nr = (xr - (p0 + phir * (p1 - p0))) / dr
            assert np.abs(n - nr).max() < 1e-4


def test_collision_set_properties():
    rng = np.random.default_rng(5)
    for _ in range(25):
        n_obs = int(rng.integers(1, 9))
        obs, pts = [], []
        p0, p1 = rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.3, 0.3, 3)
        while len(obs) < n_obs:
            lo = rng.uniform(-1.0, 0.8, 3); hi = lo + rng.uniform(0.05, 0.3, 3)
            A, b, V = _box(lo, hi)
            x, phi = closest_pair_segment_polytope(A, b, p0, p1)
            if np.linalg.norm(p0 + phi * (p1 - p0) - x) > 0.02:      # keep the segment free
                obs.append([A, b]); pts.append(V)
        a, bb, collision = find_set_collision_avoidance(obs, pts, p0, p1, e_max=0.7)
        assert not collision
        a0, b0 = init_halfspaces_point(p0, 0.7)
        assert np.array_equal(a[:6], np.array(a0)) and np.array_equal(bb[:6], np.array(b0))      # box rows first
        assert 6 < a.shape[0] <= 6 + n_obs
        assert np.abs(np.linalg.norm(a[6:], axis=1) - 1).max() < 1e-12                          # unit normals
        # the whole segment is inside the set, every obstacle is outside at least one halfspace (minus 1e-4)
        for ph in np.linspace(0, 1, 11):
            assert np.all(a @ (p0 + ph * (p1 - p0)) <= bb + 1e-9)
        for V in pts:
            # (the defining obstacle touches its own halfspace up to the 1 mm shrink of the QP, ConvexSetFinder.py:495)
            assert any(np.min(V @ a[r] - bb[r]) >= -2e-3 for r in range(6, a.shape[0]))
        # obstacle-free: exactly the box
        a, bb, _ = find_set_collision_avoidance([], [], p0, p1, e_max=0.7)
        assert a.shape == (6, 3)
