"""Collision-free convex sets for the plan phase: IRIS-style polyhedron growth around a seed point with a maximum-volume
inscribed ellipsoid, and the segment-based set with its ellipsoid.

Mirrors the planner-side interface of the reference's ConvexSetFinder
(/root/reference/bound_planner/BoundPlanner/ConvexSetFinder.py): find_set_around_point (:190-240), compute_polyhedron
(:423-469), compute_set_projs (:471-489), init_halfspaces (:377-398), find_set_collision_avoidance(..., compute_ellipsoid=True)
(:309-375), mvie_socp / mvie_socp_fixed_mid (:512-560).  The per-step variant (limit_space=True) lives in collision_sets.py
and on the device (csrc/bmpc_loop.hpp).  The third-party solvers of the reference are replaced by planner_opt.py.
"""
import numpy as np

from . import planner_opt as PO
from .collision_sets import closest_pair_segment_polytope


def _inv_sym(q_inv):
    """q_ellipse = q_inv^-1 through the SVD, as the reference inverts it (ConvexSetFinder.py:223-224)."""
    u, s, vh = np.linalg.svd(q_inv)
    return vh.T @ np.diag(1.0 / s) @ u.T, s


class ConvexSetFinder:
    def __init__(self, obs_sets, obs_points_sets, e_max, e_min):
        self.obs_sets = [[np.asarray(a, float), np.asarray(b, float)] for a, b in obs_sets]
        self.obs_points_sets = [np.asarray(p, float) for p in obs_points_sets]
        self.e_max, self.e_min = np.asarray(e_max, float), np.asarray(e_min, float)
        self.max_iter = 5
        self.ell_time = self.proj_time = 0.0

    # -- workspace box: rows +x, -x, +y, -y, +z, -z
    def init_halfspaces(self):
        a, b = [], []
        for i in range(3):
            e = np.eye(3)[i]
            a += [e.copy(), -e]
            b += [float(self.e_max[i]), float(-self.e_min[i])]
        return a, b

    # -- closest point of every obstacle to p0 in the metric of the ellipsoid p0 + E u
    def compute_set_projs(self, obs_sets, p0, ellipse_mat):
        pts = np.empty((len(obs_sets), 3))
        for i, (a_set, b_set) in enumerate(obs_sets):
            u = PO.project_polytope(a_set @ ellipse_mat, b_set - a_set @ p0, np.zeros(3))
            pts[i] = ellipse_mat @ u + p0
        return pts

    # -- one IRIS step: separating halfspaces tangent to the inflated ellipsoid, nearest obstacle first; obstacles that lie
    #    entirely behind a chosen halfspace are dropped
    def compute_polyhedron(self, q_inv, q_ellipse, p_seed, a_init, b_init):
        remain = list(range(len(self.obs_sets)))
        a_set, b_set = list(a_init), list(b_init)
        pts = self.compute_set_projs(self.obs_sets, p_seed, q_inv)
        dist = np.linalg.norm(q_ellipse @ (pts - p_seed).T, axis=0)
        while remain:
            idx = min(remain, key=lambda i: dist[i])
            if dist[idx] < 0.99:
                raise RuntimeError("Ellipse violates constraints")
            cp = pts[idx]
            a = 2 * (q_ellipse @ q_ellipse.T) @ (cp - p_seed)
            bh = a @ cp
            na = np.linalg.norm(a)
            a, bh = a / na, bh / na
            drop = [idx] + [i for i in remain if i != idx and np.min(self.obs_points_sets[i] @ a - bh) >= -1e-4]
            remain = [i for i in remain if i not in drop]
            a_set.append(a); b_set.append(bh)
        return a_set, b_set

    def mvie_socp(self, a_set, b_set):
        return PO.mvie(a_set, b_set)

    def mvie_socp_fixed_mid(self, a_set, b_set, p_mid):
        return PO.mvie(a_set, b_set, fixed_mid=p_mid)

    def find_set_around_point(self, p_seed, fixed_mid=False, optimize=True):
        """Alternate polyhedron growth and ellipsoid inflation from a tiny ball at p_seed until the ellipsoid volume settles
        (1 %), at most max_iter rounds.  Returns (A, b, q_ellipse, centre)."""
        p_seed = np.array(p_seed, float)
        q_inv = np.diag([1e-4] * 3)
        q_ellipse = np.diag([1e4] * 3)
        a_init, b_init = self.init_halfspaces()
        det_old, det = 1.0, 100.0
        k = 0
        while abs(det - det_old) / det_old > 0.01:
            k += 1
            if k > self.max_iter:
                break
            a_set, b_set = self.compute_polyhedron(q_inv, q_ellipse, p_seed, a_init, b_init)
            a_np, b_np = np.array(a_set), np.array(b_set)
            if not optimize:
                return a_np, b_np, q_ellipse, p_seed
            det_old = det
            if fixed_mid:
                q_inv, p_seed = self.mvie_socp_fixed_mid(a_np, b_np, p_seed)
            else:
                q_inv, p_seed = self.mvie_socp(a_np, b_np)
            q_ellipse, sv = _inv_sym(q_inv)
            det = np.linalg.det(q_ellipse)
            if np.min(sv) < 1e-3:          # the ellipsoid collapsed (fixed centre on a face)
                break
        if fixed_mid:
            q_inv, p_seed = self.mvie_socp(a_np, b_np)
            q_ellipse, _ = _inv_sym(q_inv)
        return a_np, b_np, q_ellipse, p_seed

    def find_set_collision_avoidance(self, p0, p1, compute_ellipsoid=False):
        """Set around the segment [p0, p1] inside the workspace box: nearest-first separating halfspaces between the segment and
        the obstacles (each shrunk by 1 mm).  Returns (A, b[, q_ellipse, centre], collision)."""
        p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
        a_set, b_set = self.init_halfspaces()
        collision = False
        remain = list(range(len(self.obs_sets)))
        pts, closest, dist = {}, {}, {}
        for i in remain:
            x, phi = closest_pair_segment_polytope(self.obs_sets[i][0], self.obs_sets[i][1] - 0.001, p0, p1)
            pts[i], closest[i] = x, p0 + phi * (p1 - p0)
            dist[i] = np.linalg.norm(x - closest[i])
        while remain:
            idx = min(remain, key=lambda i: dist[i])
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
            drop = [idx] + [i for i in remain if i != idx and np.min(self.obs_points_sets[i] @ a - bh) >= -1e-4]
            remain = [i for i in remain if i not in drop]
            a_set.append(a); b_set.append(bh)
        a_np, b_np = np.array(a_set), np.array(b_set)
        if compute_ellipsoid:
            q_inv, p_mid = self.mvie_socp(a_np, b_np)
            q_ellipse, _ = _inv_sym(q_inv)
            return a_np, b_np, q_ellipse, p_mid, collision
        return a_np, b_np, collision
