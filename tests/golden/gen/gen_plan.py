"""Generate tests/golden/plan.npz: plans produced by the REFERENCE's own planner logic.

Runs, in the build container only, the reference's unmodified BoundPlanner.plan_convex_set_path
(/root/reference/bound_planner/BoundPlanner/BoundPlanner.py:174-584 with compute_via_points :586-743, add_edges :789-896,
check_intersection / set_intersection :745-787) and ConvexSetFinder.find_set_around_point / find_set_collision_avoidance /
compute_polyhedron (ConvexSetFinder.py:190-240, 309-375, 423-469) on its example scene (boundplanner_with_mpc_example.py:38-111)
and on two smaller scenes, under the import stubs of tests/golden/gen/stubs.  The third-party slots on this path -- none of
them available in this image -- are filled by the small exact solvers of boundplanner_amd/planner_opt.py:
    qpOASES / OSQP qpsol objects (projections, "end effector fits")    -> project_polytope, fits
    cvxpy + Clarabel MVIE problems (socp_prob / socpfm_prob .solve())   -> mvie
    pycddlib (compute_polytope_vertices, reduce_ineqs)                  -> polytope_vertices, reduce_ineqs
    IPOPT via-point / rotation NLP (via_point_rot_optimization_problem) -> via_rot_problem  [*]
so the fixture pins the planner LOGIC of this repository (graph construction, set growth, via-point selection) against
the reference's; the sub-problem solvers are validated against scipy in tests/test_planner.py.
[*] that NLP's FORMULATION is pinned separately (round 4): tests/golden/gen/gen_via_rot.py builds the reference's own problem under the
numeric casadi stand-in (its ca.jacobian / ca.Function calls are served by re-evaluation) and tests/golden/via_rot.npz holds f and g at
sample points, which planner_opt.via_rot_reference_fg reproduces; via_rot_problem solves that problem with phi_max eliminated.
The fixture is data only; no reference source is copied.

    python tests/golden/gen/gen_plan.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
OUT = os.path.abspath(os.path.join(HERE, ".."))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(1, REF)
sys.path.insert(2, os.path.join(ROOT, "tests"))
sys.path.insert(3, ROOT)
os.chdir(REF)

import oracle_lib as O  # noqa: E402
from boundplanner_amd import planner_opt as PO, scenes  # noqa: E402
from boundplanner_amd.collision_sets import closest_pair_segment_polytope  # noqa: E402
import bound_planner.BoundPlanner  # noqa: E402,F401
BPM = sys.modules["bound_planner.BoundPlanner.BoundPlanner"]      # the module (the package attribute of that name is the class)
from scipy.spatial.transform import Rotation as R  # noqa: E402


class DM(np.ndarray):
    """numpy array with CasADi's .full()"""
    def __new__(cls, a):
        return np.asarray(a, float).view(cls)

    def full(self):
        return np.asarray(self).reshape(-1, 1)


class QpSlot:
    """Stands in for a CasADi qpsol function object (call convention of BoundPlanner.py:847-854, ConvexSetFinder.py:478-485)."""
    def __init__(self, fn):
        self.fn, self.ok = fn, True

    def __call__(self, x0, lbx, ubx, lbg, ubg, p):
        x, self.ok = self.fn(np.asarray(p, float).ravel(), np.asarray(x0, float).ravel())
        return {"x": DM(x), "g": DM(np.zeros(1))}

    def stats(self):
        return {"success": self.ok}


def _split(p, S):
    return p[:3 * S].reshape(3, S).T, p[3 * S:4 * S]


def proj20(p, x0):          # optimization_functions.py:107-137: params = (a_set 20x3 column-major, b_set, xd)
    A, b = _split(p, 20)
    return PO.project_polytope(A, b, p[80:83]), True


def proj15(p, x0):          # ConvexSetFinder.py:10-49: params = (a_set 15x3, b_set); min |x|^2
    A, b = _split(p, 15)
    return PO.project_polytope(A, b, np.zeros(3)), True


def projl15(p, x0):         # ConvexSetFinder.py:52-99: closest pair of a segment and a polytope
    A, b = _split(p, 15)
    x, phi = closest_pair_segment_polytope(A, b, p[60:63], p[63:66])
    return np.concatenate((x, [phi])), True


def fit20(p, x0):           # optimization_functions.py:140-185: params = (l_ee, a_set 20x3, b_set)
    A, b = _split(p[3:], 20)
    return x0, PO.fits(A, b, p[:3])


class MvieSlot:
    """Stands in for the cvxpy Problem of ConvexSetFinder.cvx_mvie_socp / _fixed_mid: .solve() reads the Parameter values the
    caller has just set (a: per-row x_size x 3, c, d) and leaves the optimum in params["x"].value."""
    def __init__(self, params, x_size):
        self.params, self.x_size = params, x_size

    def solve(self, **kw):
        a = np.array([self.params["a"][i].value for i in range(20)])       # [20][x_size][3]
        A = np.stack((a[:, 0, 0], a[:, 1, 0], a[:, 3, 0]), axis=1)            # a2[[0, 1, 3], 0, i] = a_set[i] (ConvexSetFinder.py:519)
        d = np.asarray(self.params["d"].value, float)
        keep = np.abs(A).sum(axis=1) > 0
        x = np.zeros(self.x_size)
        if self.x_size == 12:
            q, c = PO.mvie(A[keep], d[keep])
            x[6:9] = c
        else:                                                               # fixed mid: d = b - A p_mid, centre at the origin
            q, _ = PO.mvie(A[keep], d[keep], fixed_mid=np.zeros(3))
        L = np.linalg.cholesky(q)
        x[:6] = L[np.tril_indices(3)]
        self.params["x"].value = x


class ViaRotSlot:
    """Stands in for the IPOPT nlpsol object of optimization_functions.py:385 (call convention of BoundPlanner.py:668-676)."""
    def __init__(self, nr_via, max_set_size):
        self.nr_via, self.S, self.ok = nr_via, max_set_size, True

    def __call__(self, x0, lbx, ubx, lbg, ubg, p):
        x, self.ok = PO.via_rot_problem(self.nr_via, self.S, np.asarray(x0, float).ravel(), np.asarray(p, float).ravel())
        return {"x": DM(x), "g": DM(np.zeros(1))}

    def stats(self):
        return {"success": self.ok}


def via_rot_factory(nr_via, max_set_size):
    return ViaRotSlot(nr_via, max_set_size), [0.0], [0.0], [0.0], [0.0]


def make_reference_planner(obstacles, ws_max, ws_min):
    BPM.via_point_rot_optimization_problem = via_rot_factory
    BPM.compute_polytope_vertices = lambda a, b: list(PO.polytope_vertices(a, b))
    BPM.reduce_ineqs = lambda a, b: PO.reduce_ineqs(a, b)
    pl = BPM.BoundPlanner(e_p_max=0.5, obstacles=[list(o) for o in obstacles], workspace_max=list(ws_max), workspace_min=list(ws_min))
    pl.rng = np.random.default_rng(7)        # the reference's generator is unseeded (BoundPlanner.py:50)
    pl.proj_solver = QpSlot(proj20)
    pl.solver_fit = QpSlot(fit20)
    f = pl.set_finder
    f.proj_solver = QpSlot(proj15)
    f.projl_solver = QpSlot(projl15)
    f.socp_prob = MvieSlot(f.socp_params, 12)
    f.socpfm_prob = MvieSlot(f.socpfm_params, 9)
    return pl


def cases():
    boxes, q0, goal_p, goal_r = scenes.example_scene()
    fk = O.fk_batch(q0[None])
    p0, r0 = fk["ee_pos"][0], fk["ee_rot"][0]
    out = [dict(name="example", boxes=boxes, ws_max=[1.0, 0.38, 1.0], ws_min=[-0.14, -1.0, 0.0], start=p0, end=goal_p, r0=r0, r1=goal_r)]
    # a wall between start and goal: two sets, one via point
    wall = np.array([[0.35, -0.05, 0.0, 0.65, 0.05, 0.45]])
    out.append(dict(name="wall", boxes=wall, ws_max=[1.0, 1.0, 1.2], ws_min=[-1.0, -1.0, 0.0], start=np.array([0.5, 0.35, 0.3]),
                    end=np.array([0.5, -0.35, 0.3]), r0=r0, r1=R.from_euler("XYZ", [0, 120, 20], degrees=True).as_matrix()))
    # nothing in the way (one far obstacle: the reference's finder needs at least one): the goal is inside the start set
    out.append(dict(name="free", boxes=np.array([[-0.9, 0.8, 0.0, -0.8, 0.9, 0.1]]), ws_max=[1.0, 1.0, 1.2], ws_min=[-1.0, -1.0, 0.0], start=np.array([0.4, 0.2, 0.5]),
                    end=np.array([0.5, -0.2, 0.4]), r0=r0, r1=goal_r))
    return out


def main():
    save = {}
    names = []
    for c in cases():
        pl = make_reference_planner(c["boxes"], c["ws_max"], c["ws_min"])
        p_via, r_via, bp1, sets = pl.plan_convex_set_path(c["start"].copy(), c["end"].copy(), c["r0"], c["r1"])
        n = c["name"]; names.append(n)
        print(n, "via points", len(p_via), "sets", len(sets), "graph sets", pl.nr_sets)
        for k in ("boxes", "ws_max", "ws_min", "start", "end", "r0", "r1"):
            save[f"{n}_{k}"] = np.asarray(c[k], float)
        save[f"{n}_p_via"] = np.array(p_via); save[f"{n}_r_via"] = np.array(r_via); save[f"{n}_bp1"] = np.array(bp1)
        save[f"{n}_A"] = np.array([s[0] for s in sets]); save[f"{n}_b"] = np.array([s[1] for s in sets])
        save[f"{n}_nr_sets"] = np.array(pl.nr_sets)
    # replanning on the example scene: halfway along the first segment, with a horizon of points ahead on the old path
    c = cases()[0]
    pl = make_reference_planner(c["boxes"], c["ws_max"], c["ws_min"])
    p_via, r_via, bp1, sets = pl.plan_convex_set_path(c["start"].copy(), c["end"].copy(), c["r0"], c["r1"])
    p_via = [np.asarray(p) for p in p_via]
    s0 = p_via[0] + 0.5 * (p_via[1] - p_via[0])
    horizon = [s0 + t * (p_via[1] - s0) for t in np.linspace(0.0, 0.6, 8)]
    p2, r2, bp2, sets2 = pl.plan_convex_set_path(s0.copy(), c["end"].copy(), np.asarray(r_via[0]), c["r1"], replanning=True, p_horizon=horizon)
    print("replan: via points", len(p2), "replanning_phi", pl.replanning_phi)
    save.update(replan_start=s0, replan_horizon=np.array(horizon), replan_r0=np.asarray(r_via[0]), replan_p_via=np.array(p2), replan_r_via=np.array(r2),
                replan_bp1=np.array(bp2), replan_A=np.array([s[0] for s in sets2]), replan_b=np.array([s[1] for s in sets2]),
                replan_phi=np.array(pl.replanning_phi))
    save["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "plan.npz"), **save)


if __name__ == "__main__":
    main()
