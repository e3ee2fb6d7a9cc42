"""Plan phase: boundplanner_amd.bound_planner / convex_set_finder / planner_opt.

tests/golden/plan.npz holds plans produced by the reference's own BoundPlanner.plan_convex_set_path and ConvexSetFinder run
under the import stubs (tests/golden/gen/gen_plan.py) with its third-party slots filled by planner_opt's small solvers: the
planner LOGIC here must reproduce them.  The small solvers themselves are checked against scipy below."""
import os

import numpy as np
import pytest
from scipy.optimize import minimize

from boundplanner_amd import planner_opt as PO
from boundplanner_amd.bound_planner import BoundPlanner


@pytest.fixture(scope="module")
def plan(golden_dir):
    return np.load(os.path.join(golden_dir, "plan.npz"))


def _planner(d, n):
    return BoundPlanner(obstacles=d[f"{n}_boxes"], e_p_max=0.5, workspace_max=d[f"{n}_ws_max"], workspace_min=d[f"{n}_ws_min"], seed=7)


@pytest.mark.parametrize("name", ["example", "wall", "free"])
def test_plan_matches_the_reference_planner(plan, name):
    d = plan
    pl = _planner(d, name)
    p_via, r_via, bp1, sets = pl.plan_convex_set_path(d[f"{name}_start"], d[f"{name}_end"], d[f"{name}_r0"], d[f"{name}_r1"])
    assert len(p_via) == d[f"{name}_p_via"].shape[0] and pl.nr_sets == int(d[f"{name}_nr_sets"])
    assert np.abs(np.array(p_via) - d[f"{name}_p_via"]).max() < 1e-6
    assert np.abs(np.array(r_via) - d[f"{name}_r_via"]).max() < 1e-6
    assert np.abs(np.array(bp1) - d[f"{name}_bp1"]).max() < 1e-6
    assert np.abs(np.array([s[0] for s in sets]) - d[f"{name}_A"]).max() < 1e-6
    assert np.abs(np.array([s[1] for s in sets]) - d[f"{name}_b"]).max() < 1e-6
    # what the tracker needs from a plan (BoundMPC.update): consecutive via points share a set, padded rows are (0, 10)
    for k, (a, b) in enumerate(sets):
        assert a.shape == (15, 3) and b.shape == (15,)
        for p in (p_via[k], p_via[k + 1]):
            assert (a @ p - b).max() < 2e-3
    # ... and the path clears the (inflated) obstacles
    for a, b in pl.obs_sets:
        for k in range(len(p_via) - 1):
            for t in np.linspace(0, 1, 21):
                p = p_via[k] + t * (p_via[k + 1] - p_via[k])
                assert (a @ p - b).max() > -1e-9


def test_replanning_matches_the_reference_planner(plan):
    d = plan
    pl = _planner(d, "example")
    pl.plan_convex_set_path(d["example_start"], d["example_end"], d["example_r0"], d["example_r1"])
    p_via, r_via, bp1, sets = pl.plan_convex_set_path(d["replan_start"], d["example_end"], d["replan_r0"], d["example_r1"], replanning=True,
                                                      p_horizon=list(d["replan_horizon"]))
    assert abs(pl.replanning_phi - float(d["replan_phi"])) < 1e-9
    assert np.abs(np.array(p_via) - d["replan_p_via"]).max() < 1e-6
    assert np.abs(np.array(r_via) - d["replan_r_via"]).max() < 1e-6
    assert np.abs(np.array([s[1] for s in sets]) - d["replan_b"]).max() < 1e-6


def _rand_poly(rng, n):
    A = rng.normal(size=(n, 3)); A /= np.linalg.norm(A, axis=1)[:, None]
    return np.vstack((np.eye(3), -np.eye(3), A)), np.concatenate((np.ones(6), rng.uniform(0.2, 1.0, n)))


def test_projection_is_the_qp_solution():
    rng = np.random.default_rng(0)
    for _ in range(25):
        A, b = _rand_poly(rng, 10)
        y = rng.normal(size=3) * 2
        x = PO.project_polytope(A, b, y)
        r = minimize(lambda z: (z - y) @ (z - y), np.zeros(3), jac=lambda z: 2 * (z - y), method="SLSQP", options={"ftol": 1e-14},
                     constraints=[{"type": "ineq", "fun": lambda z: b - A @ z, "jac": lambda z: -A}])
        assert np.abs(x - r.x).max() < 1e-6 and (A @ x - b).max() < 1e-9
        # scale invariance (the ellipsoid-metric projections of compute_set_projs scale the rows by 1e-4)
        assert np.abs(PO.project_polytope(1e-4 * A, 1e-4 * b, y) - x).max() < 1e-9


def test_mvie_is_the_optimum_of_the_reference_socp():
    """Objective (L00 L11^2 L22)^(1/4) of ConvexSetFinder.py:790-810, constraints |L^T a_i| <= b_i - a_i . c (:512-537)."""
    rng = np.random.default_rng(1)
    for _ in range(5):
        A, b = _rand_poly(rng, 8)
        q, c = PO.mvie(A, b)
        L = np.linalg.cholesky(q)
        obj = lambda x: -(0.25 * np.log(x[0]) + 0.5 * np.log(x[2]) + 0.25 * np.log(x[5]))

        def con(x):
            Lm = np.zeros((3, 3)); Lm[np.tril_indices(3)] = x[:6]
            return b - A @ x[6:9] - np.linalg.norm(A @ Lm, axis=1)
        x0 = np.zeros(9); x0[[0, 2, 5]] = 0.05; x0[6:9] = PO.chebyshev_center(A, b)[0]
        r = minimize(obj, x0, constraints=[{"type": "ineq", "fun": con}], method="SLSQP", options={"ftol": 1e-14, "maxiter": 500},
                     bounds=[(1e-6, None), (None, None), (1e-6, None), (None, None), (None, None), (1e-6, None)] + [(None, None)] * 3)
        xm = np.concatenate((L[np.tril_indices(3)], c))
        assert con(xm).min() > -1e-9 and abs(obj(xm) - r.fun) < 1e-7 and np.abs(c - r.x[6:9]).max() < 1e-5
    # the cube: the inscribed ball
    A = np.vstack((np.eye(3), -np.eye(3))); b = np.full(6, 0.5)
    q, c = PO.mvie(A, b)
    assert np.abs(q - 0.25 * np.eye(3)).max() < 1e-7 and np.abs(c).max() < 1e-7
    q, c = PO.mvie(A, b, fixed_mid=np.array([0.2, 0.0, 0.0]))
    assert np.abs(c - [0.2, 0, 0]).max() == 0 and (np.linalg.eigvalsh(q) > 0).all()


def test_polytope_bookkeeping():
    A = np.vstack((np.eye(3), -np.eye(3))); b = np.array([1, 2, 3, 0, 0, 0.])
    v = PO.polytope_vertices(A, b)
    assert v.shape == (8, 3) and {tuple(x) for x in np.round(v, 9)} == {(x, y, z) for x in (0, 1) for y in (0, 2) for z in (0, 3)}
    A2 = np.vstack((A, [[1, 0, 0]], [[0, 0, 0]], [[1, 1, 0]])); b2 = np.concatenate((b, [5, 10, 2.5]))
    Ar, br = PO.reduce_ineqs(A2, b2)
    assert Ar.shape[0] == 7 and np.allclose(Ar[-1], [1, 1, 0])          # the far face and the zero row go, the cut stays
    assert PO.fits(A, b, np.array([0.5, 0, 0])) and not PO.fits(A, b, np.array([1.5, 0, 0]))


def test_plan_feeds_the_tracker(plan):
    """plan -> BoundMPC.update: the outputs have the shapes and meaning update_reference takes
    (boundplanner_with_mpc_example.py:125-135)."""
    from boundplanner_amd.bound_mpc import BoundMPC
    from boundplanner_amd.params import Params, get_default_params
    d = plan
    pl = _planner(d, "example")
    p_via, r_via, bp1, sets = pl.plan_convex_set_path(d["example_start"], d["example_end"], d["example_r0"], d["example_r1"])
    n = len(bp1)
    assert len(p_via) == n + 1 == len(r_via) == len(sets) + 1
    base = get_default_params()
    prm = Params(n=10, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    erb = [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180] * n
    from scipy.spatial.transform import Rotation as R
    p0 = np.concatenate((p_via[0], R.from_matrix(r_via[0]).as_rotvec()))
    mpc = BoundMPC([p_via[0].copy(), p_via[0].copy()], [r_via[0].copy(), r_via[0].copy()], [np.array([1.0, 0, 0])], [np.array([1.0, 0, 0])],
                   [erb[0].copy()], [np.zeros((15, 3))], [np.ones(15)], [], p0=p0, params=prm)
    mpc.update([p.copy() for p in p_via], [r.copy() for r in r_via], [b.copy() for b in bp1], [np.array([0, 0, 1.0])] * n, [e.copy() for e in erb],
               [s[0] for s in sets], [s[1] for s in sets], [], np.zeros(6), p0=p0, params=prm)
    assert mpc.phi_max > 0.5


def test_via_rot_formulation_matches_the_reference(golden_dir):
    """The via-point / rotation NLP (optimization_functions.py:227-387): objective and every constraint row of the reference's own
    problem construction at sample points (tests/golden/via_rot.npz, generated by tests/golden/gen/gen_via_rot.py from the
    unmodified reference under the numeric casadi stand-in) equal planner_opt.via_rot_reference_fg; and the form via_rot_problem
    solves -- phi_max eliminated -- is that problem: with phi_max at the stationary point of its row's sweep the reference's
    stationarity rows vanish and its swept rows are the eliminated constraints."""
    from boundplanner_amd import planner_opt as PO
    d = np.load(os.path.join(golden_dir, "via_rot.npz"))
    n_stat = n_stat_active = 0
    for tag in ("a", "b", "c"):
        nr_via, S = int(d[f"{tag}_nr_via"]), int(d[f"{tag}_S"])
        assert d[f"{tag}_g"].shape[1] == 4 * S * nr_via + 2 * S == len(d[f"{tag}_lbg"])
        for x, p, f, g in zip(d[f"{tag}_x"], d[f"{tag}_p"], d[f"{tag}_f"], d[f"{tag}_g"]):
            f2, g2 = PO.via_rot_reference_fg(nr_via, S, x, p)
            assert abs(f2 - f) <= 1e-12 * max(1.0, abs(f))
            assert np.abs(g2 - g).max() <= 1e-9, (tag, np.abs(g2 - g).argmax())
            for i in range(nr_via):
                st = g[4 * S * i + S:4 * S * i + 3 * S:2]
                n_stat += S; n_stat_active += int((st != 0).sum())
            # eliminated form: put every phi_max at the stationary point (or leave it where the sweep is monotone)
            step = 4 + S
            xs = x.copy()
            pp, op = p[0:3], 0.0
            o = 13 + nr_via + 1 + 4 * S * nr_via
            for i in range(nr_via):
                Av = p[o + 4 * S * i:o + 4 * S * i + 3 * S].reshape(3, S).T
                P, O = x[step * i:step * i + 3], x[step * i + 3]
                for j in range(S):
                    phi = PO._sweep_row(Av[j], pp, P, op, O, p[6:9], p[9:12], p[12])
                    if phi is not None:
                        xs[step * i + 4 + j] = phi
                pp, op = P, O
            _, gs = PO.via_rot_reference_fg(nr_via, S, xs, p)
            for i in range(nr_via):
                assert np.abs(gs[4 * S * i + S:4 * S * i + 3 * S:2]).max() <= 1e-9      # stationarity rows at the eliminated phi_max
    assert 0 < n_stat_active < n_stat        # both cases occur in the fixture: interior stationary point / monotone sweep
