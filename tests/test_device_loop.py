"""The device-resident closed-loop step logic (boundplanner_amd/csrc/bmpc_loop.hpp) against the REFERENCE's own
closed-loop trace: the CPU build of the identical source (tests/emu/emu_loop.cpp) replays
tests/golden/closed_loop.npz -- every solver argument it prepares (start vector, bounds, the 875 parameters) and
all state it carries (split indices, switch with via-point adaptation, rotation reference, path parameter,
joint state) must equal what the reference's BoundMPC.step / compute_return_data / integrate_joint produced."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import emu_loop_lib as E
import oracle_lib as O
from boundplanner_amd.device_loop import pack_state, state_view
from boundplanner_amd.mpc_node import MPCNode
from boundplanner_amd.params import Params, get_default_params
from boundplanner_amd.robot_model import RobotModel
from test_closed_loop import ReplaySolver


def test_so3_helpers_match_scipy():
    rng = np.random.default_rng(0)
    for i in range(400):
        v = rng.normal(size=3) * rng.choice([1e-5, 1e-3, 0.3, 1.0, 2.5])
        if i % 7 == 0:
            v = v / np.linalg.norm(v) * (np.pi - 1e-3 * rng.uniform())      # near the half turn
        M = R.from_rotvec(rng.normal(size=3) * rng.choice([1e-4, 0.5, 2.0])).as_matrix()
        Rv, v2, e = E.so3(v, M)
        assert np.abs(Rv - R.from_rotvec(v).as_matrix()).max() < 1e-14
        assert np.abs(v2 - R.from_matrix(M).as_rotvec()).max() < 1e-12
        assert np.abs(e - R.from_matrix(M).as_euler("zyx")).max() < 1e-12


def compare_record_with_host_mirror(rec, N, node, k, tol=1e-9):
    """The device loop's MPCData record of a step (decoded by mpc_data.from_device_record) against mpc_data.from_node of the host
    mirror that took the same step: every field both sides produce."""
    from boundplanner_amd import mpc_data
    d, h = mpc_data.from_device_record(rec, N), mpc_data.from_node(node)
    assert d["iterations"] == h["iterations"] and d["sector"] is not None and abs(d["phi_max"] - h["phi_max"]) < tol, k
    for key in ("p", "v", "a", "q", "dq", "ddq", "dddq", "e_p", "de_p", "e_r", "de_r", "e_r_par", "e_r_orth1", "e_r_orth2", "p_ref"):
        assert len(d[key]) == len(h[key]) > 0, (k, key, len(d[key]), len(h[key]))
        worst = max(np.abs(np.ravel(a) - np.ravel(b)).max() for a, b in zip(d[key], h[key]))
        assert worst < tol, (k, key, worst)
    assert np.abs(d["phi"] - h["phi"]).max() < tol and np.abs(d["dphi"] - h["dphi"]).max() < tol, k
    for key in ("a_set", "b_set", "a_set_next", "b_set_next", "a_set_j3", "b_set_j3", "a_set_j5", "a_set_j6", "a_set_j67", "a_set_elbow", "b_set_elbow"):
        if np.size(h[key]):
            assert np.abs(np.ravel(d[key]) - np.ravel(h[key])).max() < 2e-6, (k, key)       # (collision sets: closest pairs to ~1e-7)
    assert set(mpc_data.FIELDS) <= set(d)


@pytest.mark.parametrize("fixture", ["closed_loop.npz", "closed_loop_n15.npz", "closed_loop_fail.npz", "closed_loop_patch.npz",
                                     "closed_loop_scene.npz"])
def test_replay_of_the_reference_trace(golden_dir, fixture):
    """(scenarios: see tests/test_closed_loop.py TRACES)"""
    g = np.load(os.path.join(golden_dir, fixture))
    N = int(g["N"])
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    lay = E.layout()
    shadow = MPCNode(g["in_q"][0], RobotModel(O.fk_batch), lambda n, dt: ReplaySolver(g, tol=1e-9), params=params)
    obs = None
    if "boxes" in g.files:          # scene obstacles: per-step collision sets on the "device" (a10 / f2 inside the loop)
        from boundplanner_amd import scenes
        obs = scenes.boxes_to_sets(g["boxes"])
        shadow.mpc.set_obstacle_sets(*obs)
    n_w = 44 * N + 6
    prev = np.zeros(n_w)
    pack = lambda: pack_state(lay, shadow.mpc, shadow.q, shadow.dq, shadow.ddq, shadow.jerk, shadow.qf, shadow.v, shadow.p_lie)
    S = pack()
    n_steps, n_update = g["in_q"].shape[0], int(g["n_update"])
    big = lambda a: np.nan_to_num(np.asarray(a, float), posinf=1e20, neginf=-1e20)
    worst = {}
    for k in range(n_steps):
        if k == n_update:
            # new plan: host-side BoundMPC.update / ReferencePath construction, re-serialised; the carried
            # warm start, slacks0 and error count are NOT reset (a15) and stay what the device loop holds
            shadow.update_reference([p.copy() for p in g["via_p_via"]], [r.copy() for r in g["via_r_via"]],
                                    [b.copy() for b in g["via_bp1"]], [b.copy() for b in g["via_br1"]],
                                    [e.copy() for e in g["via_e_r_bound"]], [a.copy() for a in g["via_a_sets"]],
                                    [b.copy() for b in g["via_b_sets"]], [])
            keep = {f: state_view(lay, S)[f].copy() for f in ("slacks0", "error_count", "has_prev", "q", "dq", "ddq", "jerk", "v", "p_lie")}
            S = pack()
            for f, val in keep.items():
                assert np.abs(state_view(lay, S)[f] - val).max() < 1e-9, f
                if f != "has_prev":
                    state_view(lay, S)[f][:] = val
            state_view(lay, S)["has_prev"][:] = keep["has_prev"]
        x0, lbx, ubx, p = E.prepare(N, S, prev) if obs is None else E.prepare_obs(N, S, prev, *obs)
        for name, mine in (("x0", x0), ("lbx", lbx), ("ubx", ubx), ("p", p)):
            d = np.abs(mine - big(g["call_" + name][k])).max()
            worst[name] = max(worst.get(name, 0.0), d)
            # (collision-set rows with obstacles: closest-pair search resolved to ~1e-7 on both sides)
            assert d < (2e-6 if (obs is not None and name == "p") else 1e-9), (k, name, d, np.argmax(np.abs(mine - big(g["call_" + name][k]))))
        _, rec = E.finish(N, params.dt, S, g["call_x"][k], prev, int(g["status"][k]), float(g["viol"][k]), int(g["iters"][k]), par=p)
        shadow.step()                                   # keeps the host mirror in lock step for the re-plan
        compare_record_with_host_mirror(rec, N, shadow, k)      # the MPCData record the finish logic writes (boundmpcmsg/msg/MPCData.msg)
        V = state_view(lay, S)
        assert [int(s) for s in V["split"]] == list(g["split_idxs"][k]), k
        assert int(V["sw"][0]) == int(g["switch"][k]) and int(V["error_count"][0]) == int(g["error_count"][k]), k
        assert int(V["rp_sector"][0]) == int(g["sector"][k]), k
        for f, key in (("pr_ref", "pr_ref"), ("iw_ref", "iw_ref"), ("phi_current", "phi_current"), ("dphi_current", "dphi_current"),
                       ("phi_max", "phi_max"), ("slacks0", "slacks0"), ("rp_pd", "rp_pd"), ("rp_phi_switch", "rp_phi_switch"),
                       ("q", "out_q"), ("dq", "out_dq"), ("ddq", "out_ddq"), ("jerk", "out_jerk"), ("v", "out_v"),
                       ("qf", "out_qf"), ("p_lie", "out_p_lie")):
            d = np.abs(V[f] - np.asarray(g[key][k]).reshape(-1)).max()
            worst[f] = max(worst.get(f, 0.0), d)
            assert d < 1e-9, (k, f, d)
    assert g["switch"].sum() >= 1 and g["sector"][-1] >= 1      # the trace exercises a switch + via-point adaptation
    if "fail" in fixture:
        assert g["error_count"].max() == 2 and (g["status"] != 0).sum() == 3
    if "patch" in fixture:
        assert g["patched"].sum() >= 1                            # the omega re-basing branch of the warm start ran
    print("max deviation from the reference trace:", {k: float(f"{v:.2e}") for k, v in worst.items()})


def test_failure_path_matches_host_mirror():
    """Solver failures (BoundMPC.py:619-645, Q12): the step falls back to the previous solution and shifts its outputs
    by error_count columns; a failure before any accepted solution keeps the current one.  The device logic and the host
    mirror (boundplanner_amd.bound_mpc / post, the restatement pinned by the golden trace on the success path) must
    carry identical state through a scripted sequence of failures."""
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd import scenes
    from boundplanner_amd.params import normalize_set_size
    N = 8
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)

    class FakeBackend:           # kinematics from the oracle, solves from the oracle: checker-side stand-ins
        def __init__(self):
            self.N, self.n_w = N, 44 * N + 6
            self.script = []

        def fk(self, q, dq=None):
            return O.fk_batch(q, dq)

        def solve_batch(self, x0, lbx, ubx, p, want_g=False):
            # both sides get the SAME solution (the NLP solve amplifies rounding-level argument differences in the
            # weakly determined jerk block); the host's own arguments must equal the device logic's
            args, r, fail = self.script.pop(0)
            for mine, dev in zip((x0, lbx, ubx, p), args):
                assert np.abs(mine[0] - dev).max() < 1e-9
            r = {k: np.array(v, copy=True) for k, v in r.items()}
            if fail:
                r["status"][:] = 1
                r["viol"][:] = 1.0
            return r

    be = FakeBackend()
    rng = np.random.default_rng(5)
    q_start, q_goal = scenes.sample_start_goal(rng, be.fk, 1)
    fg = be.fk(q_goal)
    host = BatchMPCNode(be, q_start, params)
    lay = E.layout()
    S = pack_state(lay, host.mpcs[0], host.q[0], host.dq[0], host.ddq[0], host.jerk[0], host.qf[0], host.v[0], host.p_lie[0])
    prev = np.zeros(44 * N + 6)
    a_ee, b_ee = scenes._box_set([-1.0, -1.0, 0.0], [1.0, 1.0, 1.2])
    # failure on the very first solve (no previous solution), successes, two failures in a row, recovery
    script = [True, False, False, True, True, False, False, True, False]
    for k, fail in enumerate(script):
        if k == 1:
            sets = normalize_set_size([[a_ee, b_ee]], 15)
            host.update_reference(0, [host.p_lie[0][:3].copy(), fg["ee_pos"][0].copy()],
                                  [R.from_rotvec(host.p_lie[0][3:]).as_matrix(), fg["ee_rot"][0].copy()], [np.array([0.0, 0, 1])],
                                  [np.array([0.0, 0, 1])], [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180], [sets[0][0]], [sets[0][1]])
            keep = {f: state_view(lay, S)[f].copy() for f in ("slacks0", "error_count", "has_prev")}
            S = pack_state(lay, host.mpcs[0], host.q[0], host.dq[0], host.ddq[0], host.jerk[0], host.qf[0], host.v[0], host.p_lie[0])
            for f, val in keep.items():
                state_view(lay, S)[f][:] = val
        x0, lbx, ubx, p = E.prepare(N, S, prev)
        r = O.solve_batch(N, x0[None], lbx[None], ubx[None], p[None])
        st, viol = (1, 1.0) if fail else (int(r["status"][0]), float(r["viol"][0]))
        E.finish(N, params.dt, S, r["x"][0], prev, st, viol)
        be.script = [((x0, lbx, ubx, p), r, fail)]
        host.step()
        m, V = host.mpcs[0], state_view(lay, S)
        assert int(V["error_count"][0]) == m.error_count, k
        assert (V["has_prev"][0] != 0) == (m.prev_solution is not None), k
        assert [int(s) for s in V["split"]] == list(m.split_idxs), k
        for f, ref in (("q", host.q[0]), ("dq", host.dq[0]), ("ddq", host.ddq[0]), ("jerk", host.jerk[0]), ("qf", host.qf[0]),
                       ("v", host.v[0]), ("p_lie", host.p_lie[0]), ("slacks0", m.slacks0), ("pr_ref", m.pr_ref), ("iw_ref", m.iw_ref),
                       ("phi_current", m.phi_current), ("dphi_current", m.dphi_current)):
            assert np.abs(V[f] - np.asarray(ref).reshape(-1)).max() < 1e-9, (k, f)
        if m.prev_solution is not None:
            assert np.abs(prev - m.prev_solution).max() < 1e-9, k
    assert host.mpcs[0].error_count == 0 and max(script) is True


def _box_scene(rng, n, rotated=0.0):
    """n boxes (as [A, b] + their 8 corners) scattered through the arm's workspace; a fraction `rotated` of them is
    turned by a random rotation (general polytopes: the projection then runs Hildreth's iteration, not the box clamp)."""
    sets, pts = [], []
    for _ in range(n):
        c = rng.uniform([-0.7, -0.7, 0.0], [0.7, 0.7, 1.1]); h = rng.uniform(0.03, 0.15, size=3)
        Q = R.from_rotvec(rng.normal(size=3)).as_matrix() if rng.uniform() < rotated else np.eye(3)
        A = np.vstack((Q.T, -Q.T))                    # rows: +-(box axes); x = c + Q u, |u| <= h
        sets.append([A, np.concatenate((Q.T @ c + h, -(Q.T @ c) + h))])
        pts.append(np.array([c + Q @ (np.array([sx, sy, sz]) * h) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]))
    return sets, pts


def test_collision_sets_with_obstacles_match_host_finder():
    """a10 on the device: per-step collision sets of the 6 collision points against scene obstacles
    (ConvexSetFinder.py:309-375) -- the 875 parameters the device logic prepares equal the host mirror's
    (BoundMPC.prepare + collision_sets.find_set_collision_avoidance, itself checked against an independent QP solve)."""
    from boundplanner_amd.bound_mpc import BoundMPC
    from boundplanner_amd.params import Q_LIM_LOWER, Q_LIM_UPPER
    N = 8
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    lay = E.layout()
    rng = np.random.default_rng(11)
    robot = RobotModel(O.fk_batch)
    n_rows_seen = set()
    for trial in range(12):
        sets, pts = _box_scene(rng, int(rng.integers(1, 9)), rotated=0.5 if trial % 2 else 0.0)
        q = rng.uniform(0.6 * Q_LIM_LOWER, 0.6 * Q_LIM_UPPER)
        qf = q + rng.normal(size=7) * rng.choice([0.0, 0.05, 0.4])         # includes the degenerate segment qf == q
        dq = rng.normal(size=7) * 0.1
        p_lie = robot.fk(q)
        mpc = BoundMPC([p_lie[:3]] * 2, [R.from_rotvec(p_lie[3:]).as_matrix()] * 2, [np.array([1.0, 0, 0])], [np.array([1.0, 0, 0])],
                       [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180], [np.zeros((15, 3))], [np.ones(15)], [], p0=p_lie,
                       params=params, robot_model=robot)
        mpc.set_obstacle_sets(sets, pts)
        v = np.concatenate((robot.velocity_ee(q, dq), robot.omega_ee(q, dq)))
        S = pack_state(lay, mpc, q, dq, np.zeros(7), np.zeros(7), qf, v, p_lie)
        try:
            _, _, _, p_host, _ = mpc.prepare(q, dq, np.zeros(7), p_lie, v, np.zeros(7), qf)
        except ValueError:                     # more than 15 rows: the host raises, the device freezes the rollout
            x0, lbx, ubx, p = E.prepare_obs(N, S, np.zeros(44 * N + 6), sets, pts)
            assert state_view(lay, S)["dead"][0] == 2.0
            continue
        x0, lbx, ubx, p = E.prepare_obs(N, S, np.zeros(44 * N + 6), sets, pts)
        assert state_view(lay, S)["dead"][0] == 0.0
        assert np.abs(p[:515] - p_host[:515]).max() < 1e-12
        # halfspace rows: the golden-section closest pair is resolved to ~1e-8 along the segment on both sides
        assert np.abs(p[515:] - p_host[515:]).max() < 1e-6, (trial, np.abs(p[515:] - p_host[515:]).max())
        a_j = p_host[515:785].reshape(6, 3, 15)
        n_rows_seen.update(int((np.abs(a_j[j]).sum(axis=0) > 0).sum()) for j in range(6))
    assert max(n_rows_seen) > 7 and min(n_rows_seen) >= 6       # scenes with several active obstacle halfspaces


def _rollout_state(lay, params, robot, q, qf):
    """A rollout at rest in configuration q whose horizon end is qf, on the trivial start-up path (MPCNode.py:44-60)."""
    from boundplanner_amd.bound_mpc import BoundMPC
    p_lie = robot.fk(q)
    mpc = BoundMPC([p_lie[:3]] * 2, [R.from_rotvec(p_lie[3:]).as_matrix()] * 2, [np.array([1.0, 0, 0])], [np.array([1.0, 0, 0])],
                   [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180], [np.zeros((15, 3))], [np.ones(15)], [], p0=p_lie,
                   params=params, robot_model=robot)
    return mpc, pack_state(lay, mpc, q, np.zeros(7), np.zeros(7), np.zeros(7), qf, np.zeros(6), p_lie), p_lie


def test_collision_sets_match_the_reference_finder(golden_dir):
    """a10 / f2, device logic (CPU build of the identical bmpc_loop.hpp): the collision-set block of the 875 parameters
    equals what the REFERENCE's finder produced on its example scene (tests/golden/colsets.npz)."""
    from boundplanner_amd import scenes
    from test_collision_sets import expected_set_params
    g = np.load(os.path.join(golden_dir, "colsets.npz"))
    N = 8
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    lay, robot = E.layout(), RobotModel(O.fk_batch)
    sets, pts = scenes.boxes_to_sets(g["boxes"])
    worst = 0.0
    for i in range(g["q0"].shape[0]):
        _, S, _ = _rollout_state(lay, params, robot, g["q0"][i], g["qf"][i])
        _, _, _, p = E.prepare_obs(N, S, np.zeros(44 * N + 6), sets, pts)
        assert state_view(lay, S)["dead"][0] == 0.0
        a_ref, b_ref, b_ok = expected_set_params(g, i)
        worst = max(worst, np.abs(p[515:785] - a_ref).max(), np.abs(p[785:875] - b_ref)[b_ok].max())
    assert worst < 1e-6, worst
    ssets, spts = scenes.boxes_to_sets(g["shell_boxes"])
    _, S, _ = _rollout_state(lay, params, robot, g["shell_q"][0], g["shell_q"][0])
    E.prepare_obs(N, S, np.zeros(44 * N + 6), ssets, spts)
    assert state_view(lay, S)["dead"][0] == 2.0            # 20 halfspaces do not fit max_set_size: the rollout is frozen
