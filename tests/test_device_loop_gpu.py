"""The device-resident closed loop on the MI355X (C ABI bmpc_loop_*): the HIP kernels replay the reference's
closed-loop trace, and the full device loop (prepare kernel -> batched HIP solve -> finish kernel) tracks the
host loop (BatchMPCNode: the pinned per-instance host mirror around the same HIP solver)."""
import os

import numpy as np
import pytest

from boundplanner_amd.params import Params, get_default_params

pytestmark = pytest.mark.gpu


def _params(N):
    base = get_default_params()
    return Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)


@pytest.mark.parametrize("fixture", ["closed_loop.npz", "closed_loop_n15.npz", "closed_loop_fail.npz", "closed_loop_patch.npz",
                                     "closed_loop_scene.npz"])
def test_kernels_replay_the_reference_trace(golden_dir, fixture):
    import oracle_lib as O
    from boundplanner_amd.device_loop import DeviceLoop, state_view
    from boundplanner_amd.mpc_node import MPCNode
    from boundplanner_amd.robot_model import RobotModel
    from boundplanner_amd.solver import HipBoundMPC
    from test_closed_loop import ReplaySolver
    g = np.load(os.path.join(golden_dir, fixture))
    N = int(g["N"])
    params = _params(N)
    be = HipBoundMPC(N)
    R = 3                                     # identical rollouts: the result may not depend on the batch slot
    loop = DeviceLoop(be, R)
    shadow = MPCNode(g["in_q"][0], RobotModel(O.fk_batch), lambda n, dt: ReplaySolver(g, tol=1e-9), params=params)
    has_obs = "boxes" in g.files
    if has_obs:                     # BASELINE configs[0] scene: collision sets per step by bmpc_loop_k_colpairs + prepare
        from boundplanner_amd import scenes
        obs = scenes.boxes_to_sets(g["boxes"])
        shadow.mpc.set_obstacle_sets(*obs)
        loop.set_obstacles(*obs)
    for r in range(R):
        loop.set_rollout(r, shadow.mpc, shadow.q, shadow.dq, shadow.ddq, shadow.jerk, shadow.qf, shadow.v, shadow.p_lie)
    loop.upload()
    loop.set_record([0, 2])
    from test_device_loop import compare_record_with_host_mirror
    n_steps, n_update = g["in_q"].shape[0], int(g["n_update"])
    big = lambda a: np.nan_to_num(np.asarray(a, float), posinf=1e20, neginf=-1e20)
    for k in range(n_steps):
        if k == n_update:
            # new plan: host-side BoundMPC.update / ReferencePath construction, re-serialised; warm start, slacks0 and
            # error count are NOT reset (a15) and stay what the device holds
            V = loop.download()
            keep = {f: V[f].copy() for f in ("slacks0", "error_count", "has_prev", "q", "dq", "ddq", "jerk", "v", "p_lie")}
            prev = loop.prev.copy()
            shadow.update_reference([p.copy() for p in g["via_p_via"]], [q.copy() for q in g["via_r_via"]],
                                    [b.copy() for b in g["via_bp1"]], [b.copy() for b in g["via_br1"]],
                                    [e.copy() for e in g["via_e_r_bound"]], [a.copy() for a in g["via_a_sets"]],
                                    [b.copy() for b in g["via_b_sets"]], [])
            for r in range(R):
                loop.set_rollout(r, shadow.mpc, keep["q"][r], keep["dq"][r], keep["ddq"][r], keep["jerk"][r], keep["q"][r],
                                 keep["v"][r], keep["p_lie"][r])
                W = state_view(loop.lay, loop.state[r])
                for f in ("slacks0", "error_count", "has_prev"):
                    W[f][:] = keep[f][r]
            loop.prev[:] = prev
            loop.upload()
        loop.prepare()
        x0, lbx, ubx, p = loop.problem()
        for r in range(R):
            for name, mine in (("x0", x0), ("lbx", lbx), ("ubx", ubx), ("p", p)):
                d = np.abs(mine[r] - big(g["call_" + name][k])).max()
                assert d < (2e-6 if (has_obs and name == "p") else 1e-9), (k, r, name, d)
        loop.set_solution(np.tile(g["call_x"][k], (R, 1)), np.full(R, int(g["iters"][k])), np.full(R, int(g["status"][k])),
                          np.full(R, float(g["viol"][k])))
        log = loop.finish()
        shadow.step()
        recs = loop.records()                   # MPCData records written by bmpc_loop_k_finish for rollouts 0 and 2
        assert recs.shape[:2] == (1, 2)
        for j in range(2):
            compare_record_with_host_mirror(recs[0, j], N, shadow, k)
        V = loop.download()
        for r in range(R):
            assert [int(s) for s in V["split"][r]] == list(g["split_idxs"][k]), (k, r)
            assert int(V["rp_sector"][r][0]) == int(g["sector"][k])
            for f, key in (("pr_ref", "pr_ref"), ("phi_current", "phi_current"), ("phi_max", "phi_max"), ("slacks0", "slacks0"),
                           ("rp_pd", "rp_pd"), ("q", "out_q"), ("dq", "out_dq"), ("ddq", "out_ddq"), ("v", "out_v"),
                           ("qf", "out_qf"), ("p_lie", "out_p_lie")):
                assert np.abs(V[f][r] - np.asarray(g[key][k]).reshape(-1)).max() < 1e-9, (k, r, f)
            assert abs(log[r, loop.LOG["phi"]] - g["phi_current"][k][0]) < 1e-9
    assert ((V["phi_current"][:, 0] >= V["phi_max"][:, 0] - 0.001) == (g["phi_current"][-1][0] >= g["phi_max"][-1][0] - 0.001)).all()
    assert (V["error_count"][:, 0] == int(g["error_count"][-1])).all()


def _scenario(be, R, N, seed):
    """configs[4] style: random start/goal, fixed EE workspace box, obstacle-free collision sets."""
    from boundplanner_amd import scenes
    from boundplanner_amd.params import normalize_set_size
    rng = np.random.default_rng(seed)
    q_start, q_goal = scenes.sample_start_goal(rng, be.fk, R)
    fs, fg = be.fk(q_start), be.fk(q_goal)
    a_ee, b_ee = scenes._box_set([-1.0, -1.0, 0.0], [1.0, 1.0, 1.2])
    plans = []
    for r in range(R):
        sets = normalize_set_size([[a_ee, b_ee]], 15)
        plans.append(dict(goal=fg["ee_pos"][r].copy(), r_via=[fs["ee_rot"][r].copy(), fg["ee_rot"][r].copy()],
                          bp1=[np.array([0.0, 0, 1])], br1=[np.array([0.0, 0, 1])],
                          erb=[np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180], a=[sets[0][0]], b=[sets[0][1]]))
    return q_start, plans


def test_device_loop_tracks_host_loop():
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.solver import HipBoundMPC
    N, R, steps = 10, 12, 14
    params = _params(N)
    be = HipBoundMPC(N, max_batch=R)
    q_start, plans = _scenario(be, R, N, 4096)
    cp = lambda P: dict(goal=P["goal"].copy(), r_via=[m.copy() for m in P["r_via"]], bp1=[b.copy() for b in P["bp1"]],
                        br1=[b.copy() for b in P["br1"]], erb=[e.copy() for e in P["erb"]], a=[a.copy() for a in P["a"]],
                        b=[b.copy() for b in P["b"]])
    # host loop
    host = BatchMPCNode(be, q_start, params)
    host.step()                                          # start-up solve on the trivial path
    for r in range(R):
        P = cp(plans[r])
        host.update_reference(r, [host.p_lie[r][:3].copy(), P["goal"]], P["r_via"], P["bp1"], P["br1"], P["erb"], P["a"], P["b"])
    # device loop: same start-up step, then the same plans
    ref = BatchMPCNode(be, q_start, params)               # only its freshly constructed host objects are used
    loop = DeviceLoop(be, R)
    for r in range(R):
        loop.set_rollout(r, ref.mpcs[r], ref.q[r], ref.dq[r], ref.ddq[r], ref.jerk[r], ref.qf[r], ref.v[r], ref.p_lie[r])
    loop.upload()
    loop.run(1)
    V = loop.download()
    assert np.abs(V["q"] - host.q).max() < 1e-9 and np.abs(V["p_lie"] - host.p_lie).max() < 1e-9
    for r in range(R):
        P = cp(plans[r])
        loop.replan(r, ref.mpcs[r], [V["p_lie"][r][:3].copy(), P["goal"]], P["r_via"], P["bp1"], P["br1"], P["erb"], P["a"], P["b"])
    loop.upload()
    dmax = 0.0
    for k in range(steps):
        host.step()
        log = loop.run(1)[0]
        phi_host = np.array([m.phi_current[0] for m in host.mpcs])
        dmax = max(dmax, np.abs(log[:, loop.LOG["q"]] - host.q).max(), np.abs(log[:, loop.LOG["p_lie"]] - host.p_lie).max(),
                   np.abs(log[:, loop.LOG["phi"]] - phi_host).max())
        assert (log[:, loop.LOG["iters"]] == host.iters[-1]).mean() > 0.9, k
        assert (log[:, loop.LOG["split1"]] == np.array([m.split_idxs[1] for m in host.mpcs])).all(), k
    # same solver, same arguments to rounding: the loops stay together far below the solver tolerance
    assert dmax < 1e-6, dmax
    assert (phi_host > 0.05).all()                       # the rollouts actually move along their paths


def test_device_loop_with_obstacles_tracks_host_loop():
    """Scene obstacles: the per-step collision sets come from the colpairs kernel + the greedy selection in the prepare
    kernel on the device, from collision_sets.find_set_collision_avoidance on the host."""
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.solver import HipBoundMPC
    from test_device_loop import _box_scene
    N, R, steps = 10, 4, 8
    params = _params(N)
    be = HipBoundMPC(N, max_batch=R)
    q_start, plans = _scenario(be, R, N, 77)
    # boxes near, but not touching, the arms: a collision point inside an obstacle has no separating halfspace (0/0 in
    # the reference's finder as well)
    col0 = be.fk(q_start)["col_pts"].reshape(-1, 3)
    rng, sets, pts = np.random.default_rng(3), [], []
    while len(sets) < 6:
        s1, p1 = _box_scene(rng, 1)
        lo, hi = p1[0].min(axis=0), p1[0].max(axis=0)
        gap = np.linalg.norm(np.maximum(np.maximum(lo - col0, col0 - hi), 0.0), axis=1).min()
        if 0.15 < gap < 0.45:
            sets += s1; pts += p1
    cp = lambda P: dict(goal=P["goal"].copy(), r_via=[m.copy() for m in P["r_via"]], bp1=[b.copy() for b in P["bp1"]],
                        br1=[b.copy() for b in P["br1"]], erb=[e.copy() for e in P["erb"]], a=[a.copy() for a in P["a"]],
                        b=[b.copy() for b in P["b"]])
    host = BatchMPCNode(be, q_start, params)
    ref = BatchMPCNode(be, q_start, params)
    for m in host.mpcs + ref.mpcs:
        m.set_obstacle_sets(sets, pts)
    loop = DeviceLoop(be, R)
    loop.set_obstacles(sets, pts)
    for r in range(R):
        P = cp(plans[r])
        host.update_reference(r, [host.p_lie[r][:3].copy(), P["goal"]], P["r_via"], P["bp1"], P["br1"], P["erb"], P["a"], P["b"])
        P = cp(plans[r])
        ref.update_reference(r, [ref.p_lie[r][:3].copy(), P["goal"]], P["r_via"], P["bp1"], P["br1"], P["erb"], P["a"], P["b"])
        loop.set_rollout(r, ref.mpcs[r], ref.q[r], ref.dq[r], ref.ddq[r], ref.jerk[r], ref.qf[r], ref.v[r], ref.p_lie[r])
    loop.upload()
    dmax, rows = 0.0, 0
    for k in range(steps):
        loop.prepare()
        p_dev = loop.problem()[3]
        assert np.isfinite(p_dev).all(), k
        host.step()
        loop.solve()
        log = loop.finish()
        rows = max(rows, int((np.abs(p_dev[:, 515:785].reshape(R, 6, 3, 15)).sum(axis=2) > 0).sum(axis=2).max()))
        assert np.isfinite(log).all() and np.isfinite(host.q).all(), k
        dmax = max(dmax, np.abs(log[:, loop.LOG["q"]] - host.q).max(), np.abs(log[:, loop.LOG["p_lie"]] - host.p_lie).max())
    assert rows > 6                    # obstacle halfspaces were active in the collision sets
    assert np.isfinite(dmax) and dmax < 1e-5, dmax           # the halfspaces agree to ~1e-7 (golden section), the closed loops stay together
    assert (log[:, loop.LOG["dead"]] == 0).all()


@pytest.mark.parametrize("fixture", ["closed_loop.npz", "closed_loop_scene.npz"])
def test_device_loop_tracks_reference_trace_with_hip_solver(golden_dir, fixture):
    """The reference's closed-loop scenario end to end on the device (prepare kernel -> HIP solve -> finish kernel):
    same switching step, same number of steps to the path end, states within the stated solver tolerance of the
    golden trace (which was produced by the reference's host code with the CPU oracle in the solver slot).
    closed_loop_scene.npz = BASELINE configs[0]: the reference's example scene (start, goal, workspace, 12 box obstacles,
    N=15), per-step collision sets on the device against the reference's own ConvexSetFinder in the trace."""
    import oracle_lib as O
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.mpc_node import MPCNode
    from boundplanner_amd.robot_model import RobotModel
    from boundplanner_amd.solver import HipBoundMPC
    g = np.load(os.path.join(golden_dir, fixture))
    N = int(g["N"])
    params = _params(N)
    be = HipBoundMPC(N)
    loop = DeviceLoop(be, 1)
    if "boxes" in g.files:
        from boundplanner_amd import scenes
        loop.set_obstacles(*scenes.boxes_to_sets(g["boxes"]))
    seed = MPCNode(g["in_q"][0], RobotModel(be.fk), lambda n, dt: None, params=params)   # host construction only
    loop.set_rollout(0, seed.mpc, seed.q, seed.dq, seed.ddq, seed.jerk, seed.qf, seed.v, seed.p_lie)
    loop.upload()
    n_steps, n_update = g["in_q"].shape[0], int(g["n_update"])
    L, dq_max, dp_max, iters = loop.LOG, 0.0, 0.0, []
    for k in range(n_steps):
        if k == n_update:
            V = loop.download()
            loop.replan(0, seed.mpc, [p.copy() for p in g["via_p_via"]], [r.copy() for r in g["via_r_via"]],
                        [b.copy() for b in g["via_bp1"]], [b.copy() for b in g["via_br1"]], [e.copy() for e in g["via_e_r_bound"]],
                        [a.copy() for a in g["via_a_sets"]], [b.copy() for b in g["via_b_sets"]])
            loop.upload()
        row = loop.run(1)[0, 0]
        iters.append(row[L["iters"]])
        assert int(row[L["split1"]]) == int(g["split_idxs"][k][1]) and int(row[L["sector"]]) == int(g["sector"][k]), k
        assert row[L["error_count"]] == 0
        dq_max = max(dq_max, np.abs(row[L["q"]] - g["out_q"][k]).max())
        dp_max = max(dp_max, np.abs(row[L["p_lie"]] - g["out_p_lie"][k]).max())
    assert dp_max < 1e-3 and dq_max < 5e-3, (dp_max, dq_max)
    assert row[L["phi"]] >= row[L["phi_max"]] - 0.001
    assert abs(np.mean(iters) - g["iters"].mean()) < 1.0


def test_device_collision_sets_match_the_reference_finder(golden_dir):
    """a10 / f2 on the MI355X: bmpc_loop_k_colpairs + the greedy selection of bmpc_loop_k_prepare against the REFERENCE's own
    finder on its example scene (tests/golden/colsets.npz: 56 (q0, qf) pairs x 6 collision points x 12 boxes)."""
    from boundplanner_amd import scenes
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.robot_model import RobotModel
    from boundplanner_amd.solver import HipBoundMPC
    from test_collision_sets import expected_set_params
    from test_device_loop import _rollout_state
    g = np.load(os.path.join(golden_dir, "colsets.npz"))
    N, R = 8, g["q0"].shape[0]
    params = _params(N)
    be = HipBoundMPC(N, max_batch=R)
    robot = RobotModel(be.fk)
    loop = DeviceLoop(be, R)
    loop.set_obstacles(*scenes.boxes_to_sets(g["boxes"]))
    for r in range(R):
        mpc, _, p_lie = _rollout_state(loop.lay, params, robot, g["q0"][r], g["qf"][r])
        loop.set_rollout(r, mpc, g["q0"][r], np.zeros(7), np.zeros(7), np.zeros(7), g["qf"][r], np.zeros(6), p_lie)
    loop.upload()
    loop.prepare()
    p = loop.problem()[3]
    worst = 0.0
    for r in range(R):
        a_ref, b_ref, b_ok = expected_set_params(g, r)
        worst = max(worst, np.abs(p[r, 515:785] - a_ref).max(), np.abs(p[r, 785:875] - b_ref)[b_ok].max())
    assert worst < 1e-6, worst
    assert (loop.download()["dead"] == 0).all()
    # a scene that needs 20 halfspaces around one collision point: the rollout is frozen (the reference's normalizer prints
    # an error and leaves the set ragged, util_functions.py:126-134)
    loop2 = DeviceLoop(be, 1)
    loop2.set_obstacles(*scenes.boxes_to_sets(g["shell_boxes"]))
    mpc, _, p_lie = _rollout_state(loop2.lay, params, robot, g["shell_q"][0], g["shell_q"][0])
    loop2.set_rollout(0, mpc, g["shell_q"][0], np.zeros(7), np.zeros(7), np.zeros(7), g["shell_q"][0], np.zeros(6), p_lie)
    loop2.upload()
    loop2.prepare()
    assert loop2.download()["dead"][0] == 2.0


def _config4_loop(be, q_start, fs, fg, params, rows):
    """configs[4] rollouts `rows` on the device: start-up solve on the trivial path, then the 2-via-point plan to the goal
    pose with the fixed workspace box as EE set."""
    from boundplanner_amd import scenes
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.params import normalize_set_size
    seed_objs = BatchMPCNode(be, q_start[rows], params)
    loop = DeviceLoop(be, len(rows))
    for i in range(len(rows)):
        loop.set_rollout(i, seed_objs.mpcs[i], seed_objs.q[i], seed_objs.dq[i], seed_objs.ddq[i], seed_objs.jerk[i], seed_objs.qf[i],
                         seed_objs.v[i], seed_objs.p_lie[i])
    loop.upload()
    loop.run(1, log=False)
    V = loop.download()
    a_ee, b_ee = scenes._box_set([-1.0, -1.0, 0.0], [1.0, 1.0, 1.2])
    for i, r in enumerate(rows):
        sets = normalize_set_size([[a_ee, b_ee]], 15)
        loop.replan(i, seed_objs.mpcs[i], [V["p_lie"][i][:3].copy(), fg["ee_pos"][r].copy()], [fs["ee_rot"][r].copy(), fg["ee_rot"][r].copy()],
                    [np.array([0.0, 0, 1])], [np.array([0.0, 0, 1])], [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180],
                    [sets[0][0]], [sets[0][1]])
    loop.upload()
    return loop


def test_config4_closed_loop_full_size():
    """BASELINE configs[4] at its full size (4096 rollouts x 200 steps, N=30, warm start as the reference, SURVEY 8(d) generator,
    seed 4096) through the device-resident loop: (1) every accepted step passes the reference's acceptance test; (2) a
    rollout's trajectory does not depend on the batch it is stepped in (the first 512 rollouts alone, bitwise);
    (3) who reaches the path end (all 4096 rollouts after the 200 steps).  The SURVEY generator draws goals from U(0.5 q_lim) and fixes the
    end-effector set to the workspace box [-1,-1,0]-[1,1,1.2]; the iiwa reaches z = 1.43 there, so 49 % of the goal poses
    lie OUTSIDE the set the end effector must stay in.  Those rollouts stop at the box (their last solve is a KKT point with
    the set rows active: tools/closed_loop_device.py --diagnose, profiles/r02_closed_loop_diag.json) -- a property of the
    synthetic scene, not a solver stall: rollouts whose goal is inside the box reach it."""
    from boundplanner_amd import scenes
    from boundplanner_amd.params import Q_LIM_LOWER, Q_LIM_UPPER
    from boundplanner_amd.solver import HipBoundMPC
    N, R, steps = 30, 4096, 200
    params = _params(N)
    be = HipBoundMPC(N, max_batch=R)
    rng = np.random.default_rng(4096)
    q_start, q_goal = scenes.sample_start_goal(rng, be.fk, R)
    fs, fg = be.fk(q_start), be.fk(q_goal)
    loop = _config4_loop(be, q_start, fs, fg, params, np.arange(R))
    log = loop.run(steps)                                   # [steps][R][logw]
    L = loop.LOG
    it, st, viol, err, dead = (log[:, :, L[k]] for k in ("iters", "status", "viol", "error_count", "dead"))
    accepted = (err == 0) & (dead == 0)
    assert ((st[accepted] == 0) | (viol[accepted] < 1e-4)).all()
    assert (err > 0).mean() < 1e-3 and dead[-1].sum() <= 8                # (observed: 8 failed solves of 819 200, nobody frozen)
    assert it[0].mean() > it[10:].mean()                     # the warm start pays: later steps are cheaper than the first
    print(f"configs[4], 4096 rollouts x {steps} steps: mean iterations {it.mean():.1f} (first step {it[0].mean():.1f}), failed steps "
          f"{(err > 0).mean():.4f}, frozen rollouts {int(dead[-1].sum())}")
    # (2) sub-run
    sub = _config4_loop(be, q_start, fs, fg, params, np.arange(512))
    log2 = sub.run(12)
    for k in ("q", "p_lie", "phi", "iters", "status"):
        assert np.array_equal(log2[:, :, L[k]], log[:12, :512, L[k]]), k
    # (3) path end, all rollouts after the config's 200 steps
    rows = np.arange(R)
    lg = log[-1]
    reached = lg[:, L["phi"]] >= lg[:, L["phi_max"]] - 0.001
    pg = fg["ee_pos"][rows]
    inside = (np.abs(pg[:, :2]) < 0.98).all(axis=1) & (pg[:, 2] < 1.18) & (pg[:, 2] > 0.02)
    far_out = pg[:, 2] > 1.25
    q_end = lg[:, L["q"]]
    at_limit = (np.minimum(q_end - Q_LIM_LOWER, Q_LIM_UPPER - q_end) < 2e-3).any(axis=1)
    ok = inside & ~at_limit & (lg[:, L["dead"]] == 0)
    print(f"  after {steps} steps: at the path end {reached.mean():.3f}; goals inside the workspace box {inside.mean():.3f} -> at the end "
          f"{reached[ok].mean():.3f} (not at a joint limit); goals above z = 1.25 ({far_out.mean():.3f}) -> at the end {reached[far_out].mean():.3f}")
    # observed: 1.000 of the goals inside the box, 0.16 of the goals above z = 1.25 (the set rows are soft: pslack)
    assert reached[ok].mean() > 0.99 and reached[far_out].mean() < 0.3


def test_async_closed_loop_equals_lock_step():
    """bmpc_loop_run_async: rollouts that do not wait for each other (a rollout's slot of the solver pool is re-admitted
    with its next problem as soon as its own solve has retired) produce, rollout by rollout and step by step, bitwise the
    log of the lock-step loop -- with and without scene obstacles (collision sets on the device)."""
    from boundplanner_amd import scenes
    from boundplanner_amd.solver import HipBoundMPC
    N, R, steps = 10, 96, 14
    params = _params(N)
    be = HipBoundMPC(N, max_batch=R)
    rng = np.random.default_rng(7)
    q_start, q_goal = scenes.sample_start_goal(rng, be.fk, R)
    fs, fg = be.fk(q_start), be.fk(q_goal)
    boxes, _, _, _ = scenes.example_scene()
    for obstacles in (None, scenes.boxes_to_sets(boxes[4:8])):
        logs = []
        # run_async: one lane (default), and two lanes (BMPC_FAST_LANE: the rollouts that lag behind iterate in a fast lane of their
        # own, bmpc_capi.hip) -- default knobs, a small fast lane with short bursts, the lanes on disjoint sets of CUs
        variants = [("run", {}), ("run_async", {"BMPC_FAST_LANE": "64"}), ("run_async", {}), ("run_async", {"BMPC_FAST_LANE": "7", "BMPC_FAST_BURST": "1"}),
                    ("run_async", {"BMPC_FAST_LANE": "24", "BMPC_FAST_CUS": "32"})]
        lanes = []
        for mode, env in variants:
            loop = _config4_loop(be, q_start, fs, fg, params, np.arange(R))
            if obstacles is not None:
                loop.set_obstacles(*obstacles)
            old_env = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                logs.append(getattr(loop, mode)(steps))
            finally:
                for k, v in old_env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            lanes.append(be.lane_stats() if mode == "run_async" else None)
        assert np.isfinite(logs[0]).all()
        L = loop.LOG
        assert len(set(logs[0][:, :, L["iters"]].ravel().tolist())) > 3          # the solves do take different numbers of iterations
        for lg in logs[1:]:
            assert np.array_equal(logs[0], lg)
        assert lanes[1]["fast_super_steps"] > 0 and lanes[1]["fast_lane_instances_mean"] > 0      # the fast lane did run, with instances in it
        assert lanes[2]["bursts"] == 0                                                            # ... and not by default


def test_config0_plan_then_track_with_replanning_on_the_device_loop(golden_dir):
    """BASELINE configs[0] as the reference runs it (boundplanner_with_mpc_example.py:102-157): BoundPlanner.plan_convex_set_path
    on the 12-box scene -> update_reference -> track, on the device-resident loop with per-step collision sets; halfway the
    rollout is replanned through the planner (replanning=True, p_horizon = the MPC horizon of its last solution, BoundPlanner.py:
    231-276, 706-729) and must still arrive.  The first plan equals the fixture produced by the reference's planner logic."""
    from scipy.spatial.transform import Rotation as Rot
    from boundplanner_amd import scenes
    from boundplanner_amd.batch_node import BatchMPCNode
    from boundplanner_amd.bound_planner import BoundPlanner
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.solver import HipBoundMPC
    boxes, q0, goal_p, goal_r = scenes.example_scene()
    base = get_default_params()
    N = 15
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    be = HipBoundMPC(N)
    seed = BatchMPCNode(be, q0[None], params)
    loop = DeviceLoop(be, 1)
    loop.set_obstacles(*scenes.boxes_to_sets(boxes))
    loop.set_rollout(0, seed.mpcs[0], seed.q[0], seed.dq[0], seed.ddq[0], seed.jerk[0], seed.qf[0], seed.v[0], seed.p_lie[0])
    loop.upload()
    loop.run(1, log=False)
    loop.download()
    planner = BoundPlanner(e_p_max=0.5, obstacles=boxes, workspace_max=[1.0, 0.38, 1.0], workspace_min=[-0.14, -1.0, 0.0], seed=7)
    p_via, r_via, bp1, sets = loop.plan_and_replan(0, seed.mpcs[0], planner, goal_p, goal_r, replanning=False)
    fx = np.load(os.path.join(golden_dir, "plan.npz"))
    assert np.abs(np.array(p_via) - fx["example_p_via"]).max() < 1e-5      # (the start pose is the GPU's FK of q0: 1e-12 from the fixture's)
    loop.upload()
    L = loop.LOG
    loop.set_record([0])
    log = loop.run(10)
    from boundplanner_amd import mpc_data
    recs = loop.records()
    assert recs.shape[:2] == (10, 1)
    # a caller's buffer that is too small for the recorded steps is refused, not overrun (bmpc_loop_records takes its capacity)
    import ctypes
    small = np.zeros((3, 1, recs.shape[2])); nst = ctypes.c_int(0)
    assert loop.lib.bmpc_loop_records(loop._l, small.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 3, ctypes.byref(nst)) == 1
    assert nst.value == 10 and not small.any()
    for k in range(10):                          # the trace of the run: one MPCData record per step, consistent with the log
        md = mpc_data.from_device_record(recs[k, 0], N)
        assert md["iterations"] == int(log[k, 0, L["iters"]]) and len(md["p"]) == N - 1 and len(md["dddq"]) == N
        assert set(mpc_data.FIELDS) <= set(md) and np.isfinite(np.concatenate([np.ravel(v) for v in md["e_r"]])).all()
    loop.set_record([])
    assert (log[:, 0, L["error_count"]] == 0).all()
    phi_mid, phi_max0 = log[-1, 0, L["phi"]], log[-1, 0, L["phi_max"]]
    assert 0.05 < phi_mid < phi_max0 - 0.01                                 # under way, not there yet
    loop.download()
    p2, _, _, sets2 = loop.plan_and_replan(0, seed.mpcs[0], planner, goal_p, goal_r, replanning=True)
    assert planner.replanning and planner.replanning_phi >= 0.0 and np.abs(p2[-1] - goal_p).max() < 1e-9
    loop.upload()
    arrived, steps, fails = False, 0, 0
    while steps < 300 and not arrived:
        log = loop.run(10)
        steps += 10
        fails += int((log[:, 0, L["error_count"]] > 0).sum())
        arrived = bool((log[:, 0, L["phi"]] >= log[:, 0, L["phi_max"]] - 0.001).any())
    assert arrived and fails <= 2
    p_end = log[-1, 0, L["p_lie"]][:3]
    assert np.linalg.norm(p_end - goal_p) < 0.02
    # the tracked path stays clear of the obstacles themselves
    for bx in boxes:
        assert not ((p_end >= bx[:3]) & (p_end <= bx[3:])).all()
