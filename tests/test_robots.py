"""Row f4 of SURVEY.md section 8: the reference's second arm (RobotModel.py:10-48, USE_IIWA = False: Kinova Gen3,
gen3_arm.urdf) and arbitrary URDFs with the same joint / frame names, as a robot TABLE (include/boundmpc.h bmpc_robot,
boundplanner_amd/robots.py) that the solver handle, the device loop, the oracle and the host classes take.
The tables IIWA14 / GEN3 equal the reference's URDF files number for number (checked in the build container by
tests/golden/gen/gen_robot_tables.py, which parses the files with robots.table_from_urdf)."""
import numpy as np
import pytest

import oracle_lib as O
from boundplanner_amd import robots, scenes

MINI_URDF = """<robot name="mini">
  <link name="base"/>
  %s
  <joint name="ee" type="fixed"><origin xyz="0 0 0.1" rpy="0 1.5 0"/><parent link="l7"/><child link="end_effector_link"/></joint>
  <joint name="link4_col" type="fixed"><origin xyz="0 0.05 0"/><parent link="l4"/><child link="link4_col_link"/></joint>
</robot>""" % "\n  ".join(
    f'<joint name="joint_{i}" type="revolute"><origin xyz="0 {0.01 * i} {0.1 + 0.02 * i}" rpy="{0.3 * i} 0 {1.0 - 0.2 * i}"/>'
    f'<parent link="{"base" if i == 1 else f"l{i - 1}"}"/><child link="l{i}"/><axis xyz="0 0 1"/>'
    f'<limit lower="{-10 if i % 2 else -2.0}" upper="{10 if i % 2 else 2.0}" velocity="{1.0 + 0.1 * i}" effort="1"/></joint>' for i in range(1, 8))


@pytest.fixture
def gen3_oracle():
    O.set_robot(robots.GEN3)
    yield robots.GEN3
    O.set_robot(None)


def test_table_from_urdf():
    t = robots.table_from_urdf(MINI_URDF, [0.05] * 7, name="mini")
    assert t["joint_xyz"][2] == [0.0, 0.03, 0.16] and abs(t["joint_rpy"][4][0] - 1.5) < 1e-15 and t["ee_rpy"] == [0.0, 1.5, 0.0]
    assert t["q_lower"][0] == -robots.BIG and t["q_upper"][1] == 2.0 and abs(t["dq_max"][6] - 1.7) < 1e-12      # +-10 rad = unlimited (RobotModel.py:46-48)
    assert t["link4_col_xyz"] == [0.0, 0.05, 0.0]
    r = robots.from_struct(robots.to_struct(t))
    for k in ("joint_xyz", "joint_rpy", "ee_xyz", "q_lower", "dq_max", "col_joint_sizes"):
        assert np.array_equal(np.asarray(r[k]), np.asarray(t[k], float))
    with pytest.raises(ValueError):
        robots.table_from_urdf(MINI_URDF.replace('<parent link="l4"/><child link="link4_col_link"/>',
                                                 '<parent link="l3"/><child link="link4_col_link"/>'), [0.05] * 7)


def test_gen3_oracle_kinematics_match_plain_urdf_fk(gen3_oracle):
    rng = np.random.default_rng(1)
    q = rng.uniform(-2.5, 2.5, (40, 7))
    f = O.fk_batch(q)
    for i in range(40):
        p, R, c = robots.chain_fk(gen3_oracle, q[i])
        assert np.abs(p - f["ee_pos"][i]).max() < 1e-13 and np.abs(R - f["ee_rot"][i]).max() < 1e-13
        assert np.abs(c - f["col_pts"][i]).max() < 1e-13
    # the Jacobian is the derivative of that position
    eps = 1e-6
    for j in range(7):
        dq = np.zeros(7); dq[j] = eps
        d = (robots.chain_fk(gen3_oracle, q[0] + dq)[0] - robots.chain_fk(gen3_oracle, q[0] - dq)[0]) / (2 * eps)
        assert np.abs(d - f["jac"][0][:3, j]).max() < 1e-8


def test_gen3_instances_solve_to_kkt_points(gen3_oracle):
    """The Gen3's limits reach the NLP as bounds (four unlimited joints -> no bound rows), its sphere radii as set offsets."""
    from test_oracle_solver import check_multipliers
    N = 10
    b = scenes.make_batch(4, N, 3, O.fk_batch, randomize_sets=True, robot=gen3_oracle)
    assert np.isinf(b["ubx"][0][1]) and b["ubx"][0][N + 1] == 2.24 and b["ubx"][0][7 * N + 1] == 1.3963     # q_1 free, q_2, dq_1 limited
    for i in range(4):
        r = O.solve(N, b["x0"][i], b["lbx"][i], b["ubx"][i], b["p"][i], tol=1e-8)
        assert r["status"] == 0
        lbx = np.where(np.isinf(b["lbx"][i]), -1e20, b["lbx"][i]); ubx = np.where(np.isinf(b["ubx"][i]), 1e20, b["ubx"][i])
        check_multipliers(N, r["x"], b["p"][i], lbx, ubx, r["lam_g"], r["lam_x"], 1e-7, 2e-7)


@pytest.mark.gpu
def test_gen3_on_the_gpu(gen3_oracle):
    """bmpc_set_robot: kinematics, batched solves and the device loop's bound rows for the Gen3 against the oracle / the table."""
    from boundplanner_amd.device_loop import DeviceLoop
    from boundplanner_amd.params import Params, get_default_params, make_bounds
    from boundplanner_amd.robot_model import RobotModel
    from boundplanner_amd.solver import HipBoundMPC
    from test_device_loop import _rollout_state
    N, B = 10, 24
    be = HipBoundMPC(N, robot="gen3")
    rng = np.random.default_rng(2)
    q = rng.uniform(-2.5, 2.5, (64, 7)); dq = rng.normal(size=(64, 7))
    a, o = be.fk(q, dq), O.fk_batch(q, dq)
    for k in a:
        assert np.abs(a[k] - o[k]).max() < 1e-12, k
    assert np.abs(a["ee_pos"][0] - robots.chain_fk(robots.GEN3, q[0])[0]).max() < 1e-13
    batch = scenes.make_batch(B, N, 11, be.fk, randomize_sets=True, robot=robots.GEN3)
    r = be.solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    ro = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    assert np.array_equal(r["status"], ro["status"]) and (r["status"] == 0).sum() >= B - 1
    same = (r["status"] == 0) & (r["iters"] == ro["iters"])
    assert same.sum() >= B - 3
    assert np.abs(r["x"][same][:, 28 * N:40 * N] - ro["x"][same][:, 28 * N:40 * N]).max() < 2e-5
    # an iiwa handle solves a different problem from the same numbers
    r_iiwa = HipBoundMPC(N).solve_batch(batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
    assert np.abs(r_iiwa["x"] - r["x"]).max() > 1e-3
    # device loop: constant bound rows and collision-set offsets come from the handle's robot
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    loop = DeviceLoop(be, 1)
    robot = RobotModel(be.fk, robot=robots.GEN3)
    mpc, _, p_lie = _rollout_state(loop.lay, params, robot, q[1] * 0.3, q[1] * 0.3)
    loop.set_rollout(0, mpc, q[1] * 0.3, np.zeros(7), np.zeros(7), np.zeros(7), q[1] * 0.3, np.zeros(6), p_lie)
    loop.upload(); loop.prepare()
    x0, lbx, ubx, p = loop.problem()
    w0, lbx_h, ubx_h, p_h, _ = mpc.prepare(q[1] * 0.3, np.zeros(7), np.zeros(7), p_lie, np.zeros(6), np.zeros(7), q[1] * 0.3)
    big = lambda v: np.nan_to_num(v, posinf=1e20, neginf=-1e20)
    assert np.abs(lbx[0] - big(lbx_h)).max() < 1e-12 and np.abs(ubx[0] - big(ubx_h)).max() < 1e-12
    assert np.abs(p[0] - p_h).max() < 1e-9
