"""Closed-loop driver of one MPC instance with the reference's MPCNode interface
(/root/reference/bound_planner/BoundMPC/MPCNode.py:12-160) and the jerk-hat joint integrator it
advances the robot state with (utils/util_functions.py:55-65, BoundMPC/jerk_trajectory_casadi.py:78-175).

No wall-clock pacing (the reference sleeps to dt per step, MPCNode.py:160): batched / offline use.
"""
import numpy as np
from scipy.spatial.transform import Rotation as R

from .bound_mpc import BoundMPC
from .params import get_default_params


def _hat_terms(jerk, h, t):
    """Contributions of the piecewise-linear ('hat') jerk samples jerk[:, j] at times j*h to
    (angle, velocity, acceleration) at time t in (0, h]; closed forms of the first interval only,
    which is all the closed loop evaluates (t == h).  jerk: (7, n)."""
    n = jerk.shape[1]
    ang = np.zeros(jerk.shape[0]); vel = np.zeros_like(ang); acc = np.zeros_like(ang)
    for j in range(n):
        p = jerk[:, j]
        if j == 0:                                  # falling half-hat starting at 0
            if 0 < t <= h:
                acc += -p * t * (t - 2 * h) / h / 2
                vel += -p * t ** 2 * (t - 3 * h) / h / 6
                ang += -p * t ** 3 * (t - 4 * h) / h / 24
            else:
                acc += p * h / 2
                vel += p * h * (3 * t - h) / 6
                ang += p * (h * h / 6 - 2 / 3 * t * h + t ** 2) * h / 4
        else:                                       # rising flank of the hat centred at j*h starts at (j-1)*h
            c1 = (j - 1) * h
            last = (j == n - 1)
            if c1 < t <= c1 + h:
                acc += p * (t - c1) ** 2 / h / 2
                vel += -p * (c1 - t) ** 3 / h / 6
                ang += p * (c1 - t) ** 4 / h / 24
            elif t > c1 + h:
                if last:
                    acc += p * h / 2
                    vel += p * h * (3 * t - 2 * h - 3 * c1) / 6
                    ang += p * h * (h * h / 2 + (-4 / 3 * t + 4 / 3 * c1) * h + (t - c1) ** 2) / 4
                elif t <= c1 + 2 * h:
                    acc += -(h * h + (-2 * t + 2 * c1) * h + (t - c1) ** 2 / 2) * p / h
                    vel += p * (h ** 3 + (-3 * t + 3 * c1) * h * h + 3 * (t - c1) ** 2 * h - (t - c1) ** 3 / 2) / h / 3
                    ang += -(h ** 4 + (-4 * t + 4 * c1) * h ** 3 + 6 * (t - c1) ** 2 * h * h - 4 * (t - c1) ** 3 * h
                             + (t - c1) ** 4 / 2) * p / h / 12
                else:
                    acc += p * h
                    vel += -h * p * (c1 + h - t)
                    ang += 7 / 12 * h * (h * h + (-12 / 7 * t + 12 / 7 * c1) * h + 6 / 7 * (t - c1) ** 2) * p
    return ang, vel, acc


def integrate_joint(model, jerk_matrix, q, dq, ddq, dt):
    """State after dt under the hat-function jerk trajectory; returns (q, dq, ddq, p_lie, v, a, j).
    Quirk kept from the reference (util_functions.py:61-62): the returned Cartesian velocity is
    J(q) dq at the OLD state."""
    t = dt
    a_t, v_t, acc_t = _hat_terms(np.asarray(jerk_matrix, float), dt, t)
    qn = ddq * t ** 2 / 2 + dq * t + q + a_t
    dqn = ddq * t + dq + v_t
    ddqn = ddq + acc_t
    pn_lie, jac, djac = model.forward_kinematics(qn, dqn)
    vn = np.concatenate((model.velocity_ee(q, dq), model.omega_ee(q, dq)))
    an = djac @ dqn + jac @ ddqn
    jn = 2 * djac @ ddqn + jac @ ddqn
    return qn, dqn, ddqn, pn_lie, vn, an, jn


class MPCNode:
    def __init__(self, q0, robot_model, solver_factory, params=None):
        """solver_factory(N, dt) -> object call-compatible with the CasADi function of BoundMPC.py:594-607
        (HipNlpSolver in the product)."""
        self.fails = []
        self.t_mpc = 0.0
        self.robot_model = robot_model
        self.q0 = np.asarray(q0, float)
        self.traj = self.ref_data = self.traj_data = None
        self.p0, _, _ = self.robot_model.forward_kinematics(self.q0, self.q0)
        self.params = params or get_default_params()
        self.dt = self.params.dt
        self._solver = solver_factory(self.params.n, self.dt)
        self.reset()

    def reset(self):
        """Trivial 2-point path at the current pose, sets (A=0, b=1) (MPCNode.py:44-80)."""
        self.p = self.p0
        p_via = [self.p0[:3]] * 2
        r_via = [R.from_rotvec(self.p0[3:]).as_matrix()] * 2
        self.mpc = BoundMPC(p_via, r_via, [np.array([1.0, 0.0, 0.0])], [np.array([1.0, 0.0, 0.0])],
                            [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180], [np.zeros((15, 3))],
                            [np.ones(15)], [], p0=self.p0, params=self.params, solver=self._solver,
                            robot_model=self.robot_model)
        self.q = self.q0
        self.qf = self.q0
        self.dq = np.zeros(7); self.ddq = np.zeros(7); self.jerk = np.zeros(7)
        self.p_lie = self.p0
        self.p_ref = self.p0
        self.v = np.zeros(6)
        self.t_current = 0.0
        self.k_current = 0
        self.iters = []

    def update_reference(self, p_via, r_via, bp1, br1, e_r_bound, a_sets, b_sets, obstacles):
        self.p0 = np.copy(self.p_lie)
        self.q0 = np.copy(self.q)
        self.qf = self.q0
        self.p = self.p0
        self.mpc.update(p_via, r_via, bp1, br1, e_r_bound, a_sets, b_sets, obstacles, self.v, p0=self.p0,
                        params=self.params)

    def step(self):
        self.p_lie, _, _ = self.robot_model.forward_kinematics(self.q, self.dq)
        traj_data, ref_data, err_data, self.t_mpc, iters = self.mpc.step(
            self.q, self.dq, self.ddq, self.p_lie, self.v, self.jerk, self.qf)
        self.p_ref = ref_data["p"][1]
        self.traj, self.traj_data, self.ref_data, self.err_data = traj_data["p"], traj_data, ref_data, err_data
        self.iters.append(iters)
        self.fails.append(1.0 if self.mpc.error_count > 0 else 0.0)
        self.t_current += self.mpc.dt
        self.k_current += 1
        jerk_traj = traj_data["dddq"]
        new = integrate_joint(self.robot_model, jerk_traj, self.q, self.dq, self.ddq, self.mpc.dt)
        self.q, self.dq, self.ddq, self.p_lie, self.v = new[0], new[1], new[2], new[3], new[4]
        self.qf = traj_data["q"][:, -1]
        self.p = self.p_lie
        self.jerk = jerk_traj[:, 1]
        return traj_data
