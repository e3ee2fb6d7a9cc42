"""Host-side RobotModel with the reference's method names (RobotModel/RobotModel.py:15-267).

The arithmetic is done by a batched kinematics backend `fk_fn(q[B,7], dq[B,7]) -> dict` with
keys ee_pos [B,3], ee_rot [B,3,3], col_pts [B,6,3], jac [B,6,7], dvdq [B,6,7].  The product
backend is the HIP library (`boundplanner_amd.solver.HipBoundMPC.fk`); there is NO CPU fallback
in this package -- tests inject the oracle's FK explicitly.
"""
import numpy as np
from scipy.spatial.transform import Rotation as R

from .params import COL_JOINT_SIZES, DQ_LIM, Q_LIM_LOWER, Q_LIM_UPPER, U_MAX


class RobotModel:
    def __init__(self, fk_fn=None, robot=None):
        """robot: table of boundplanner_amd.robots (None = iiwa14, RobotModel.py:10 USE_IIWA = True); `fk_fn` must have been
        given the same table (HipBoundMPC(N, robot=...).fk)."""
        if fk_fn is None:
            from .solver import default_fk_fn  # raises loudly when the HIP library is missing
            fk_fn = default_fk_fn()
        self._fk = fk_fn
        self.robot = robot
        if robot is None:
            self.col_joint_sizes = list(COL_JOINT_SIZES)
            self.q_lim_lower, self.q_lim_upper = Q_LIM_LOWER.copy(), Q_LIM_UPPER.copy()
            self.dq_lim_lower, self.dq_lim_upper = -DQ_LIM.copy(), DQ_LIM.copy()
        else:
            self.col_joint_sizes = list(robot["col_joint_sizes"])
            ql, qh = np.asarray(robot["q_lower"], float), np.asarray(robot["q_upper"], float)
            self.q_lim_lower, self.q_lim_upper = np.where(ql <= -1e19, -np.inf, ql), np.where(qh >= 1e19, np.inf, qh)
            self.dq_lim_lower, self.dq_lim_upper = -np.asarray(robot["dq_max"], float), np.asarray(robot["dq_max"], float)
        self.tau_lim_lower = [-320, -320, -176, -176, -110, -40, -40]
        self.tau_lim_upper = [320, 320, 176, 176, 110, 40, 40]
        self.u_max, self.u_min = U_MAX, -U_MAX

    def get_robot_limits(self):
        return (self.q_lim_upper, self.q_lim_lower, self.dq_lim_upper, self.dq_lim_lower,
                self.tau_lim_upper, self.tau_lim_lower, self.u_max, self.u_min)

    def _one(self, q, dq=None):
        q = np.asarray(q, float).reshape(1, 7)
        dq = np.zeros((1, 7)) if dq is None else np.asarray(dq, float).reshape(1, 7)
        return {k: v[0] for k, v in self._fk(q, dq).items()}

    def batch(self, q, dq=None):
        q = np.ascontiguousarray(q, float)
        dq = np.zeros_like(q) if dq is None else np.ascontiguousarray(dq, float)
        return self._fk(q, dq)

    def fk_pos(self, q):
        return self._one(q)["ee_pos"]

    def fk_pos_col(self, q, i):
        return self._one(q)["col_pts"][i]

    def hom_transform_endeffector(self, q):
        o = self._one(q)
        h = np.eye(4)
        h[:3, :3], h[:3, 3] = o["ee_rot"], o["ee_pos"]
        return h

    def fk(self, q):
        o = self._one(q)
        return np.concatenate((o["ee_pos"], R.from_matrix(o["ee_rot"]).as_rotvec()))

    def jacobian_fk(self, q):
        return self._one(q)["jac"]

    def velocity_ee(self, q, dq):
        return (self.jacobian_fk(q) @ dq)[:3]

    def omega_ee(self, q, dq):
        return (self.jacobian_fk(q) @ dq)[3:]

    def forward_kinematics(self, q, dq):
        """(p_lie, jac, djac) as RobotModel.py:70-77; djac is not used on the MPC path and is
        returned as zeros."""
        o = self._one(q, dq)
        p = np.concatenate((o["ee_pos"], R.from_matrix(o["ee_rot"]).as_rotvec()))
        return p, o["jac"], np.zeros((6, 7))
