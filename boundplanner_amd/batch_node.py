"""Batched closed loop: B independent MPCNode-style rollouts advanced in lock step, ONE batched
solve per MPC step (the build's batched counterpart of the reference's single-instance track
driver, /root/reference/bound_planner/BoundMPC/MPCNode.py:106-160; BASELINE.json configs[4]).

Per step: every rollout prepares its NLP on the host (BoundMPC.prepare, the reference's sequential
prep logic), the B problems go to the GPU in one call (HipBoundMPC.solve_batch), every rollout
post-processes its solution (BoundMPC.finish) and integrates its joint state.  The kinematics of all
rollouts are evaluated in one batched FK call per step.
"""
import time

import numpy as np
from scipy.spatial.transform import Rotation as R

from .bound_mpc import BoundMPC
from .mpc_node import integrate_joint
from .params import get_default_params
from .robot_model import RobotModel


class _CachedRobot(RobotModel):
    """RobotModel whose single-configuration queries are served from one batched FK evaluation."""

    def __init__(self, fk_fn):
        super().__init__(fk_fn)
        self._cache = {}

    def prime(self, qs, dqs=None):
        out = self._fk(np.ascontiguousarray(qs, float), None if dqs is None else np.ascontiguousarray(dqs, float))
        for i, q in enumerate(np.asarray(qs, float)):
            self._cache[q.tobytes()] = {k: v[i] for k, v in out.items()}

    def _one(self, q, dq=None):
        key = np.asarray(q, float).tobytes()
        hit = self._cache.get(key)
        if hit is not None:          # ee_pos/ee_rot/col_pts/jac do not depend on dq
            return hit
        return super()._one(q, dq)


class BatchMPCNode:
    def __init__(self, backend, q0s, params=None):
        """backend: HipBoundMPC (N must equal params.n); q0s: [B, 7] start configurations."""
        self.be = backend
        self.params = params or get_default_params()
        assert backend.N == self.params.n
        self.robot = _CachedRobot(backend.fk)
        q0s = np.asarray(q0s, float)
        self.B = q0s.shape[0]
        self.robot.prime(q0s)
        self.q = q0s.copy()
        self.dq = np.zeros_like(q0s); self.ddq = np.zeros_like(q0s); self.jerk = np.zeros_like(q0s)
        self.qf = q0s.copy()
        self.v = np.zeros((self.B, 6))
        self.p_lie = np.array([self.robot.fk(q) for q in q0s])
        self.mpcs = []
        for b in range(self.B):
            p0 = self.p_lie[b]
            self.mpcs.append(BoundMPC([p0[:3]] * 2, [R.from_rotvec(p0[3:]).as_matrix()] * 2, [np.array([1.0, 0, 0])],
                                      [np.array([1.0, 0, 0])], [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180],
                                      [np.zeros((15, 3))], [np.ones(15)], [], p0=p0, params=self.params,
                                      robot_model=self.robot))
        self.iters, self.t_solve, self.t_host, self.fails = [], [], [], []

    def update_reference(self, b, p_via, r_via, bp1, br1, e_r_bound, a_sets, b_sets, obstacles=()):
        self.qf[b] = self.q[b]
        self.mpcs[b].update(p_via, r_via, bp1, br1, e_r_bound, a_sets, b_sets, list(obstacles), self.v[b],
                            p0=np.copy(self.p_lie[b]), params=self.params)

    def done(self):
        return np.array([m.phi_current[0] >= m.phi_max[0] - 0.001 for m in self.mpcs])

    def step(self):
        t0 = time.perf_counter()
        B = self.B
        self.robot._cache.clear()
        self.robot.prime(np.vstack((self.q, self.qf)))
        self.p_lie = np.array([self.robot.fk(self.q[b]) for b in range(B)])
        prep = [self.mpcs[b].prepare(self.q[b], self.dq[b], self.ddq[b], self.p_lie[b], self.v[b], self.jerk[b], self.qf[b])
                for b in range(B)]
        big = lambda a: np.nan_to_num(a, posinf=1e20, neginf=-1e20)
        x0 = np.array([p[0] for p in prep]); lbx = big(np.array([p[1] for p in prep]))
        ubx = big(np.array([p[2] for p in prep])); par = np.array([p[3] for p in prep])
        t1 = time.perf_counter()
        r = self.be.solve_batch(x0, lbx, ubx, par)
        t2 = time.perf_counter()
        trajs = []
        for b in range(B):
            traj, _, _ = self.mpcs[b].finish(self.q[b], self.dq[b], self.ddq[b], self.jerk[b], self.p_lie[b], r["x"][b],
                                             int(r["status"][b]) == 0, float(r["viol"][b]), prep[b][4])
            trajs.append(traj)
        qn = np.empty_like(self.q)
        for b in range(B):
            new = integrate_joint(self.robot, trajs[b]["dddq"], self.q[b], self.dq[b], self.ddq[b], self.params.dt)
            qn[b], self.dq[b], self.ddq[b], self.p_lie[b] = new[0], new[1], new[2], new[3]
            self.v[b] = new[4]
            self.qf[b] = trajs[b]["q"][:, -1]
            self.jerk[b] = trajs[b]["dddq"][:, 1]
        self.q = qn
        t3 = time.perf_counter()
        self.iters.append(r["iters"].copy()); self.t_solve.append(t2 - t1); self.t_host.append((t1 - t0) + (t3 - t2))
        self.fails.append(np.array([m.error_count > 0 for m in self.mpcs]))
        return trajs
