"""Host mirror of the reference's BoundMPC class for ONE instance
(/root/reference/bound_planner/BoundMPC/BoundMPC.py:27-1040): same constructor/update/step
signatures and carried state, with the NLP solve delegated to a solver object that is
call-compatible with the CasADi function at BoundMPC.py:594-607 (HipNlpSolver in the product).

`prepare()` is the part of step() before the solver call (BoundMPC.py:388-589) and is also
what the batched scene generators use to build problem instances.
"""
import copy as cp
import time

import numpy as np
from scipy.spatial.transform import Rotation as R

from . import so3
from .collision_sets import find_set_collision_avoidance
from .params import (COL_JOINT_SIZES, NR_JOINTS, cold_start, make_bounds, n_w, normalize_set_size,
                     pack_params)
from .reference_path import ReferencePath


class BoundMPC:
    def __init__(self, pos_points, rot_points, bp1, br1, e_r_bound, a_sets, b_sets, obstacles,
                 p0=np.zeros(6), params=None, solver=None, robot_model=None, robot=None):
        self.robot = robot if robot is not None else getattr(robot_model, "robot", None)   # table of .robots, None = iiwa14
        self.N = params.n
        self.dt = params.dt
        self.nr_segs = params.nr_segs
        self.nr_joints = NR_JOINTS
        self.nr_slacks = 6 + 4 * self.N
        self.robot_model = robot_model
        self.solver = solver
        self.obstacles = obstacles
        self.obs_sets, self.obs_points_sets = [], []
        self.p0 = p0
        self.qd = np.zeros(7)
        self.error_count = 0
        self.slacks0 = np.zeros(6)
        self.ref_path = ReferencePath(pos_points, rot_points, bp1, br1, e_r_bound, a_sets,
                                      b_sets, self.nr_segs)
        self.split_idxs = [0] + [self.N] * self.nr_segs
        self.switch = False
        S = self.nr_segs
        self.dtau_init = np.empty((3, S))
        self.dtau_init_par = np.empty((3, S))
        self.dtau_init_orth1 = np.empty((3, S))
        self.dtau_init_orth2 = np.empty((3, S))
        self.phi_max = np.array([self.ref_path.phi_max])
        self.weights = np.array(params.weights)
        self.dp_ref = None
        self.pr_ref = p0[3:]
        self.iw_ref = np.zeros(3)
        self.phi_current = np.array([0.0])
        self.dphi_current = np.array([0.0])
        self.prev_solution = None
        self.lam_g0 = 0
        self.lam_x0 = 0
        self.updated = False

    # ------------------------------------------------------------------ replanning entry
    def update(self, pos_points, rot_points, bp1, br1, e_r_bound, a_sets, b_sets, obstacles, v,
               p0=np.zeros(6), params=None):
        """BoundMPC.py:271-336.  prev_solution, slacks0 and error_count are NOT reset."""
        self.updated = True
        self.split_idxs = [0] + [self.N] * self.nr_segs
        self.switch = False
        self.p0 = p0
        self.obstacles = obstacles
        self.ref_path = ReferencePath(pos_points, rot_points, bp1, br1, e_r_bound, a_sets,
                                      b_sets, self.nr_segs)
        self.phi_max = np.array([self.ref_path.phi_max])
        self.weights = np.array(params.weights)
        dp0 = self.ref_path.dp[0]
        dp0 /= np.linalg.norm(dp0)
        dp1 = self.ref_path.dp[1]
        dp1 /= np.linalg.norm(dp1)
        self.phi_current = np.array([(p0[:3] - pos_points[0]).T @ dp0])
        self.dp_ref = dp0
        self.dphi_current = np.array([v[:3].T @ dp0])
        self.pr_ref = so3.integrate_rotation_reference(
            R.from_matrix(rot_points[0]).as_rotvec(), self.ref_path.dr[0], 0.0, self.phi_current)
        self.iw_ref = self.ref_path.pd[3:, 0] + self.phi_current * self.ref_path.dpd[3:, 0]

    def set_obstacle_sets(self, obs_sets, obs_points_sets):
        """Obstacle polytopes [A,b] and their vertices for the per-step collision sets."""
        self.obs_sets, self.obs_points_sets = obs_sets, obs_points_sets

    # ------------------------------------------------------------------ step: before the solve
    def prepare(self, q0, dq0, ddq0, p0, v0, jerk_current, qf=np.zeros(7), col_pts0=None,
                col_ptsf=None):
        N, S = self.N, self.nr_segs
        p_ref, dp_normed_ref, dp_ref, _, phi_switch = self.ref_path.get_parameters(self.switch)
        self.switch = False
        if self.dp_ref is None:
            self.dp_ref = dp_ref[:3, 0]
        bp1, bp2, br1, br2 = self.ref_path.get_basis_vectors()
        e_r_bound, a_set, b_set = self.ref_path.get_bound_params()

        # warm start: previous solution UNSHIFTED (Q11), with the omega-reversal patch
        if self.prev_solution is None:
            w0 = cold_start(N, q0, p0)
        else:
            w0 = cp.deepcopy(self.prev_solution)
            i_omega = np.reshape(w0[28 * N:34 * N], (6, N))
            if np.linalg.norm(p0[3:] - i_omega[3:, 0]) > 1.5:
                prev_p1 = i_omega[3:, 0].copy()
                i_omega[3:, :-1] = (p0[3:] + (i_omega[3:, 1:].T - prev_p1)).T
                i_omega[3:, -1] = i_omega[3:, -2]
            w0[28 * N:34 * N] = i_omega.flatten()

        prs = [self.pr_ref] + [self.ref_path.r_taud[:, i + 1] for i in range(S - 1)]
        for i in range(S):
            e0, e_par, e_o1, e_o2 = so3.compute_initial_rot_errors(
                p0[3:], prs[i], dp_normed_ref[:, i], br1[:, i], br2[:, i])
            self.dtau_init[:, i] = e0
            self.dtau_init_par[:, i] = e_par
            self.dtau_init_orth1[:, i] = e_o1
            self.dtau_init_orth2[:, i] = e_o2
        v_1, v_2, v_3, jac_dtau_l, jac_dtau_r = so3.orientation_projection_vectors(
            self.dtau_init, self.dtau_init_par, self.dtau_init_orth1, br1, br2, dp_normed_ref)

        # Q10: w_phi rescale only when phi_max < 1; phi target clamped to phi + 5
        x_phi_d = np.array([self.phi_max[0], 0.0, 0.0])
        weights_current = np.copy(self.weights)
        if x_phi_d[0] < 1 and self.phi_max[0] > 0.001:
            weights_current[4] *= min(1.0 / (self.phi_max[0] - self.phi_current[0]) ** 2, 2.0)
        phi_max = np.array([min(self.phi_current[0] + 5.0, self.phi_max[0])])
        x_phi_d[0] = min(self.phi_current[0] + 5.0, x_phi_d[0])

        # collision sets of the 6 collision points (BoundMPC.py:480-497)
        if col_pts0 is None:
            col_pts0 = [self.robot_model.fk_pos_col(q0, i) for i in range(6)]
            col_ptsf = [self.robot_model.fk_pos_col(qf, i) for i in range(6)]
        sizes = COL_JOINT_SIZES if self.robot is None else self.robot["col_joint_sizes"]
        set_joints = []
        for i in range(6):
            a_c, b_c, _ = find_set_collision_avoidance(
                self.obs_sets, self.obs_points_sets, np.asarray(col_pts0[i]),
                np.asarray(col_ptsf[i]), e_max=0.7)
            set_joints.append([a_c, b_c - sizes[i]])
        sets_normed = normalize_set_size(set_joints, 15)
        a_set_joints = [x[0] for x in sets_normed]
        b_set_joints = np.array([x[1] for x in sets_normed])

        params = pack_params(self.split_idxs, self.slacks0, self.iw_ref, self.dtau_init,
                             self.dtau_init_par, self.dtau_init_orth1, self.dtau_init_orth2,
                             x_phi_d, phi_switch, jac_dtau_r, jac_dtau_l, p_ref, dp_ref,
                             dp_normed_ref, bp1, bp2, br1, br2, e_r_bound, weights_current,
                             phi_max, v_1, v_2, v_3, self.qd, a_set, b_set, a_set_joints,
                             b_set_joints)
        lbx, ubx = make_bounds(N, q0, dq0, ddq0, jerk_current, p0, v0, robot=self.robot)
        aux = dict(e_r_bound=e_r_bound, jac_dtau_l=jac_dtau_l, jac_dtau_r=jac_dtau_r,
                   p_ref=p_ref, dp_normed_ref=dp_normed_ref, dp_ref=dp_ref,
                   phi_switch=phi_switch, bp1=bp1, bp2=bp2, br1=br1, br2=br2, v1=v_1, v2=v_2,
                   v3=v_3, x_phi_d=x_phi_d, phi_max=phi_max, a_set=a_set, b_set=b_set,
                   a_set_joints=a_set_joints, b_set_joints=b_set_joints)
        return np.asarray(w0, float), lbx, ubx, params, aux

    # ------------------------------------------------------------------ step
    def step(self, q0, dq0, ddq0, p0, v0, jerk_current, qf=np.zeros(7)):
        """One optimisation step (BoundMPC.py:388-676)."""
        w0, lbx, ubx, params, aux = self.prepare(q0, dq0, ddq0, p0, v0, jerk_current, qf)
        self.last_aux = aux          # sets / bases of this step (mpc_data record)
        t0 = time.perf_counter()
        sol = self.solver(x0=w0, lbx=lbx, ubx=ubx, lbg=None, ubg=None, p=params)
        self.last_cost = float(sol["f"].full().ravel()[0]) if "f" in sol else 0.0
        w_curr = sol["x"].full().flatten()
        time_elapsed = time.perf_counter() - t0
        stats = self.solver.stats()
        traj_data, ref_data, err_data = self.finish(q0, dq0, ddq0, jerk_current, p0, w_curr, stats["success"],
                                                    stats["g_viol"], aux, lam_g=sol["lam_g"], lam_x=sol["lam_x"])
        return traj_data, ref_data, err_data, time_elapsed, stats["iter_count"]

    # ------------------------------------------------------------------ step: after the solve
    def finish(self, q0, dq0, ddq0, jerk_current, p0, w_curr, solver_success, g_viol, aux, lam_g=0, lam_x=0):
        """BoundMPC.py:604-676 + compute_return_data: acceptance test, fallback to the previous solution,
        post-processing.  Split from step() so that a batch of instances can share ONE batched solve."""
        self.slacks0 += w_curr[-6:]          # Q1: adds the last six dpslacks
        success = solver_success or g_viol < 1e-4   # Q8 (BoundMPC.py:613-617)
        using_previous = False
        if not success:
            self.error_count += 1
            if self.prev_solution is not None:
                w_opt = np.copy(self.prev_solution)
            else:
                self.error_count = 0
                w_opt = w_curr
            using_previous = True
        else:
            self.error_count = 0
            w_opt = w_curr
            self.prev_solution = cp.deepcopy(w_opt)
            self.lam_g0, self.lam_x0 = lam_g, lam_x
        from .post import compute_return_data
        return compute_return_data(self, q0, dq0, ddq0, jerk_current, p0, w_opt, using_previous, aux)
