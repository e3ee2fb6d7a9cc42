"""Piecewise-linear pose reference with a sliding window of `nr_segs` segments.

Host-side parameter source of the MPC step; restates the behaviour of
/root/reference/bound_planner/ReferencePath/ReferencePath.py:12-231 (same attribute and method
names, same padding rules, same list mutation of its arguments -- SURVEY Q15).
"""
import numpy as np
from scipy.spatial.transform import Rotation as R

from .so3 import gram_schmidt


class ReferencePath:
    def __init__(self, p, r, bp1, br1, e_r_bound, a_sets, b_sets, nr_segs=2, phi_bias=0):
        self.p, self.r = p, r
        n_pts = len(p)
        self.num_sectors = n_pts - 2
        self.nr_segs = nr_segs
        self.phi_bias = phi_bias
        self.switched = True
        self.sector = 0
        pad = nr_segs - 1

        # bounds/sets are padded with copies of their last entry (ReferencePath.py:43-46)
        self.e_r_bound, self.a_sets, self.b_sets = e_r_bound, a_sets, b_sets
        for lst in (self.e_r_bound, self.a_sets, self.b_sets):
            for _ in range(pad):
                lst.append(lst[-1])

        # orientation increments between via points (ReferencePath.py:51-76)
        self.dr, self.dr_normed, self.iw = [], [], [np.zeros(3)]
        axis_prev = np.array([0.0, 1.0, 0.0])
        for i in range(1, n_pts):
            drot = R.from_matrix(self.r[i] @ self.r[i - 1].T).as_rotvec()
            self.dr.append(drot)
            nrm = np.linalg.norm(drot)
            if nrm > 1e-4:
                axis = drot / nrm
                if np.linalg.norm(axis_prev + axis) < 1e-4:  # keep the projection axis on reversal
                    axis = -axis
            else:
                axis = axis_prev
            self.dr_normed.append(axis)
            axis_prev = np.copy(axis)
            self.iw.append(self.iw[i - 1] + drot)
        for _ in range(pad):
            self.dr.append(np.array(self.dr[-1]))
            self.dr_normed.append(self.dr_normed[-1])
            self.iw.append(self.iw[-1])
            self.r.append(self.r[-1])
        self.r_tau = [R.from_matrix(ri).as_rotvec() for ri in self.r]

        # position increments (ReferencePath.py:78-88)
        self.dp = []
        for i in range(1, n_pts):
            d = self.p[i] - self.p[i - 1]
            if np.linalg.norm(d) < 1e-3:
                d = self.dp[-1] if i > 1 else np.array([0.0, 1.0, 0.0])
            self.dp.append(d)
        for _ in range(pad):
            self.p.append(self.p[-1])
            self.dp.append(self.dp[-1])

        # arc lengths; a pure rotation gets |dr|/pi of path parameter (ReferencePath.py:91-106)
        lengths = []
        for i in range(1, n_pts):
            li = np.linalg.norm(self.p[i] - self.p[i - 1])
            if li < 1e-3:
                li = np.linalg.norm(self.dr[i - 1]) / np.pi
            lengths.append(li)
        self.phi = [0] + lengths + [1] * pad
        self.phi_max = float(np.sum(lengths)) + self.phi_bias if lengths else self.phi_bias

        # orthonormal bases around the position / rotation directions (ReferencePath.py:109-150)
        self.bp1, self.br1, self.bp2, self.br2 = bp1, br1, [], []
        for i in range(len(self.bp1)):
            dirn = self.dp[i] / np.linalg.norm(self.dp[i])
            b = gram_schmidt(dirn, self.bp1[i])
            if abs(b @ self.dp[i]) > 1e-6:
                print(f"[WARNING] Pos Basis vector {i} not orthogonal on path")
            if np.linalg.norm(b) < 1e-3:
                b = gram_schmidt(dirn, np.array([1.0, 1, 1]))
                print(f"[WARNING] Pos Basis vector {i} is too close to direction, using {b}")
            self.bp1[i] = b / np.linalg.norm(b)
            c = np.cross(dirn, self.bp1[i])
            self.bp2.append(c / np.linalg.norm(c))
        for i in range(len(self.bp1)):
            b = gram_schmidt(self.dr_normed[i], self.br1[i])
            if abs(b @ self.dr[i]) > 1e-6:
                print(f"[WARNING] Rot Basis vector {i} not orthogonal on path")
            if np.linalg.norm(b) < 1e-3:
                b = gram_schmidt(self.dr_normed[i], np.array([1.0, 1, 1]))
                print(f"[WARNING] Rot Basis vector {i} is too close to direction, using {b}")
            self.br1[i] = b / np.linalg.norm(b)
            c = np.cross(self.dr_normed[i], self.br1[i])
            self.br2.append(c / np.linalg.norm(c))
        for _ in range(pad):
            self.bp1.append(self.bp1[-1])
            self.br1.append(self.br1[-1])
            self.bp2.append(self.bp2[-1])
            self.br2.append(self.br2[-1])

        # angular velocity per unit path parameter (ReferencePath.py:153-155)
        for i in range(n_pts):
            if self.phi[i + 1] > 1e-8:
                self.dr[i] = self.dr[i] / self.phi[i + 1]

        S = nr_segs
        self.pd = np.zeros((6, S))
        self.r_taud = np.zeros((3, S))
        self.dpd = np.zeros((6, S))
        self.dpd_normed = np.zeros((3, S))
        self.ddpd = np.zeros((6, S))
        self.phi_switch = np.ones(S + 1) * self.phi_bias
        for i in range(S):
            self.set_point(i)

    def set_point(self, idx):
        j = self.sector + idx
        self.pd[:3, idx] = self.p[j]
        self.pd[3:, idx] = self.iw[j]
        self.r_taud[:, idx] = self.r_tau[j]
        self.dpd[:3, idx] = self.dp[j] / np.linalg.norm(self.dp[j])
        self.dpd[3:, idx] = self.dr[j]
        self.dpd_normed[:, idx] = self.dr_normed[j]
        self.phi_switch[idx + 1] = np.cumsum(self.phi)[j + 1] + self.phi_bias

    def update(self, switch):
        if self.sector >= self.num_sectors or not switch:
            self.switched = False
            return
        self.switched = True
        self.sector += 1
        S = self.nr_segs
        self.pd[:, : S - 1] = self.pd[:, 1:].copy()
        self.dpd[:, : S - 1] = self.dpd[:, 1:].copy()
        self.r_taud[:, : S - 1] = self.r_taud[:, 1:].copy()
        self.dpd_normed[:, : S - 1] = self.dpd_normed[:, 1:].copy()
        self.phi_switch[: S - 1] = self.phi_switch[1:S].copy()
        self.phi_switch[S - 1] = self.phi_switch[S] + self.phi_bias
        self.set_point(S - 1)

    def get_parameters(self, switch):
        self.update(switch)
        return self.pd, self.dpd_normed, self.dpd, self.ddpd, self.phi_switch

    def get_basis_vectors(self):
        sl = slice(self.sector, self.sector + self.nr_segs)
        return (np.array(self.bp1[sl]).T, np.array(self.bp2[sl]).T,
                np.array(self.br1[sl]).T, np.array(self.br2[sl]).T)

    def get_bound_params(self):
        sl = slice(self.sector, self.sector + self.nr_segs)
        return np.array(self.e_r_bound[sl]), np.array(self.a_sets[sl]), np.array(self.b_sets[sl])
