"""SO(3) helpers and orientation-error preparation used by the host side of the MPC step.

Restates (numpy only):
  jac_SO3_inv_left/right, skew_matrix, rodrigues_matrix   utils/optimization_functions.py:35-104
  integrate_rotation_reference, compute_initial_rot_errors BoundMPC/bound_mpc_functions.py:16-46
  BoundMPC.compute_orientation_projection_vectors          BoundMPC/BoundMPC.py:338-386
  gram_schmidt                                             utils/util_functions.py:110-118
"""
import numpy as np
from scipy.spatial.transform import Rotation as R


def skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def _jac_inv(axis, sign):
    # Q14: angle = |axis| + 1e-6 (optimization_functions.py:37,54)
    angle = np.linalg.norm(axis) + 1e-6
    W = skew(axis)
    coef = 1.0 / angle**2 - (1.0 + np.cos(angle)) / (2.0 * angle * np.sin(angle))
    return np.eye(3) + sign * 0.5 * W + coef * (W @ W)


def jac_SO3_inv_right(axis):
    return _jac_inv(np.asarray(axis, dtype=float), +1.0)


def jac_SO3_inv_left(axis):
    return _jac_inv(np.asarray(axis, dtype=float), -1.0)


def rodrigues_matrix(omega, phi):
    W = skew(omega)
    return np.eye(3) + np.sin(phi) * W + (1.0 - np.cos(phi)) * (W @ W)


def gram_schmidt(v, b):
    """One Gram-Schmidt step: remove from b its component along v."""
    return b - (v @ b) * v


def integrate_rotation_reference(pr_ref, omega, phi0, phi1):
    r0 = R.from_rotvec(pr_ref).as_matrix()
    n = np.linalg.norm(omega)
    if n > 1e-4:
        r1 = rodrigues_matrix(omega / n, float(np.squeeze(phi1 - phi0)) * n) @ r0
    else:
        r1 = r0
    return R.from_matrix(r1).as_rotvec()


def compute_initial_rot_errors(pr, pr_ref, dp_normed_ref, br1, br2):
    tauc = R.from_rotvec(pr).as_matrix()
    taud = R.from_rotvec(pr_ref).as_matrix()
    dtau_init = R.from_matrix(tauc @ taud.T).as_rotvec()
    r01 = np.column_stack((br2, dp_normed_ref, br1))
    dtau_01 = r01.T @ R.from_rotvec(dtau_init).as_matrix() @ r01
    eul = R.from_matrix(dtau_01).as_euler("zyx")
    # a half turn comes out as +pi or -pi depending on the sign of a floating-point zero in dtau_01 (structural zeros of the
    # padded segments' bases): made deterministic (+pi) here and on the device (csrc/bmpc_loop.hpp lp_euler_zyx)
    for i in (0, 2):
        if abs(abs(eul[i]) - np.pi) < 1e-12:
            eul[i] = np.pi
    return [dtau_init, eul[1] * dp_normed_ref, eul[0] * br1, eul[2] * br2]


def orientation_projection_vectors(dtau_init, dtau_init_par, dtau_init_orth1, br1, br2, dp_normed_ref):
    """Returns v_1, v_2, v_3 (3 x S each), jac_dtau_l, jac_dtau_r.  One jac_dtau_l/r, taken from
    segment 0, serves all segments (Q14)."""
    S = dp_normed_ref.shape[1]
    jac_r = jac_SO3_inv_right(dtau_init[:, 0])
    jac_l = jac_SO3_inv_left(dtau_init[:, 0])
    v_1 = np.empty((3, S))
    v_2 = np.empty((3, S))
    v_3 = np.empty((3, S))
    r_init0 = R.from_rotvec(dtau_init[:, 0]).as_matrix()
    for i in range(S):
        rest1 = r_init0 @ R.from_rotvec(dtau_init_orth1[:, i]).as_matrix().T
        rest2 = rest1 @ R.from_rotvec(dtau_init_par[:, i]).as_matrix().T
        jac_r1 = jac_SO3_inv_right(R.from_matrix(rest1).as_rotvec())
        jac_r2 = jac_SO3_inv_right(R.from_matrix(rest2).as_rotvec())
        g = jac_r @ br1[:, i]
        h = jac_r1 @ dp_normed_ref[:, i]
        k = jac_r2 @ br2[:, i]
        # rows of the inverse of [g h k] (dual basis), written through the Gram matrix
        Bm = np.column_stack((g, h, k))
        dual = np.linalg.solve(Bm.T @ Bm, Bm.T)
        v_1[:, i], v_2[:, i], v_3[:, i] = dual[0], dual[1], dual[2]
    return v_1, v_2, v_3, jac_l, jac_r
