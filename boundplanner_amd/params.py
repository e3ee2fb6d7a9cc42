"""Constants, defaults and vector layouts of the BoundMPC step.

  get_default_params / Params     utils/util_functions.py:13-52
  joint/velocity/acc/jerk limits   RobotModel/iiwa.urdf <limit>, RobotModel.py:44-54, BoundMPC.py:171-191
  decision/parameter layouts       casadi_ocp_formulation.py:89-101, 383-415
  normalize_set_size               utils/util_functions.py:121-135
"""
from collections import namedtuple

import numpy as np

Params = namedtuple("Params", ["n", "dt", "build", "weights", "nr_segs"])

NR_JOINTS = 7
NR_SEGS = 4
MAX_SET_SIZE = 15
N_P = 875

# iiwa.urdf <limit lower/upper/velocity>
Q_LIM_UPPER = np.array([2.9670597283903604, 2.0943951023931953, 2.9670597283903604,
                        2.0943951023931953, 2.9670597283903604, 2.0943951023931953,
                        3.0543261909900763])
Q_LIM_LOWER = -Q_LIM_UPPER
DQ_LIM = 10.0 * np.ones(7)
DDQ_LIM = 5.0          # BoundMPC.py:182
U_MAX = 35.0           # RobotModel.py:53-54
COL_JOINT_SIZES = [0.09, 0.12, 0.09, 0.10, 0.07, 0.09, 0.075]  # RobotModel.py:37


def get_default_params():
    w_speed = 0.5
    w_phi, w_dphi = 5.5 * w_speed, 4.06
    scal = 0.5 / w_phi
    weights = np.array([0.05, 0.1, 0.1, 0.01, w_phi * scal, w_dphi * scal, 0.001, 0.0001, 1.0, 10, 500])
    return Params(n=15, dt=0.1, build=True, weights=weights, nr_segs=4)


def n_w(N):
    return 44 * N + 6


def n_g(N):
    return 147 * (N - 1) + 21


def w_offsets(N):
    return {"q": 0, "dq": 7 * N, "ddq": 14 * N, "u": 21 * N, "p": 28 * N, "v": 34 * N,
            "dslacks": 40 * N, "rslacks": 40 * N + 6, "drslacks": 41 * N + 6,
            "pslacks": 42 * N + 6, "dpslacks": 43 * N + 6}


def normalize_set_size(sets, max_set_size=MAX_SET_SIZE):
    """Pad [A, b] to max_set_size rows with (A=0, b=10) (Q13).  Unlike the reference, which only
    prints an error and leaves an oversize set ragged (util_functions.py:126-134), an oversize
    set raises: the downstream parameter vector has no room for it."""
    for s in sets:
        n = s[0].shape[0]
        if n > max_set_size:
            raise ValueError(f"set size {n} exceeds max set size {max_set_size}")
        a = np.zeros((max_set_size, 3))
        b = 10.0 * np.ones(max_set_size)
        a[:n] = s[0]
        b[:n] = s[1]
        s[0], s[1] = a, b
    return sets


def pack_params(split_idxs, slacks0, iw_ref, dtau_init, dtau_init_par, dtau_init_orth1,
                dtau_init_orth2, x_phi_d, phi_switch, jac_dtau_r, jac_dtau_l, p_ref, dp_ref,
                dp_normed_ref, bp1, bp2, br1, br2, e_r_bound, weights, phi_max, v_1, v_2, v_3, qd,
                a_set, b_set, a_set_joints, b_set_joints):
    """The 875-vector in the order of BoundMPC.py:507-542 (shapes as produced there)."""
    parts = [np.asarray(split_idxs, float), slacks0, iw_ref, dtau_init.T.ravel(),
             dtau_init_par.T.ravel(), dtau_init_orth1.T.ravel(), dtau_init_orth2.T.ravel(),
             x_phi_d, phi_switch, jac_dtau_r.T.ravel(), jac_dtau_l.T.ravel(), p_ref.ravel(),
             dp_ref.ravel(), dp_normed_ref.ravel(), bp1.ravel(), bp2.ravel(), br1.ravel(),
             br2.ravel(), np.asarray(e_r_bound).T.ravel(), weights, np.atleast_1d(phi_max),
             v_1.ravel(), v_2.ravel(), v_3.ravel(), qd]
    parts += [np.asarray(a).T.ravel() for a in a_set]
    parts.append(np.asarray(b_set).T.ravel())
    parts += [np.asarray(a).T.ravel() for a in a_set_joints]
    parts.append(np.asarray(b_set_joints).T.ravel())
    out = np.concatenate([np.asarray(x, float).ravel() for x in parts])
    assert out.size == N_P, out.size
    return out


def make_bounds(N, q0, dq0, ddq0, jerk0, p0, v0, robot=None):
    """lbx/ubx of BoundMPC.py:544-589: limits, stage 0 pinned, slacks >= 0.  robot: a table of boundplanner_amd.robots
    (None = iiwa14); infinite joint limits (+-1e20 in a table) become +-inf like the reference's (RobotModel.py:46-48)."""
    inf = np.inf
    if robot is None:
        ql, qh, dqm, ddqm, um = Q_LIM_LOWER, Q_LIM_UPPER, DQ_LIM, DDQ_LIM, U_MAX
    else:
        ql = np.where(np.asarray(robot["q_lower"], float) <= -1e19, -inf, np.asarray(robot["q_lower"], float))
        qh = np.where(np.asarray(robot["q_upper"], float) >= 1e19, inf, np.asarray(robot["q_upper"], float))
        dqm, ddqm, um = np.asarray(robot["dq_max"], float), robot["ddq_max"], robot["u_max"]
    lo = [np.repeat(ql, N), np.repeat(-dqm, N), -ddqm * np.ones(7 * N),
          -um * np.ones(7 * N), -inf * np.ones(6 * N), -inf * np.ones(6 * N)]
    hi = [np.repeat(qh, N), np.repeat(dqm, N), ddqm * np.ones(7 * N),
          um * np.ones(7 * N), inf * np.ones(6 * N), inf * np.ones(6 * N)]
    for arr_l, arr_h, val in zip(lo, hi, (q0, dq0, ddq0, jerk0, p0, v0)):
        arr_l[0:-1:N] = val      # Q7: x[0:-1:N] = value on joint-major arrays
        arr_h[0:-1:N] = val
    nsl = 6 + 4 * N
    lbx = np.concatenate(lo + [np.zeros(nsl)])
    ubx = np.concatenate(hi + [inf * np.ones(nsl)])
    return lbx, ubx


def cold_start(N, q0, p0):
    """Initial guess of BoundMPC.py:412-416."""
    w0 = np.zeros(n_w(N))
    w0[0:7 * N] = np.repeat(q0, N)
    w0[28 * N:34 * N] = np.repeat(p0, N)
    return w0
