"""Post-processing of one MPC step: the part of the reference's BoundMPC.step() after the solver call
(/root/reference/bound_planner/BoundMPC/BoundMPC.py:678-1040, `compute_return_data`) together with
the numpy branches of `reference_function` / `error_function`
(bound_mpc_functions.py:49-390, mpc_utils_casadi.py:6-70) it evaluates per horizon stage.

Restated, not copied: per-stage segment selection by split index, path parameter and reference
pose, position / orientation error decomposition, integration of the rotation reference, the
split-index countdown with its switch test (in-set, rotation-bound and path-parameter conditions),
via-point adaptation and the monotone split fix.  Sequential, data-dependent host logic exactly as
in the reference; pinned by tests/golden/closed_loop.npz (reference code run under stubs).
"""
import numpy as np
from scipy.spatial.transform import Rotation as R

from .so3 import integrate_rotation_reference

IN_SET_ACCURACY = 0.005          # BoundMPC.py:917
ROT_MARGIN = 5 * np.pi / 180     # BoundMPC.py:963-964
PHI_SWITCH_MARGIN = 0.03         # BoundMPC.py:927


def _segment(idx, split_idx, n_rows):
    """Row i of a per-segment table used at stage `idx` (get_current_segments_split, numpy branch):
    the last i+1 with idx > split_idx[i+1]."""
    i = 0
    for j in range(n_rows - 2):
        if idx > split_idx[j + 1]:
            i = j + 1
    return i


def _next_selector(split_idx, n_horizon):
    """Index of the 'next' set / rotation-error tables (Q5: 1, 2 or 3 by the first unset split)."""
    if split_idx[1] == n_horizon:
        return 1
    if split_idx[2] == n_horizon:
        return 2
    return 3


def stage_reference(aux, split_idx, idx, n_horizon, p, v):
    """reference_function (numpy branch) at horizon stage `idx` for pose p (6) and velocity v (6)."""
    S = aux["p_ref"].shape[1]
    s = _segment(idx, split_idx, S)
    dp_d, dp_n = aux["dp_ref"][:, s].copy(), aux["dp_ref"][:, s + 1]
    p_c, p_n = aux["p_ref"][:, s], aux["p_ref"][:, s + 1]
    phi_start = aux["phi_switch"][_segment(idx, split_idx, S + 1)]     # its table has S+1 rows: one more candidate
    phi_l = float((p[:3] - p_c[:3]) @ dp_d[:3])
    phi_next = float((p[:3] - p_n[:3]) @ dp_n[:3])
    dphi = float(v[:3] @ dp_d[:3])
    p_d = np.concatenate((p_c[:3] + dp_d[:3] * phi_l, dp_d[3:] * phi_l + p_c[3:]))
    p_dr_next = dp_n[3:] * phi_next + p_n[3:]
    nxt = _next_selector(split_idx, n_horizon)
    erb = aux["e_r_bound"]
    # a_current follows get_current_segments_1d (first result only), b_current the split selector
    ia = 0
    for j in range(len(aux["a_set"]) - 2):
        if idx > split_idx[j + 1]:
            ia = j + 1
    out = dict(
        p_d=p_d, p_dr_next=p_dr_next, p_r_omega0=p_c[3:], dp_d=dp_d, ddp_d=0.0 * dp_d,
        bp1=aux["bp1"][:, s], bp2=aux["bp2"][:, s], br1=aux["br1"][:, s], br2=aux["br2"][:, s],
        br1_next=aux["br1"][:, s + 1], br2_next=aux["br2"][:, s + 1],
        dp_normed=aux["dp_normed_ref"][:, s], dp_normed_next=aux["dp_normed_ref"][:, s + 1],
        v1=aux["v1"][:, s], v2=aux["v2"][:, s], v3=aux["v3"][:, s],
        v1_next=aux["v1"][:, s + 1], v2_next=aux["v2"][:, s + 1], v3_next=aux["v3"][:, s + 1],
        r_bound_upper=erb[s][:3], r_bound_lower=erb[s][3:],
        r_bound_upper_next=erb[s + 1][:3], r_bound_lower_next=erb[s + 1][3:],
        a_current=aux["a_set"][ia], b_current=aux["b_set"][s], a_next=aux["a_set"][nxt], b_next=aux["b_set"][nxt],
        phi_end_seg=aux["phi_switch"][nxt + 1],          # Q5: [2|3|4] in the numpy twin
        phi=phi_l + phi_start, dphi=dphi, phi_start=phi_start, seg=s)
    return out


def stage_errors(mpc, aux, ref, split_idx, idx, n_horizon, p, v, i_omega_0, iw_ref_0):
    """error_function (numpy branch): position error split along / across the path, first-order
    Lie-space orientation error and its decomposition on the current and the next segment."""
    dp3, ddp3 = ref["dp_d"][:3], ref["ddp_d"][:3]
    dphi = ref["dphi"]
    e = p[:3] - ref["p_d"][:3]
    e_par = (dp3 @ e) * dp3
    e_orth = e - e_par
    de = v[:3] - dp3 * dphi
    de_par = (dp3 @ de) * dp3 + ((ddp3 * dphi) @ e) * dp3 + (dp3 @ e) * ddp3 * dphi
    de_orth = de - de_par
    i_w_ref_0 = iw_ref_0 if idx <= split_idx[1] else ref["p_r_omega0"]
    s = ref["seg"]
    nxt = _next_selector(split_idx, n_horizon)
    jl, jr = aux["jac_dtau_l"], aux["jac_dtau_r"]
    e_init, e_initn = mpc.dtau_init[:, s], mpc.dtau_init[:, nxt]
    dlie = jl @ (p[3:] - i_omega_0)
    e_r = e_init + dlie - jr @ (ref["p_d"][3:] - i_w_ref_0)
    e_rn = e_initn + dlie - jr @ (ref["p_dr_next"] - i_w_ref_0)
    de_r = jl @ v[3:] - jr @ (ref["dp_d"][3:] * dphi)
    d, dn = e_r - e_init, e_rn - e_initn
    out = dict(
        e_p=e, de_p=de, e_p_par=e_par, e_p_orth=e_orth, de_p_par=de_par, de_p_orth=de_orth, e_r=e_r, de_r=de_r,
        e_r_orth1=mpc.dtau_init_orth1[:, s] + (d @ ref["v1"]) * ref["br1"],
        e_r_par=mpc.dtau_init_par[:, s] + (d @ ref["v2"]) * ref["dp_normed"],
        e_r_orth2=mpc.dtau_init_orth2[:, s] + (d @ ref["v3"]) * ref["br2"],
        e_r_orth1n=mpc.dtau_init_orth1[:, s + 1] + (dn @ ref["v1_next"]) * ref["br1_next"],
        e_r_parn=mpc.dtau_init_par[:, s + 1] + (dn @ ref["v2_next"]) * ref["dp_normed_next"],
        e_r_orth2n=mpc.dtau_init_orth2[:, s + 1] + (dn @ ref["v3_next"]) * ref["br2_next"])
    return out


def compute_return_data(mpc, q0, dq0, ddq0, jerk_current, p0, w_opt, using_previous, aux):
    N, ec = mpc.N, mpc.error_count
    grab = lambda lo, n: np.reshape(w_opt[lo:lo + n * N], (N, n), "F").T[:, ec:]
    opt_q, opt_dq, opt_ddq, opt_jerk = grab(0, 7), grab(7 * N, 7), grab(14 * N, 7), grab(21 * N, 7)
    opt_traj, opt_vel = grab(28 * N, 6), grab(34 * N, 6)
    pslacks = w_opt[-2 * N:-N]
    n = opt_jerk.shape[1]
    opt_phi, opt_dphi = np.empty(n), np.empty(n)
    iw_ref_0 = np.copy(mpc.iw_ref)
    split_prev = list(mpc.split_idxs)
    ref_data = {k: [None] * n for k in (
        "p", "dp", "ddp", "dp_normed", "dp_normedn", "bp1", "bp2", "br1", "br2", "br1_next", "br2_next", "v1", "v2", "v3",
        "v1_next", "v2_next", "v3_next", "p_r_omega0", "r_bound_lower", "r_bound_upper", "r_bound_lower_next",
        "r_bound_upper_next")}
    err_data = {k: [None] * n for k in (
        "e_p", "de_p", "e_p_par", "e_p_orth", "de_p_par", "de_p_orth", "e_r", "de_r", "e_r_par", "e_r_orth1", "e_r_orth2",
        "e_r_parn", "e_r_orth1n", "e_r_orth2n")}
    for i in range(n):
        ref = stage_reference(aux, split_prev, i, N, opt_traj[:, i], opt_vel[:, i])
        opt_phi[i], opt_dphi[i] = ref["phi"], ref["dphi"]
        for key, src in (("p", "p_d"), ("dp", "dp_d"), ("ddp", "ddp_d"), ("dp_normed", "dp_normed"),
                         ("dp_normedn", "dp_normed_next"), ("bp1", "bp1"), ("bp2", "bp2"), ("br1", "br1"), ("br2", "br2"),
                         ("br1_next", "br1_next"), ("br2_next", "br2_next"), ("v1", "v1"), ("v2", "v2"), ("v3", "v3"),
                         ("v1_next", "v1_next"), ("v2_next", "v2_next"), ("v3_next", "v3_next"),
                         ("p_r_omega0", "p_r_omega0"), ("r_bound_lower", "r_bound_lower"),
                         ("r_bound_upper", "r_bound_upper"), ("r_bound_lower_next", "r_bound_lower_next"),
                         ("r_bound_upper_next", "r_bound_upper_next")):
            ref_data[key][i] = ref[src]
        if i == 1:
            ref_data["a_current"], ref_data["b_current"] = ref["a_current"].flatten(), ref["b_current"]
            ref_data["a_next"], ref_data["b_next"] = ref["a_next"].flatten(), ref["b_next"]
            for name, j in (("j3", 0), ("j5", 1), ("j6", 2), ("j67", 3), ("elbow", 4)):
                ref_data["a_" + name] = aux["a_set_joints"][j].flatten()
                ref_data["b_" + name] = aux["b_set_joints"][j]
        err = stage_errors(mpc, aux, ref, split_prev, i, N, opt_traj[:, i], opt_vel[:, i], p0[3:], iw_ref_0)
        for key in ("e_p", "de_p", "e_p_par", "e_p_orth", "de_p_par", "de_p_orth", "de_r"):
            err_data[key][i] = err[key]
        err_data["e_r"][i] = np.copy(err["e_r"])
        # scalar coordinates of the decomposed orientation errors along their basis vectors
        err_data["e_r_par"][i] = err["e_r_par"] @ ref["dp_normed"]
        err_data["e_r_orth1"][i] = err["e_r_orth1"] @ ref["br1"]
        err_data["e_r_orth2"][i] = err["e_r_orth2"] @ ref["br2"]
        err_data["e_r_parn"][i] = err["e_r_parn"] @ ref["dp_normed_next"]
        err_data["e_r_orth1n"][i] = err["e_r_orth1n"] @ ref["br1_next"]
        err_data["e_r_orth2n"][i] = err["e_r_orth2n"] @ ref["br2_next"]

    # rotation reference integrated to the path parameter of stage 1 (BoundMPC.py:894-914)
    p_ref, dp_ref, phi_switch = aux["p_ref"], aux["dp_ref"], aux["phi_switch"]
    rp = mpc.ref_path
    j = 1 if mpc.split_idxs[1] == 1 else 0
    mpc.pr_ref = integrate_rotation_reference(R.from_matrix(rp.r[rp.sector + j]).as_rotvec(), dp_ref[3:, j],
                                              phi_switch[j], opt_phi[1])
    mpc.iw_ref = p_ref[3:, j] + (opt_phi[1] - phi_switch[j]) * dp_ref[3:, j]

    # split indices: countdown of a set switch, or detection of a new one (BoundMPC.py:916-1021)
    a_set, b_set = aux["a_set"], aux["b_set"]
    for i in range(1, mpc.nr_segs - 1):
        if mpc.split_idxs[i] < N:
            mpc.split_idxs[i] -= 1
            if mpc.split_idxs[i] == 0:
                mpc.switch = True
                mpc.split_idxs[i] = N
        elif mpc.error_count == 0:
            dswitch = opt_phi > phi_switch[i] - PHI_SWITCH_MARGIN
            d0 = np.max(a_set[i - 1] @ opt_traj[:3, :] - b_set[i - 1][:, None], axis=0)
            d1 = np.max(a_set[i] @ opt_traj[:3, :] - b_set[i][:, None], axis=0)
            in_set0 = d0 < IN_SET_ACCURACY + pslacks
            in_set1 = d1 < IN_SET_ACCURACY + pslacks
            e_rs = np.vstack((err_data["e_r_orth1"], err_data["e_r_par"], err_data["e_r_orth2"])).T
            e_rsn = np.array([err_data["e_r_orth1n"], err_data["e_r_parn"], err_data["e_r_orth2n"]]).T
            lo, up = np.array(ref_data["r_bound_lower"]), np.array(ref_data["r_bound_upper"])
            lon, upn = np.array(ref_data["r_bound_lower_next"]), np.array(ref_data["r_bound_upper_next"])
            in_rot = (e_rs < up) * (e_rs > lo) * (e_rsn < upn + ROT_MARGIN) * (e_rsn > lon - ROT_MARGIN)
            in_rot = np.min(in_rot, axis=1)
            # only the trailing run of in-set stages counts for the next set
            outside = np.where(in_set1 == False)[0]  # noqa: E712
            if outside.shape[0] > 0:
                in_set1[:outside[-1]] = False
            idx_new = np.where(dswitch * in_set0 * in_set1 * in_rot)[0]
            not_at_end = rp.sector + (i - 1) < rp.num_sectors
            if idx_new.shape[0] > 0 and not_at_end:
                if mpc.split_idxs[i] == N:
                    mpc.split_idxs[i] = idx_new[0] - 1
                    # move the via point onto the switching position (BoundMPC.py:989-1011)
                    sec = rp.sector
                    dp, pv = dp_ref[:3, i], p_ref[:3, i]
                    corr = (opt_traj[:3, idx_new[0]] - pv) @ dp
                    pv_new = pv + corr * dp
                    rp.pd[:3, i] = pv_new
                    rp.p[sec + i] = pv_new
                    rp.phi[sec + i + 1] -= corr
                    rp.phi_switch[i + 1:] -= corr
                    rp.phi_max = np.array(rp.phi).cumsum()[rp.num_sectors + 1] + rp.phi_bias
                    mpc.phi_max = np.array([rp.phi_max])
                if mpc.split_idxs[i] == 0:
                    mpc.switch = True
    if mpc.switch:
        mpc.split_idxs[1:-1] = mpc.split_idxs[2:]
        mpc.split_idxs[-1] = N
    for i in range(1, phi_switch.shape[0] - 1):
        if mpc.split_idxs[i] <= mpc.split_idxs[i - 1]:
            mpc.split_idxs[i] = min(N, mpc.split_idxs[i - 1] + 1)

    mpc.phi_current = np.array([opt_phi[1]])
    mpc.dphi_current = np.array([opt_dphi[1]])
    ref_data["p"][0] = np.concatenate((ref_data["p"][0][:3], mpc.pr_ref))
    traj_data = dict(p=opt_traj[:, 1:], v=opt_vel[:, 1:], a=opt_vel[:, 1:],       # Q12: "a" aliases velocity
                     q=opt_q[:, 1:], dq=opt_dq[:, 1:], ddq=opt_ddq[:, 1:], dddq=opt_jerk,
                     phi=opt_phi[1:], dphi=opt_dphi[1:])
    return traj_data, ref_data, err_data
