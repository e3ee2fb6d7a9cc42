"""The trace record of one MPC step with the field set of the reference's ROS message
`boundmpcmsg/msg/MPCData.msg:1-64`.

The reference defines the message but no code in its repository fills it (its publishers live in the RViz / ROS layer that
is out of scope); the record below therefore maps every field NAME of the message onto the quantities `BoundMPC.step`
returns under the same names (`traj_data`, `ref_data`, `err_data`, `BoundMPC.py:678-1040`) and onto the carried state.
Fields the tracker does not produce (path-parameter acceleration / jerk, `e_p_off`, `e_r_off`, `p_lower`, `p_upper`) are
present and empty, so that a consumer written against the message finds every key."""
import numpy as np

# message field -> (source, key); sources: "traj", "ref", "err" (dicts returned by BoundMPC.step), "node", "mpc"
FIELDS = ["t_comp", "t_loop", "t_overhead", "phi_max", "cost", "iterations", "t_switch", "phi_switch", "fails",
          "p", "v", "a", "q", "dq", "ddq", "dddq", "phi", "dphi", "ddphi", "dddphi", "sector", "phi_switch_vector",
          "e_p", "de_p", "e_p_par", "e_p_orth", "de_p_par", "de_p_orth", "e_r", "de_r", "e_r_par", "e_r_orth1", "e_r_orth2",
          "p_ref", "dp_ref", "ddp_ref", "dp_normed_ref", "p_lower", "p_upper", "e_p_off", "e_r_off", "bp1", "bp2", "br1", "br2",
          "a_set_j3", "a_set_j5", "a_set_j6", "a_set_j67", "a_set_elbow", "a_set", "a_set_next",
          "b_set_j3", "b_set_j5", "b_set_j6", "b_set_j67", "b_set_elbow", "b_set", "b_set_next"]

# collision points of RobotModel.py:27-35 in the message's naming: joint_3, joint_5, joint_6, joint_6/7 (joint_7 origin), elbow
_COL = {"j3": 0, "j5": 2, "j6": 3, "j67": 4, "elbow": 5}


def _cols(a):
    a = np.asarray(a, float)
    return [a[:, k].copy() for k in range(a.shape[1])] if a.ndim == 2 else [a.copy()]


def _list(rows):
    return [np.asarray(r, float).ravel() for r in rows if np.size(r)]


def mpc_data(traj_data, ref_data, err_data, mpc, aux=None, t_comp=0.0, t_loop=0.0, t_overhead=0.0, cost=0.0, iterations=0, fails=()):
    """One MPCData record (dict keyed by the message's field names; `Vector[]` fields are lists of 1-D arrays, one per
    horizon stage; `Vector` fields 1-D arrays).  `mpc`: the BoundMPC object after the step; `aux`: the dict returned by
    BoundMPC.prepare (sets, bases) when available."""
    rp = mpc.ref_path
    d = {k: [] for k in FIELDS}
    d.update(t_comp=float(t_comp), t_loop=float(t_loop), t_overhead=float(t_overhead), phi_max=float(np.ravel(mpc.phi_max)[0]),
             cost=float(cost), iterations=int(iterations), fails=np.asarray(fails, float), sector=int(rp.sector),
             t_switch=np.zeros(0), phi_switch=np.asarray(rp.phi_switch, float).copy(),
             phi_switch_vector=np.asarray(rp.phi_switch, float).copy(), ddphi=np.zeros(0), dddphi=np.zeros(0))
    for k in ("p", "v", "a", "q", "dq", "ddq", "dddq"):
        d[k] = _cols(traj_data[k])
    d["phi"], d["dphi"] = np.asarray(traj_data["phi"], float).copy(), np.asarray(traj_data["dphi"], float).copy()
    for k in ("e_p", "de_p", "e_p_par", "e_p_orth", "de_p_par", "de_p_orth", "e_r", "de_r", "e_r_par", "e_r_orth1", "e_r_orth2"):
        d[k] = _list(err_data.get(k, []))
    for msg, key in (("p_ref", "p"), ("dp_ref", "dp"), ("ddp_ref", "ddp"), ("dp_normed_ref", "dp_normed"), ("bp1", "bp1"),
                     ("bp2", "bp2"), ("br1", "br1"), ("br2", "br2")):
        d[msg] = _list(ref_data.get(key, []))
    if aux is not None:
        aj, bj = aux["a_set_joints"], np.asarray(aux["b_set_joints"], float)
        for name, c in _COL.items():
            d["a_set_" + name] = np.asarray(aj[c], float).ravel()
            d["b_set_" + name] = bj[c].copy()
        a_set, b_set = np.asarray(aux["a_set"], float), np.asarray(aux["b_set"], float)
        d["a_set"], d["b_set"] = a_set[0].ravel(), b_set[0].copy()
        if a_set.shape[0] > 1:
            d["a_set_next"], d["b_set_next"] = a_set[1].ravel(), b_set[1].copy()
    return d


def from_node(node):
    """MPCData record of the step an `MPCNode` has just taken."""
    return mpc_data(node.traj_data, node.ref_data, node.err_data, node.mpc, aux=getattr(node.mpc, "last_aux", None),
                    t_comp=node.t_mpc, cost=getattr(node.mpc, "last_cost", 0.0), iterations=node.iters[-1] if node.iters else 0,
                    fails=node.fails)


def from_device_record(rec, N, fails=()):
    """MPCData record from one raw record of the device-resident loop (include/boundmpc.h bmpc_loop_records; written by
    bmpc_loop_k_finish, layout csrc/bmpc_loop.hpp LP_REC_*): the same keys and conventions as mpc_data() -- trajectories from
    stage 1 on (dddq from stage 0, BoundMPC.py:1024-1040), errors and references from stage 0 on, `a` aliasing `v` (Q12)."""
    rec = np.asarray(rec, float)
    H, W = 16, 64
    n = int(rec[4])
    st = rec[H:H + W * N].reshape(N, W)[:n]
    d = {k: [] for k in FIELDS}
    d.update(t_comp=0.0, t_loop=0.0, t_overhead=0.0, phi_max=float(rec[6]), cost=0.0, iterations=int(rec[0]),
             fails=np.asarray(fails, float), sector=int(rec[5]), t_switch=np.zeros(0), phi_switch=np.zeros(0),
             phi_switch_vector=np.zeros(0), ddphi=np.zeros(0), dddphi=np.zeros(0))
    d["status"], d["viol"], d["error_count"], d["split_idxs"] = int(rec[1]), float(rec[2]), int(rec[3]), rec[7:12].astype(int)
    col = lambda lo, hi, first=1: [st[i, lo:hi].copy() for i in range(first, n)]
    d["p"], d["v"], d["a"] = col(0, 6), col(6, 12), col(6, 12)
    d["q"], d["dq"], d["ddq"], d["dddq"] = col(12, 19), col(19, 26), col(26, 33), col(33, 40, first=0)
    d["phi"], d["dphi"] = st[1:, 40].copy(), st[1:, 41].copy()
    d["e_p"], d["de_p"], d["e_r"], d["de_r"] = col(42, 45, 0), col(45, 48, 0), col(48, 51, 0), col(51, 54, 0)
    d["e_r_orth1"], d["e_r_par"], d["e_r_orth2"] = [st[i, 54:55].copy() for i in range(n)], [st[i, 55:56].copy() for i in range(n)], [st[i, 56:57].copy() for i in range(n)]
    d["p_ref"] = col(57, 63, 0)
    d["segment"] = st[:, 63].astype(int)
    # sets: parameter-vector blocks a_set [4 segments][15 x 3 column-major], b_set [15][4], a_set_joints [6][15 x 3], b_set_joints [15][6]
    sets = rec[H + W * N:]
    a_set = sets[:180].reshape(4, 3, 15).transpose(0, 2, 1)
    b_set = sets[180:240].reshape(15, 4).T
    a_j = sets[240:510].reshape(6, 3, 15).transpose(0, 2, 1)
    b_j = sets[510:600].reshape(15, 6).T
    seg1 = int(st[1, 63]) if n > 1 else 0
    d["a_set"], d["b_set"] = a_set[seg1].ravel(), b_set[seg1].copy()
    nxt = int(rec[12])
    d["a_set_next"], d["b_set_next"] = a_set[nxt].ravel(), b_set[nxt].copy()
    for name, c in _COL.items():
        d["a_set_" + name], d["b_set_" + name] = a_j[c].ravel(), b_j[c].copy()
    return d
