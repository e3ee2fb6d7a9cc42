"""Device-resident closed loop (BASELINE.json configs[4]): R rollouts advanced in lock step on one
MI355X, every step = prepare kernel -> batched NLP solve -> finish kernel, with the per-rollout state
(robot state, ReferencePath window, split indices, warm start) living in HBM.  Only the plan-time
construction runs on the host: `BoundMPC.__init__/update` + `ReferencePath.__init__`
(/root/reference/bound_planner/BoundMPC/BoundMPC.py:28-336, ReferencePath/ReferencePath.py:12-157) build
the Python objects, `pack_state` serialises them into the state vector of
boundplanner_amd/csrc/bmpc_loop.hpp, and the HIP kernels carry on from there
(BoundMPC.step / compute_return_data / MPCNode.step, BoundMPC.py:388-1040, MPCNode.py:106-160).

No CPU fallback: `DeviceLoop` needs libboundmpc_hip.so and a GPU.  Scenes with obstacles (per-step
collision sets from the host finder, collision_sets.py) are refused.
"""
import ctypes

import numpy as np

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)

FIELDS = ["q", "dq", "ddq", "jerk", "qf", "v", "p_lie", "split", "sw", "error_count", "has_prev", "slacks0", "pr_ref",
          "iw_ref", "phi_current", "dphi_current", "phi_max", "weights", "dtau", "dtau_par", "dtau_o1", "dtau_o2",
          "jac_l", "jac_r", "v1", "v2", "v3", "patch", "patch_delta", "accept", "dead", "steps", "rp_sector",
          "rp_num_sectors", "rp_phi_bias", "rp_phi_max", "rp_p", "rp_r_tau", "rp_dr", "rp_drn", "rp_iw", "rp_dp",
          "rp_phi", "rp_bp1", "rp_bp2", "rp_br1", "rp_br2", "rp_erb", "rp_a", "rp_b", "rp_pd", "rp_r_taud", "rp_dpd",
          "rp_dpdn", "rp_phi_switch"]
NL = 11          # LP_NL of bmpc_loop.hpp: via points + nr_segs-1 padded copies


def read_layout(field_fn, size_fn):
    """name -> (offset, count) from the library that will consume the state (HIP library or, in tests, the CPU
    build of the same header)."""
    lay = {}
    for name in FIELDS:
        off, cnt = ctypes.c_int(), ctypes.c_int()
        if field_fn(name.encode(), ctypes.byref(off), ctypes.byref(cnt)) != 0:
            raise RuntimeError(f"state field {name} unknown to the library")
        lay[name] = (off.value, cnt.value)
    lay["_size"] = size_fn()
    return lay


def _rows(lst, width, n=NL):
    out = np.zeros((n, width))
    if len(lst) > n:
        raise ValueError(f"reference path with {len(lst)} list entries exceeds the device loop's {n} (8 via points)")
    for i, a in enumerate(lst):
        out[i] = np.asarray(a, float).reshape(-1)
    return out.reshape(-1)


def pack_state(lay, mpc, q, dq, ddq, jerk, qf, v, p_lie):
    """Serialise one rollout: a host BoundMPC object (after __init__ / update) + the node state."""
    if getattr(mpc, "obs_sets", None):
        raise ValueError("the device loop handles obstacle-free scenes only (per-step collision sets are host code)")
    S = np.zeros(lay["_size"])

    def put(name, val):
        off, cnt = lay[name]
        a = np.asarray(val, float).reshape(-1)
        assert a.size == cnt, (name, a.size, cnt)
        S[off:off + cnt] = a

    for name, val in (("q", q), ("dq", dq), ("ddq", ddq), ("jerk", jerk), ("qf", qf), ("v", v), ("p_lie", p_lie)):
        put(name, val)
    rp = mpc.ref_path
    put("split", mpc.split_idxs); put("sw", float(mpc.switch)); put("error_count", mpc.error_count)
    put("has_prev", 0.0 if mpc.prev_solution is None else 1.0)
    put("slacks0", mpc.slacks0); put("pr_ref", mpc.pr_ref); put("iw_ref", mpc.iw_ref)
    put("phi_current", mpc.phi_current); put("dphi_current", mpc.dphi_current); put("phi_max", mpc.phi_max)
    put("weights", mpc.weights)
    put("rp_sector", rp.sector); put("rp_num_sectors", rp.num_sectors); put("rp_phi_bias", rp.phi_bias)
    put("rp_phi_max", rp.phi_max)
    put("rp_p", _rows(rp.p, 3)); put("rp_r_tau", _rows(rp.r_tau, 3)); put("rp_dr", _rows(rp.dr, 3))
    put("rp_drn", _rows(rp.dr_normed, 3)); put("rp_iw", _rows(rp.iw, 3)); put("rp_dp", _rows(rp.dp, 3))
    phi = np.zeros(NL + 1); phi[:len(rp.phi)] = rp.phi
    put("rp_phi", phi)
    put("rp_bp1", _rows(rp.bp1, 3)); put("rp_bp2", _rows(rp.bp2, 3)); put("rp_br1", _rows(rp.br1, 3))
    put("rp_br2", _rows(rp.br2, 3)); put("rp_erb", _rows(rp.e_r_bound, 6))
    put("rp_a", _rows(rp.a_sets, 45)); put("rp_b", _rows(rp.b_sets, 15))
    put("rp_pd", rp.pd); put("rp_r_taud", rp.r_taud); put("rp_dpd", rp.dpd); put("rp_dpdn", rp.dpd_normed)
    put("rp_phi_switch", rp.phi_switch)
    return S


def state_view(lay, S):
    """dict of named views into one state vector (or a [R, size] array of them)."""
    S = np.asarray(S)
    return {k: S[..., oc[0]:oc[0] + oc[1]] for k, oc in lay.items() if k != "_size"}
