"""The device-resident closed-loop step logic (boundplanner_amd/csrc/bmpc_loop.hpp) against the REFERENCE's own
closed-loop trace: the CPU build of the identical source (tests/emu/emu_loop.cpp) replays
tests/golden/closed_loop.npz -- every solver argument it prepares (start vector, bounds, the 875 parameters) and
all state it carries (split indices, switch with via-point adaptation, rotation reference, path parameter,
joint state) must equal what the reference's BoundMPC.step / compute_return_data / integrate_joint produced."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as R

import emu_loop_lib as E
import oracle_lib as O
from boundplanner_amd.device_loop import pack_state, state_view
from boundplanner_amd.mpc_node import MPCNode
from boundplanner_amd.params import Params, get_default_params
from boundplanner_amd.robot_model import RobotModel
from test_closed_loop import ReplaySolver


def test_so3_helpers_match_scipy():
    rng = np.random.default_rng(0)
    for i in range(400):
        v = rng.normal(size=3) * rng.choice([1e-5, 1e-3, 0.3, 1.0, 2.5])
        if i % 7 == 0:
            v = v / np.linalg.norm(v) * (np.pi - 1e-3 * rng.uniform())      # near the half turn
        M = R.from_rotvec(rng.normal(size=3) * rng.choice([1e-4, 0.5, 2.0])).as_matrix()
        Rv, v2, e = E.so3(v, M)
        assert np.abs(Rv - R.from_rotvec(v).as_matrix()).max() < 1e-14
        assert np.abs(v2 - R.from_matrix(M).as_rotvec()).max() < 1e-12
        assert np.abs(e - R.from_matrix(M).as_euler("zyx")).max() < 1e-12


def test_replay_of_the_reference_trace(golden_dir):
    g = np.load(os.path.join(golden_dir, "closed_loop.npz"))
    N = int(g["N"])
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    lay = E.layout()
    shadow = MPCNode(g["in_q"][0], RobotModel(O.fk_batch), lambda n, dt: ReplaySolver(g, tol=1e-9), params=params)
    n_w = 44 * N + 6
    prev = np.zeros(n_w)
    pack = lambda: pack_state(lay, shadow.mpc, shadow.q, shadow.dq, shadow.ddq, shadow.jerk, shadow.qf, shadow.v, shadow.p_lie)
    S = pack()
    n_steps, n_update = g["in_q"].shape[0], int(g["n_update"])
    big = lambda a: np.nan_to_num(np.asarray(a, float), posinf=1e20, neginf=-1e20)
    worst = {}
    for k in range(n_steps):
        if k == n_update:
            # new plan: host-side BoundMPC.update / ReferencePath construction, re-serialised; the carried
            # warm start, slacks0 and error count are NOT reset (a15) and stay what the device loop holds
            shadow.update_reference([p.copy() for p in g["via_p_via"]], [r.copy() for r in g["via_r_via"]],
                                    [b.copy() for b in g["via_bp1"]], [b.copy() for b in g["via_br1"]],
                                    [e.copy() for e in g["via_e_r_bound"]], [a.copy() for a in g["via_a_sets"]],
                                    [b.copy() for b in g["via_b_sets"]], [])
            keep = {f: state_view(lay, S)[f].copy() for f in ("slacks0", "error_count", "has_prev", "q", "dq", "ddq", "jerk", "v", "p_lie")}
            S = pack()
            for f, val in keep.items():
                assert np.abs(state_view(lay, S)[f] - val).max() < 1e-9, f
                if f != "has_prev":
                    state_view(lay, S)[f][:] = val
            state_view(lay, S)["has_prev"][:] = keep["has_prev"]
        x0, lbx, ubx, p = E.prepare(N, S, prev)
        for name, mine in (("x0", x0), ("lbx", lbx), ("ubx", ubx), ("p", p)):
            d = np.abs(mine - big(g["call_" + name][k])).max()
            worst[name] = max(worst.get(name, 0.0), d)
            assert d < 1e-9, (k, name, d, np.argmax(np.abs(mine - big(g["call_" + name][k]))))
        E.finish(N, params.dt, S, g["call_x"][k], prev, int(g["status"][k]), float(g["viol"][k]), int(g["iters"][k]))
        shadow.step()                                   # keeps the host mirror in lock step for the re-plan
        V = state_view(lay, S)
        assert [int(s) for s in V["split"]] == list(g["split_idxs"][k]), k
        assert int(V["sw"][0]) == int(g["switch"][k]) and int(V["error_count"][0]) == int(g["error_count"][k]), k
        assert int(V["rp_sector"][0]) == int(g["sector"][k]), k
        for f, key in (("pr_ref", "pr_ref"), ("iw_ref", "iw_ref"), ("phi_current", "phi_current"), ("dphi_current", "dphi_current"),
                       ("phi_max", "phi_max"), ("slacks0", "slacks0"), ("rp_pd", "rp_pd"), ("rp_phi_switch", "rp_phi_switch"),
                       ("q", "out_q"), ("dq", "out_dq"), ("ddq", "out_ddq"), ("jerk", "out_jerk"), ("v", "out_v"),
                       ("qf", "out_qf"), ("p_lie", "out_p_lie")):
            d = np.abs(V[f] - np.asarray(g[key][k]).reshape(-1)).max()
            worst[f] = max(worst.get(f, 0.0), d)
            assert d < 1e-9, (k, f, d)
    assert g["switch"].sum() >= 1 and g["sector"][-1] == 1      # the trace exercises a switch + via-point adaptation
    print("max deviation from the reference trace:", {k: float(f"{v:.2e}") for k, v in worst.items()})
