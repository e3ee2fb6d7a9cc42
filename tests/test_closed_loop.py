"""Rows a9, a11, a12, a13, a15 of SURVEY.md section 8: the host logic around the solver call.

tests/golden/closed_loop.npz is a 37-step closed-loop trace of the REFERENCE's own BoundMPC.update /
BoundMPC.step (prep + compute_return_data), ReferencePath and integrate_joint, produced by
tests/golden/gen/gen_closed_loop.py with the CPU oracle in the solver slot.  Replaying the recorded
solutions through this package's mirror (BoundMPC.prepare -> solver -> post.compute_return_data ->
MPCNode.step) must reproduce, step by step, the solver-call arguments the reference built (start
vector, bounds, the 875 parameters), the post-processed trajectories and all carried state
(split indices, segment switch with via-point adaptation, rotation reference, path parameter)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from boundplanner_amd.mpc_node import MPCNode
from boundplanner_amd.params import Params, get_default_params
from boundplanner_amd.robot_model import RobotModel


class _DM:
    def __init__(self, a):
        self.a = np.asarray(a, float)

    def full(self):
        return self.a.reshape(-1, 1)


class ReplaySolver:
    """Returns the recorded solution of step `k` after checking that the call arguments equal the
    ones the reference produced at that step."""

    def __init__(self, g, tol):
        self.g, self.k, self.tol, self.maxdiff = g, 0, tol, {}
        self._stats = {}

    def __call__(self, x0, lbx, ubx, p, lbg=None, ubg=None):
        k = self.k
        big = lambda a: np.nan_to_num(np.asarray(a, float), posinf=1e20, neginf=-1e20)
        for name, mine in (("x0", x0), ("lbx", lbx), ("ubx", ubx), ("p", p)):
            d = np.abs(big(mine) - big(self.g["call_" + name][k])).max()
            self.maxdiff[name] = max(self.maxdiff.get(name, 0.0), d)
            # with scene obstacles the collision-set rows of p come from a closest-pair search resolved to ~1e-7
            tol = max(self.tol, 2e-6) if (name == "p" and "boxes" in self.g.files) else self.tol
            assert d < tol, f"step {k}: solver argument {name} differs from the reference's by {d}"
        self._stats = {"iter_count": int(self.g["iters"][k]), "success": int(self.g["status"][k]) == 0,
                       "return_status": "replay", "g_viol": float(self.g["viol"][k])}
        self.k += 1
        x = self.g["call_x"][k]
        return {"x": _DM(x), "g": _DM(self.g["call_g"][k]), "lam_g": _DM(0 * self.g["call_g"][k]), "lam_x": _DM(0 * x),
                "f": _DM([0.0])}

    def stats(self):
        return dict(self._stats)


TRACES = ["closed_loop.npz",          # N=10, 3 via points, one set switch with via-point adaptation, ends at the path end
          "closed_loop_n15.npz",      # the reference's default horizon N=15, 5 via points: three switches, the 4-segment window of
                                      # ReferencePath slides, phi_max > 1 (no w_phi rescale, Q10); first 60 steps
          "closed_loop_fail.npz",     # three failed solves (two of them consecutive, across the set switch): fallback to the
                                      # previous solution, outputs shifted by error_count columns (Q12)
          "closed_loop_patch.npz",    # orientation through a half turn: the rotation vector flips, the warm start's integrated
                                      # omega is re-based (Q11)
          "closed_loop_scene.npz"]    # BASELINE configs[0]: the reference's example scene (start, goal, workspace, 12 box obstacles,
                                      # N=15); the per-step collision sets in the trace come from the reference's OWN
                                      # ConvexSetFinder (a10 inside the loop); runs to the path end


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "closed_loop.npz"))


@pytest.fixture(scope="module", params=TRACES)
def trace(request, golden_dir):
    return np.load(os.path.join(golden_dir, request.param))


def _fk(q, dq=None):
    return O.fk_batch(q, dq)


def test_closed_loop_replay_matches_reference(trace):
    g = trace
    N = int(g["N"])
    base = get_default_params()
    assert np.array_equal(base.weights, g["weights"])
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    solver = ReplaySolver(g, tol=1e-9)
    q0 = g["in_q"][0]
    node = MPCNode(q0, RobotModel(_fk), lambda n, dt: solver, params=params)
    if "boxes" in g.files:
        from boundplanner_amd import scenes
        node.mpc.set_obstacle_sets(*scenes.boxes_to_sets(g["boxes"]))
    n_steps, n_update = g["in_q"].shape[0], int(g["n_update"])
    for k in range(n_steps):
        if k == n_update:      # MPCNode.update_reference with the planned via path
            node.update_reference([p.copy() for p in g["via_p_via"]], [r.copy() for r in g["via_r_via"]],
                                  [b.copy() for b in g["via_bp1"]], [b.copy() for b in g["via_br1"]],
                                  [e.copy() for e in g["via_e_r_bound"]], [a.copy() for a in g["via_a_sets"]],
                                  [b.copy() for b in g["via_b_sets"]], [])
        for key in ("q", "dq", "ddq", "jerk", "v", "qf"):
            assert np.abs(getattr(node, key) - g["in_" + key][k]).max() < 1e-9, (k, key)
        traj = node.step()
        for key in ("p", "v", "q", "dq", "ddq", "dddq", "phi", "dphi"):
            mine = np.asarray(traj[key])
            ref = g["traj_" + key][k][..., :mine.shape[-1]]          # failed steps return fewer columns (Q12)
            assert np.isfinite(ref).all() and not np.isfinite(g["traj_" + key][k][..., mine.shape[-1]:]).any(), (k, key)
            assert np.abs(mine - ref).max() < 1e-9, (k, key)
        m = node.mpc
        assert list(m.split_idxs) == list(g["split_idxs"][k]), k
        assert int(m.switch) == int(g["switch"][k]) and m.error_count == int(g["error_count"][k]), k
        assert m.ref_path.sector == int(g["sector"][k]), k
        for mine, key in ((m.pr_ref, "pr_ref"), (m.iw_ref, "iw_ref"), (m.phi_current, "phi_current"),
                          (m.dphi_current, "dphi_current"), (m.phi_max, "phi_max"), (m.slacks0, "slacks0"),
                          (m.ref_path.pd, "rp_pd"), (m.ref_path.phi_switch, "rp_phi_switch"),
                          (node.ref_data["p"][1], "ref_p1"), (node.ref_data["p"][0], "ref_p0")):
            assert np.abs(np.asarray(mine) - g[key][k]).max() < 1e-9, (k, key)
        for key in ("q", "dq", "ddq", "jerk", "v", "qf", "p_lie"):
            assert np.abs(getattr(node, key) - g["out_" + key][k]).max() < 1e-9, (k, key)
    if True:
        # the MPCData-shaped record of the last step (boundmpcmsg/msg/MPCData.msg field set)
        from boundplanner_amd import mpc_data
        rec = mpc_data.from_node(node)
        assert set(rec) == set(mpc_data.FIELDS)
        assert len(rec["q"]) == traj["q"].shape[1] and np.array_equal(rec["q"][1], traj["q"][:, 1])
        assert np.array_equal(rec["phi"], traj["phi"]) and rec["sector"] == int(g["sector"][-1])
        assert rec["a_set_j3"].shape == (45,) and rec["b_set_elbow"].shape == (15,) and rec["iterations"] == int(g["iters"][-1])
        assert len(rec["e_p"]) >= len(rec["q"]) - 1 and np.abs(rec["p_ref"][1] - g["ref_p1"][-1]).max() < 1e-9
    # every scenario exercises a set switch with via-point adaptation
    assert g["switch"].sum() >= 1 and g["sector"][-1] >= 1
    assert (node.mpc.phi_current[0] >= node.mpc.phi_max[0] - 0.001) == (g["phi_current"][-1][0] >= g["phi_max"][-1][0] - 0.001)
    print("max |argument - reference argument| over the run:", solver.maxdiff)


@pytest.mark.gpu
def test_closed_loop_hip_tracks_reference_trace(golden):
    """The same closed loop with the HIP solver in the solver slot: same number of steps to the path
    end, same switching step, trajectories within the stated solver tolerance of the golden trace
    (the trace was produced with the CPU oracle as NLP solver)."""
    from boundplanner_amd.solver import HipBoundMPC, HipNlpSolver
    g = golden
    N = int(g["N"])
    base = get_default_params()
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    be = HipBoundMPC(N)
    node = MPCNode(g["in_q"][0], RobotModel(be.fk), lambda n, dt: HipNlpSolver(n, dt, backend=be), params=params)
    n_steps, n_update = g["in_q"].shape[0], int(g["n_update"])
    dmax = 0.0
    for k in range(n_steps):
        if k == n_update:
            node.update_reference([p.copy() for p in g["via_p_via"]], [r.copy() for r in g["via_r_via"]],
                                  [b.copy() for b in g["via_bp1"]], [b.copy() for b in g["via_br1"]],
                                  [e.copy() for e in g["via_e_r_bound"]], [a.copy() for a in g["via_a_sets"]],
                                  [b.copy() for b in g["via_b_sets"]], [])
        traj = node.step()
        dmax = max(dmax, np.abs(traj["p"] - g["traj_p"][k][:, :traj["p"].shape[1]]).max())
        assert list(node.mpc.split_idxs) == list(g["split_idxs"][k]), k
        assert node.mpc.error_count == 0
    assert dmax < 1e-3, dmax      # task-space trajectories over 37 closed-loop steps
    assert node.mpc.phi_current[0] >= node.mpc.phi_max[0] - 0.001
    assert abs(np.mean(node.iters) - g["iters"].mean()) < 1.0
