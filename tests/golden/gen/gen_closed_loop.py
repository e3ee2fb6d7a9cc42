"""Generate tests/golden/closed_loop.npz: a closed-loop trace of the REFERENCE's own host logic.

Runs, in the build container only, the reference's unmodified BoundMPC.update / BoundMPC.step
(prep + compute_return_data, /root/reference/bound_planner/BoundMPC/BoundMPC.py:271-1040), its
ReferencePath and its integrate_joint (utils/util_functions.py:55-65), with the three third-party
boundaries replaced by equivalents that exist here:
  * the NLP solver call (CasADi/IPOPT)  -> the CPU oracle (oracle/), through a CasADi-function-like shim
  * Pinocchio numeric kinematics          -> the oracle's FK (pinned against the reference's .ca tapes)
  * ConvexSetFinder (no obstacles)        -> init_halfspaces_point boxes (what it returns without obstacles)
The trace (inputs, the solver-call arguments the reference produced, the solution returned, the
post-processed outputs and the carried state after every step) pins rows a9, a11, a12, a13, a15 of
SURVEY.md section 8.  The fixture is data only; no reference source is copied.

    python tests/golden/gen/gen_closed_loop.py
"""
import os
import sys
import types
from collections import defaultdict

import numpy as np
from scipy.spatial.transform import Rotation as R

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
OUT = os.path.abspath(os.path.join(HERE, ".."))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(1, REF)
sys.path.insert(2, os.path.join(ROOT, "tests"))
sys.path.insert(3, ROOT)
os.chdir(REF)

import oracle_lib as O  # noqa: E402
from boundplanner_amd.collision_sets import find_set_collision_avoidance  # noqa: E402
from bound_planner.BoundMPC.BoundMPC import BoundMPC as RefMPC  # noqa: E402
from bound_planner.utils import get_default_params, integrate_joint  # noqa: E402
from bound_planner.utils.util_functions import Params  # noqa: E402

class FakeRobot:
    """Numeric kinematics with the reference RobotModel's method names."""
    col_joint_sizes = [0.09, 0.12, 0.09, 0.10, 0.07, 0.09, 0.075]

    def _k(self, q, dq=None):
        return {k: v[0] for k, v in O.fk_batch(np.asarray(q, float)[None], None if dq is None else np.asarray(dq, float)[None]).items()}

    def fk_pos_col(self, q, i):
        return self._k(q)["col_pts"][i]

    def jacobian_fk(self, q):
        return self._k(q)["jac"]

    def fk(self, q):
        o = self._k(q)
        return np.concatenate((o["ee_pos"], R.from_matrix(o["ee_rot"]).as_rotvec()))

    def forward_kinematics(self, q, dq):
        return self.fk(q), self.jacobian_fk(q), np.zeros((6, 7))

    def velocity_ee(self, q, dq):
        return (self.jacobian_fk(q) @ dq)[:3]

    def omega_ee(self, q, dq):
        return (self.jacobian_fk(q) @ dq)[3:]


class DM:
    def __init__(self, a):
        self.a = np.asarray(a, float)

    def full(self):
        return self.a.reshape(-1, 1)

    def __array__(self, dtype=None):
        return self.a


class OracleSolver:
    """CasADi-function-like wrapper of the CPU oracle; records every call."""

    def __init__(self, n, fail_calls=()):
        self.n = n
        self.calls = []
        self._stats = {}
        self.fail_calls = set(fail_calls)      # indices of calls that come back as a failed, infeasible solve

    def __call__(self, x0, lbx, ubx, lbg, ubg, p):
        x0, lbx, ubx, p = (np.asarray(a, float).reshape(-1) for a in (x0, lbx, ubx, p))
        fail = len(self.calls) in self.fail_calls
        r = O.solve(self.n, x0, lbx, ubx, p, max_iter=2 if fail else 100)
        if fail:
            # an interrupted solve whose iterate violates a dynamics row by 1e-2 (what Maximum_Iterations_Exceeded /
            # Infeasible_Problem_Detected returns look like to BoundMPC.py:613-617)
            r["status"] = 1
            r["g"] = r["g"].copy(); r["g"][3] += 1e-2
        lb, ub = O.gbounds(self.n)
        viol = -np.sum(r["g"][r["g"] < lb - 1e-6]) + np.sum(r["g"][r["g"] > ub + 1e-6])      # BoundMPC.py:613-615
        r["viol"] = float(viol)
        self._stats = {"iter_count": r["iters"], "success": r["status"] == 0,
                       "return_status": "Solve_Succeeded" if r["status"] == 0 else "Maximum_Iterations_Exceeded"}
        self.calls.append(dict(x0=x0.copy(), lbx=lbx.copy(), ubx=ubx.copy(), p=p.copy(), x=r["x"].copy(), g=r["g"].copy(),
                               iters=r["iters"], status=r["status"], viol=r["viol"]))
        return {"x": DM(r["x"]), "g": DM(r["g"]), "lam_g": DM(r["lam_g"]), "lam_x": DM(r["lam_x"]), "f": DM([r["f"]])}

    def stats(self):
        return self._stats


def reference_finder(boxes):
    """The reference's own ConvexSetFinder on box obstacles, its qpOASES slot filled by an exact QP solver (gen_colsets.py)."""
    from bound_planner.BoundPlanner.ConvexSetFinder import ConvexSetFinder
    from bound_planner.utils import normalize_set_size
    from boundplanner_amd import scenes
    from gen_colsets import ExactQP
    sets, pts = scenes.boxes_to_sets(boxes)
    obs_sets = normalize_set_size([[a.copy(), b.copy()] for a, b in sets], 15)
    f = ConvexSetFinder(obs_sets, pts, np.array([1.0, 0.38, 1.0]), np.array([-0.14, -1.0, 0.0]))
    f.projl_solver = ExactQP()
    return f


def make_ref_mpc(pos_points, rot_points, bp1, br1, e_r_bound, a_sets, b_sets, p0, params, solver, robot, finder=None):
    """A reference BoundMPC object with the state its __init__ sets up (BoundMPC.py:28-265), minus the
    CasADi/IPOPT solver build, the Pinocchio model and the planner's cvxpy/cdd machinery."""
    from bound_planner.ReferencePath import ReferencePath
    m = RefMPC.__new__(RefMPC)
    m.N = params.n
    m.robot_model = robot
    m.ref_data = defaultdict(list)
    m.err_data = defaultdict(list)
    for _ in range(m.N):
        for k in ("p", "dp", "ddp", "dp_normed", "dp_normedn", "bp1", "bp2", "br1", "br2", "br1_next", "br2_next", "v1",
                  "v2", "v3", "v1_next", "v2_next", "v3_next", "p_r_omega0", "r_bound_lower", "r_bound_upper",
                  "r_bound_lower_next", "r_bound_upper_next"):
            m.ref_data[k].append([])
        for k in ("e_p", "de_p", "e_p_par", "e_p_orth", "de_p_par", "de_p_orth", "e_r", "de_r", "e_r_par", "e_r_orth1",
                  "e_r_orth2", "e_r_parn", "e_r_orth1n", "e_r_orth2n"):
            m.err_data[k].append([])
    m.updated = False
    m.nr_slacks = 6 + m.N * 4
    m.slacks0 = np.zeros(6)
    m.obstacles = []
    m.p0 = p0
    m.qd = np.zeros(7)
    m.error_count = 0
    m.dt = params.dt
    m.nr_segs = params.nr_segs
    m.ref_path = ReferencePath(pos_points, rot_points, bp1, br1, e_r_bound, a_sets, b_sets, m.nr_segs)
    m.split_idxs = [0] + [m.N] * m.nr_segs
    m.switch = False
    S = m.nr_segs
    m.dtau_init, m.dtau_init_par = np.empty((3, S)), np.empty((3, S))
    m.dtau_init_orth1, m.dtau_init_orth2 = np.empty((3, S)), np.empty((3, S))
    m.phi_max = np.array([m.ref_path.phi_max])
    m.weights = np.array(params.weights)
    m.dp_ref = None
    m.pr_ref = p0[3:]
    m.iw_ref = np.zeros(3)
    m.phi_current, m.dphi_current = np.array([0.0]), np.array([0.0])
    m.nr_joints = 7
    m.nr_u = 8
    m.nr_x = 40
    qu = np.array([2.9670597283903604, 2.0943951023931953, 2.9670597283903604, 2.0943951023931953,
                   2.9670597283903604, 2.0943951023931953, 3.0543261909900763])
    m.q_ub, m.q_lb = np.repeat(qu, m.N), np.repeat(-qu, m.N)
    m.dq_ub, m.dq_lb = np.repeat(10.0 * np.ones(7), m.N), np.repeat(-10.0 * np.ones(7), m.N)
    m.ddq_ub = 5.0 * np.ones(m.N * 7); m.ddq_lb = -m.ddq_ub
    m.u_ub = 35.0 * np.ones(m.N * 7); m.u_lb = -35.0 * np.ones(m.N * 7)
    m.p_ub = np.inf * np.ones(m.N * 6); m.p_lb = -m.p_ub
    m.v_ub = np.inf * np.ones(m.N * 6); m.v_lb = -m.v_ub
    m.prev_solution = None
    m.lam_g0 = m.lam_x0 = 0
    m.solver = solver
    lb, ub = O.gbounds(m.N)
    m.lbg, m.ubg = lb, ub
    if finder is None:
        finder = types.SimpleNamespace(find_set_collision_avoidance=lambda pl, pf, limit_space=True, e_max=0.7:
                                       find_set_collision_avoidance([], [], np.asarray(pl), np.asarray(pf), e_max=e_max))
    m.planner = types.SimpleNamespace(set_finder=finder, add_obstacle_reps=lambda *a, **k: None)
    return m


def box_set(lo, hi):
    a = np.vstack((np.eye(3), -np.eye(3), np.zeros((9, 3))))
    b = np.concatenate((np.asarray(hi, float), -np.asarray(lo, float), 10.0 * np.ones(9)))
    return a, b


ERB = np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180


def scenario(name, p0):
    """Via paths of the committed traces (positions relative to the start pose p0, rotations relative to its orientation)."""
    R0 = R.from_rotvec(p0[3:]).as_matrix()
    rot = lambda *e: R0 @ R.from_euler("xyz", list(e), degrees=True).as_matrix()
    big = [box_set([-0.3, -1.0, 0.05], [0.9, 0.4, 1.0])]
    if name in ("base", "fail"):
        # three via points, rotation about two axes, two large box sets (phi_max = 0.57 < 1: the w_phi rescale of Q10 is live)
        dp = [[0, 0, 0], [0.05, -0.25, 0.10], [0.15, -0.45, -0.05]]
        rots = [R0, rot(20, 0, 10), rot(20, 25, 10)]
        sets = [box_set([-0.2, -0.6, 0.2], [0.8, 0.3, 0.9]), box_set([-0.1, -0.9, 0.1], [0.9, 0.0, 0.8])]
        return dict(N=10, dp=dp, rots=rots, sets=sets, steps=70, fail_calls=(9, 10, 21) if name == "fail" else ())
    if name == "n15":
        # the reference's default horizon, five via points: three set switches, the 4-segment window of ReferencePath
        # slides, phi_max = 1.18 > 1 (no w_phi rescale, Q10); the first 60 steps are recorded
        dp = [[0, 0, 0], [0.05, -0.25, 0.10], [0.15, -0.45, -0.05], [0.30, -0.45, 0.15], [0.30, 0.05, 0.30]]
        rots = [R0, rot(20, 0, 10), rot(20, 25, 10), rot(0, 25, -10), rot(0, 0, 0)]
        return dict(N=15, dp=dp, rots=rots, sets=big * 4, steps=60, fail_calls=())
    if name == "patch":
        # the end-effector orientation passes through a half turn: its rotation vector (|.| = 2.09 at the start) flips
        # sign on the way, and the warm start's integrated omega is re-based (Q11, BoundMPC.py:423-428)
        dp = [[0, 0, 0], [0.05, -0.20, 0.05], [0.10, -0.40, 0.0]]
        rots = [R0, rot(0, 0, 55), rot(0, 0, 110)]
        return dict(N=10, dp=dp, rots=rots, sets=big * 2, steps=50, fail_calls=())
    if name == "scene":
        # BASELINE configs[0]: the reference's example scene (boundplanner_with_mpc_example.py:19-100: start configuration, goal
        # position, workspace, 12 box obstacles), its default horizon; the per-step collision sets come from the reference's
        # own ConvexSetFinder (a10 inside the loop).  The via path is hand-authored (the plan phase is not built).
        from boundplanner_amd import scenes
        boxes, _, goal_p, _ = scenes.example_scene()
        dp = [[0, 0, 0], [0.05, -0.25, 0.10], (goal_p - p0[:3]).tolist()]
        rots = [R0, rot(20, 0, 10), rot(20, 25, 10)]
        ws = [box_set([-0.14, -1.0, 0.0], [1.0, 0.38, 1.0])]
        return dict(N=15, dp=dp, rots=rots, sets=ws * 2, steps=45, fail_calls=(), boxes=boxes)
    raise KeyError(name)


def main(name):
    base = get_default_params()
    robot = FakeRobot()
    q0 = np.array([0, 0, 0, -np.pi / 2, 0, np.pi / 2, 0.0])
    p0 = robot.fk(q0)
    sc = scenario(name, p0)
    N = sc["N"]
    params = Params(n=N, dt=base.dt, build=False, weights=base.weights, nr_segs=base.nr_segs)
    solver = OracleSolver(N, sc["fail_calls"])
    finder = reference_finder(sc["boxes"]) if "boxes" in sc else None
    # start-up problem of MPCNode.reset (MPCNode.py:44-80) + the warm-up step of the example
    mpc = make_ref_mpc([p0[:3]] * 2, [R.from_rotvec(p0[3:]).as_matrix()] * 2, [np.array([1.0, 0, 0])],
                       [np.array([1.0, 0, 0])], [ERB.copy()], [np.zeros((15, 3))], [np.ones(15)], p0, params, solver, robot, finder)
    st = dict(q=q0.copy(), qf=q0.copy(), dq=np.zeros(7), ddq=np.zeros(7), jerk=np.zeros(7), p_lie=p0.copy(), v=np.zeros(6))
    trace = []

    def node_step():
        st["p_lie"], _, _ = robot.forward_kinematics(st["q"], st["dq"])
        rec = {"in_" + k: np.copy(v) for k, v in st.items()}
        patched = 0
        if mpc.prev_solution is not None:       # the condition of BoundMPC.py:423
            patched = int(np.linalg.norm(st["p_lie"][3:] - np.asarray(mpc.prev_solution)[31 * N:34 * N:N]) > 1.5)
        traj, ref_data, err_data, _, iters = mpc.step(st["q"], st["dq"], st["ddq"], st["p_lie"], st["v"], st["jerk"], st["qf"])
        new = integrate_joint(robot, traj["dddq"], st["q"], st["dq"], st["ddq"], mpc.dt)
        st["q"], st["dq"], st["ddq"], st["p_lie"], st["v"] = new[0], new[1], new[2], new[3], new[4]
        st["qf"] = traj["q"][:, -1]
        st["jerk"] = traj["dddq"][:, 1]
        c = solver.calls[-1]
        rec.update({"call_" + k: c[k] for k in ("x0", "lbx", "ubx", "p", "x", "g")})
        rec.update(iters=c["iters"], status=c["status"], viol=c["viol"], patched=patched)
        # failed steps shift the outputs by error_count columns (Q12): pad to the full width
        pad = lambda a: np.concatenate((np.asarray(a, float), np.full(np.asarray(a).shape[:-1] + (N - np.asarray(a).shape[-1],), np.nan)), axis=-1)
        rec.update({"traj_" + k: pad(traj[k]) for k in ("p", "v", "q", "dq", "ddq", "dddq", "phi", "dphi")})
        rec.update(traj_cols=np.asarray(traj["dddq"]).shape[-1])
        rec.update(ref_p1=np.array(ref_data["p"][1]), ref_p0=np.array(ref_data["p"][0]),
                   err_e_p1=np.array(err_data["e_p"][1]), err_e_r1=np.array(err_data["e_r"][1]))
        rec.update(split_idxs=np.array(mpc.split_idxs), switch=int(mpc.switch), pr_ref=np.copy(mpc.pr_ref),
                   iw_ref=np.copy(mpc.iw_ref), phi_current=mpc.phi_current.copy(), dphi_current=mpc.dphi_current.copy(),
                   phi_max=mpc.phi_max.copy(), slacks0=mpc.slacks0.copy(), error_count=mpc.error_count,
                   sector=mpc.ref_path.sector, rp_pd=mpc.ref_path.pd.copy(), rp_phi_switch=mpc.ref_path.phi_switch.copy())
        rec.update({"out_" + k: np.copy(v) for k, v in st.items()})
        trace.append(rec)

    node_step()       # warm-up on the trivial path
    p_via = [p0[:3] + np.array(d, float) for d in sc["dp"]]
    r_via = [np.array(r) for r in sc["rots"]]
    ns = len(p_via) - 1
    bp1 = [np.array([0.0, 0, 1]) for _ in range(ns)]
    br1 = [np.array([0.0, 0, 1]) for _ in range(ns)]
    e_r_bound = [ERB.copy() for _ in range(ns)]
    a_sets, b_sets = [s[0] for s in sc["sets"]], [s[1] for s in sc["sets"]]
    via = dict(p_via=np.array(p_via), r_via=np.array(r_via), bp1=np.array(bp1), br1=np.array(br1),
               e_r_bound=np.array(e_r_bound), a_sets=np.array(a_sets), b_sets=np.array(b_sets))
    # MPCNode.update_reference (MPCNode.py:82-104)
    mpc.update([p.copy() for p in p_via], [r.copy() for r in r_via], bp1, br1, e_r_bound, a_sets, b_sets, [], st["v"],
               p0=np.copy(st["p_lie"]), params=params)
    st["qf"] = st["q"].copy()
    n_update = len(trace)
    for _ in range(sc["steps"]):
        if mpc.phi_current[0] >= mpc.phi_max[0] - 0.001:
            break
        node_step()
    keys = sorted(trace[0].keys())
    out = {k: np.array([t[k] for t in trace]) for k in keys}
    out["n_update"] = n_update
    out["N"] = N
    out["weights"] = params.weights
    out.update({"via_" + k: v for k, v in via.items()})
    if "boxes" in sc:
        out["boxes"] = sc["boxes"]
    fname = "closed_loop.npz" if name == "base" else f"closed_loop_{name}.npz"
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "written:", len(trace), "steps; final phi", mpc.phi_current, "/", mpc.phi_max,
          "iters", [int(t["iters"]) for t in trace], "switches at", [i for i, t in enumerate(trace) if t["switch"]],
          "final sector", int(trace[-1]["sector"]), "failed steps", [i for i, t in enumerate(trace) if t["error_count"]],
          "omega re-based at", [i for i, t in enumerate(trace) if t["patched"]])


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    for nm in (sys.argv[1:] or ["base", "n15", "fail", "patch", "scene"]):
        main(nm)
