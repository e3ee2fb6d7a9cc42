"""Device-resident closed loop (BASELINE.json configs[4]): R rollouts advanced in lock step on one
MI355X, every step = prepare kernel -> batched NLP solve -> finish kernel, with the per-rollout state
(robot state, ReferencePath window, split indices, warm start) living in HBM.  Only the plan-time
construction runs on the host: `BoundMPC.__init__/update` + `ReferencePath.__init__`
(/root/reference/bound_planner/BoundMPC/BoundMPC.py:28-336, ReferencePath/ReferencePath.py:12-157) build
the Python objects, `pack_state` serialises them into the state vector of
boundplanner_amd/csrc/bmpc_loop.hpp, and the HIP kernels carry on from there
(BoundMPC.step / compute_return_data / MPCNode.step, BoundMPC.py:388-1040, MPCNode.py:106-160).

No CPU fallback: `DeviceLoop` needs libboundmpc_hip.so and a GPU.  Scene obstacles (polytopes with their vertices, the
inputs of ConvexSetFinder.find_set_collision_avoidance, ConvexSetFinder.py:309-375) are shared by all rollouts of a loop:
`set_obstacles`; the per-step collision sets are then computed on the device as well.
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


def pack_obstacles(obs_sets, obs_points_sets):
    """[A, b] polytopes + vertex arrays -> the flat arrays of bmpc_loop_set_obstacles (15 rows / 32 vertices each)."""
    n = len(obs_sets)
    if n > 16:
        raise ValueError("at most 16 scene obstacles")
    A = np.zeros((n, 15, 3)); b = np.zeros((n, 15)); V = np.zeros((n, 32, 3))
    nrows = np.zeros(n, np.int32); nv = np.zeros(n, np.int32)
    for i, ((a_i, b_i), v_i) in enumerate(zip(obs_sets, obs_points_sets)):
        a_i, b_i, v_i = np.asarray(a_i, float), np.asarray(b_i, float), np.asarray(v_i, float)
        if a_i.shape[0] > 15 or v_i.shape[0] > 32:
            raise ValueError("obstacle with more than 15 faces or 32 vertices")
        A[i, :a_i.shape[0]] = a_i; b[i, :a_i.shape[0]] = b_i; V[i, :v_i.shape[0]] = v_i
        nrows[i], nv[i] = a_i.shape[0], v_i.shape[0]
    return A, b, nrows, V, nv


def state_view(lay, S):
    """dict of named views into one state vector (or a [R, size] array of them)."""
    S = np.asarray(S)
    return {k: S[..., oc[0]:oc[0] + oc[1]] for k, oc in lay.items() if k != "_size"}


class DeviceLoop:
    """R closed-loop rollouts on one GPU.  `backend` is a HipBoundMPC (its handle's solver is borrowed).

    Typical use (tools/closed_loop_device.py):
        loop = DeviceLoop(backend, R)
        loop.set_rollout(r, mpc, q, ...)      # host objects after BoundMPC.__init__ / update
        loop.upload()
        log = loop.run(nsteps)                # [nsteps, R, log_doubles]
    """

    LOG = {"iters": 0, "status": 1, "viol": 2, "error_count": 3, "dead": 4, "phi": 5, "phi_max": 6, "split1": 7,
           "sector": 8, "switch": 9, "p_lie": slice(10, 16), "q": slice(16, 23)}

    def __init__(self, backend, R):
        from .solver import load_library
        self.lib = load_library()
        self.be, self.R, self.N = backend, int(R), backend.N
        self.lay = read_layout(self.lib.bmpc_loop_field, self.lib.bmpc_loop_state_doubles)
        self.logw = self.lib.bmpc_loop_log_doubles()
        self._l = ctypes.c_void_p()
        rc = self.lib.bmpc_loop_create(backend._h, self.R, ctypes.byref(self._l))
        if rc != 0:
            msg = self.lib.bmpc_loop_last_error(self._l).decode() if self._l else "invalid arguments"
            raise RuntimeError(f"bmpc_loop_create failed ({rc}): {msg} -- the device loop has no CPU fallback")
        self.state = np.zeros((self.R, self.lay["_size"]))
        self.prev = np.zeros((self.R, backend.n_w))
        self.ms_total = self.ms_solve = 0.0

    def close(self):
        if getattr(self, "_l", None):
            self.lib.bmpc_loop_destroy(self._l)
            self._l = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.bmpc_loop_last_error(self._l).decode()}")

    def _P(self, a):
        return a.ctypes.data_as(_dp) if a is not None else None

    # ---- host <-> device state
    def set_rollout(self, r, mpc, q, dq, ddq, jerk, qf, v, p_lie):
        self.state[r] = pack_state(self.lay, mpc, q, dq, ddq, jerk, qf, v, p_lie)
        if mpc.prev_solution is not None:
            self.prev[r] = mpc.prev_solution

    def set_obstacles(self, obs_sets, obs_points_sets):
        """Scene obstacles of ALL rollouts (what BoundMPC.set_obstacle_sets takes per instance on the host)."""
        A, b, nrows, V, nv = pack_obstacles(obs_sets, obs_points_sets)
        self._chk(self.lib.bmpc_loop_set_obstacles(self._l, len(obs_sets), self._P(A), self._P(b), nrows.ctypes.data_as(_ip),
                                                   self._P(V), nv.ctypes.data_as(_ip)), "bmpc_loop_set_obstacles")

    def upload(self, first=0, count=None):
        count = self.R - first if count is None else count
        self._chk(self.lib.bmpc_loop_upload(self._l, first, count, self._P(self.state[first:first + count]),
                                            self._P(self.prev[first:first + count])), "bmpc_loop_upload")

    def download(self):
        self._chk(self.lib.bmpc_loop_download(self._l, 0, self.R, self._P(self.state), self._P(self.prev)), "bmpc_loop_download")
        return state_view(self.lay, self.state)

    def replan(self, r, mpc, p_via, r_via, bp1, br1, e_r_bound, a_sets, b_sets):
        """MPCNode.update_reference for rollout r (MPCNode.py:82-104): the host BoundMPC object `mpc` of this
        rollout is updated with the new via path from the rollout's CURRENT device state and re-serialised; warm
        start, slacks0 and error count carry over (a15).  Call download() first, upload() afterwards."""
        V = state_view(self.lay, self.state[r])
        mpc.slacks0 = V["slacks0"].copy(); mpc.error_count = int(V["error_count"][0])
        mpc.prev_solution = self.prev[r].copy() if V["has_prev"][0] != 0 else None
        q = V["q"].copy()
        from .params import Params
        mpc.update(p_via, r_via, bp1, br1, e_r_bound, a_sets, b_sets, [], V["v"].copy(), p0=V["p_lie"].copy(),
                   params=Params(n=self.N, dt=self.be.opts.dt, build=False, weights=V["weights"].copy(), nr_segs=mpc.nr_segs))
        self.state[r] = pack_state(self.lay, mpc, q, V["dq"].copy(), V["ddq"].copy(), V["jerk"].copy(), q, V["v"].copy(),
                                   V["p_lie"].copy())

    def horizon_points(self, r):
        """End-effector positions of rollout r's last accepted solution, stage by stage: the `p_horizon` argument of
        BoundPlanner.plan_convex_set_path(replanning=True) (BoundPlanner.py:231-276).  Call download() first."""
        N = self.N
        x = self.prev[r]
        return [x[28 * N + np.arange(3) * N + k].copy() for k in range(N)]

    def plan_and_replan(self, r, mpc, planner, goal_p, goal_r, replanning=True, new_obs=False):
        """Plan from rollout r's CURRENT pose to the goal pose with `planner` (boundplanner_amd.bound_planner.BoundPlanner) --
        when replanning, along the MPC horizon of its last solution -- and hand the plan to replan().  Returns the plan
        (p_via, r_via, bp1_list, sets_via).  Call download() first, upload() afterwards."""
        from scipy.spatial.transform import Rotation as Rot
        V = state_view(self.lay, self.state[r])
        p_lie = V["p_lie"].copy()
        has_prev = V["has_prev"][0] != 0
        kw = dict(replanning=True, p_horizon=self.horizon_points(r), new_obs=new_obs) if (replanning and has_prev) else {}
        p_via, r_via, bp1, sets = planner.plan_convex_set_path(p_lie[:3], np.asarray(goal_p, float), Rot.from_rotvec(p_lie[3:]).as_matrix(),
                                                               np.asarray(goal_r, float), **kw)
        n = len(bp1)
        erb = [np.array([90, 90, 90, -90, -90, -90]) * np.pi / 180 for _ in range(n)]        # boundplanner_with_mpc_example.py:132-133
        # (copies: ReferencePath appends its padded segments to the lists it is given, ReferencePath.py:43-46 -- Q15)
        self.replan(r, mpc, list(p_via), list(r_via), list(bp1), [np.array([0, 0, 1.0])] * n, erb, [s_[0] for s_ in sets], [s_[1] for s_ in sets])
        return p_via, r_via, bp1, sets

    # ---- MPCData records (boundmpcmsg/msg/MPCData.msg:1-64)
    def set_record(self, rollouts):
        """Rollouts whose steps are recorded by run() / finish() from now on (empty: none)."""
        sel = np.ascontiguousarray(list(rollouts), np.int32)
        self._chk(self.lib.bmpc_loop_set_record(self._l, len(sel), sel.ctypes.data_as(_ip) if len(sel) else None), "bmpc_loop_set_record")
        self._rec_sel = [int(r) for r in sel]

    def records(self):
        """Raw records [steps][len(rollouts)][width] of the last run() / finish(); mpc_data.from_device_record decodes one."""
        n = len(getattr(self, "_rec_sel", []))
        w = self.lib.bmpc_loop_record_doubles(self.N)
        steps = ctypes.c_int(0)
        self._chk(self.lib.bmpc_loop_records(self._l, None, 0, ctypes.byref(steps)), "bmpc_loop_records")      # how many steps were recorded
        out = np.zeros((max(steps.value, 1), max(n, 1), w))
        self._chk(self.lib.bmpc_loop_records(self._l, self._P(out), out.shape[0], ctypes.byref(steps)), "bmpc_loop_records")
        return out[:steps.value, :n]

    # ---- stepping
    def run(self, nsteps, log=True):
        out = np.zeros((nsteps, self.R, self.logw)) if log else None
        mt, ms = ctypes.c_float(), ctypes.c_float()
        self._chk(self.lib.bmpc_loop_run(self._l, int(nsteps), self._P(out), ctypes.byref(mt), ctypes.byref(ms)), "bmpc_loop_run")
        self.ms_total, self.ms_solve = mt.value, ms.value
        return out

    def run_async(self, nsteps, log=True):
        """The same `nsteps` MPC steps of every rollout without lock step: a rollout starts its next step as soon as its own
        solve has retired (bmpc_loop_run_async).  Same log as run(), bitwise."""
        out = np.zeros((nsteps, self.R, self.logw)) if log else None
        mt = ctypes.c_float()
        self._chk(self.lib.bmpc_loop_run_async(self._l, int(nsteps), self._P(out), ctypes.byref(mt)), "bmpc_loop_run_async")
        self.ms_total, self.ms_solve = mt.value, mt.value
        return out

    def prepare(self):
        self._chk(self.lib.bmpc_loop_prepare(self._l), "bmpc_loop_prepare")

    def solve(self):
        self._chk(self.lib.bmpc_loop_solve(self._l), "bmpc_loop_solve")

    def finish(self):
        out = np.zeros((self.R, self.logw))
        self._chk(self.lib.bmpc_loop_finish(self._l, self._P(out)), "bmpc_loop_finish")
        return out

    def problem(self):
        nw = self.be.n_w
        x0, lbx, ubx, p = np.zeros((self.R, nw)), np.zeros((self.R, nw)), np.zeros((self.R, nw)), np.zeros((self.R, 875))
        self._chk(self.lib.bmpc_loop_problem(self._l, self._P(x0), self._P(lbx), self._P(ubx), self._P(p)), "bmpc_loop_problem")
        return x0, lbx, ubx, p

    def solution(self):
        x = np.zeros((self.R, self.be.n_w)); it = np.zeros(self.R, np.int32); st = np.zeros(self.R, np.int32); viol = np.zeros(self.R)
        self._chk(self.lib.bmpc_loop_solution(self._l, self._P(x), it.ctypes.data_as(_ip), st.ctypes.data_as(_ip), self._P(viol)),
                  "bmpc_loop_solution")
        return dict(x=x, iters=it, status=st, viol=viol)

    def set_solution(self, x, iters, status, viol):
        x = np.ascontiguousarray(x, float); viol = np.ascontiguousarray(viol, float)
        it = np.ascontiguousarray(iters, np.int32); st = np.ascontiguousarray(status, np.int32)
        self._chk(self.lib.bmpc_loop_set_solution(self._l, self._P(x), it.ctypes.data_as(_ip), st.ctypes.data_as(_ip), self._P(viol)),
                  "bmpc_loop_set_solution")
