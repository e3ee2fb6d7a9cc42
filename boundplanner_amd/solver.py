"""ctypes binding of libboundmpc_hip.so (C ABI: include/boundmpc.h) and the solver object that is
call-compatible with the CasADi function used at BoundMPC.py:594-617.

There is NO CPU fallback: if the HIP library is missing or no MI355X is visible, constructing a
solver raises.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BMPC_LIB") or os.path.join(_HERE, "csrc", "libboundmpc_hip.so")   # BMPC_LIB: experiment builds
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


class BmpcOpts(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int), ("nr_segs", ctypes.c_int), ("dt", ctypes.c_double),
                ("tol", ctypes.c_double), ("max_iter", ctypes.c_int), ("device", ctypes.c_int),
                ("hess", ctypes.c_int), ("hess_switch", ctypes.c_double), ("mu_init", ctypes.c_double),
                ("kappa_mu", ctypes.c_double), ("theta_mu", ctypes.c_double), ("kappa_eps", ctypes.c_double),
                ("mu_floor_k", ctypes.c_double), ("inertia", ctypes.c_int), ("dw0", ctypes.c_double),
                ("inertia_err", ctypes.c_double), ("stall_n", ctypes.c_int), ("slack_reset", ctypes.c_int), ("ls_alpha_mem", ctypes.c_double), ("gn_backoff", ctypes.c_int),
                ("trial_repeats", ctypes.c_int), ("watchdog_ms", ctypes.c_int), ("max_batch", ctypes.c_int), ("pool_slots", ctypes.c_int)]


EXPORTS = ["bmpc_default_opts", "bmpc_create", "bmpc_destroy", "bmpc_last_error", "bmpc_dims",
           "bmpc_gbounds", "bmpc_solve", "bmpc_solve_dev", "bmpc_solve_dev_async", "bmpc_multipliers_dev", "bmpc_wait", "bmpc_active", "bmpc_fk",
           "bmpc_last_kernel_ms", "bmpc_get_opts", "bmpc_stream", "bmpc_robot_iiwa14", "bmpc_robot_gen3", "bmpc_set_robot", "bmpc_get_robot",
           "bmpc_debug_phase_cycles", "bmpc_debug_spin", "bmpc_debug_inst_state", "bmpc_debug_time_ric", "bmpc_debug_ric_stats", "bmpc_debug_ric_stats_full", "bmpc_debug_lane_stats",
           "bmpc_loop_state_doubles", "bmpc_loop_log_doubles", "bmpc_loop_field", "bmpc_loop_create", "bmpc_loop_destroy",
           "bmpc_loop_last_error", "bmpc_loop_record_doubles", "bmpc_loop_set_record", "bmpc_loop_records", "bmpc_loop_set_obstacles", "bmpc_loop_upload", "bmpc_loop_download", "bmpc_loop_run", "bmpc_loop_run_async", "bmpc_loop_prepare",
           "bmpc_loop_solve", "bmpc_loop_finish", "bmpc_loop_problem", "bmpc_loop_solution", "bmpc_loop_set_solution"]

_lib = None


def load_library():
    """Load libboundmpc_hip.so (raises if it was not built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        # PyTorch-ROCm wheels bundle their own HIP runtime.  If this library (linked against the system ROCm) is loaded
        # first, a later `import torch` in the same process finds "No HIP GPUs" (measured on ROCm 7.2 + torch 2.10+rocm7.0);
        # the other order works and both then share one runtime.  So when torch is installed it is imported first.
        if "torch" not in sys.modules and not os.environ.get("BMPC_NO_TORCH_PRELOAD"):
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        lib.bmpc_last_error.restype = ctypes.c_char_p
        lib.bmpc_last_error.argtypes = [ctypes.c_void_p]
        lib.bmpc_create.argtypes = [ctypes.POINTER(BmpcOpts), ctypes.POINTER(ctypes.c_void_p)]
        lib.bmpc_destroy.argtypes = [ctypes.c_void_p]
        lib.bmpc_dims.argtypes = [ctypes.c_void_p, _ip, _ip, _ip]
        lib.bmpc_gbounds.argtypes = [ctypes.c_void_p, _dp, _dp]
        lib.bmpc_solve.argtypes = [ctypes.c_void_p, ctypes.c_int] + [_dp] * 4 + [_dp] * 4 + [_dp, _ip, _ip, _dp]
        lib.bmpc_solve_dev.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 10 + [ctypes.c_void_p]
        lib.bmpc_solve_dev_async.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 10
        lib.bmpc_multipliers_dev.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.bmpc_wait.argtypes = [ctypes.c_void_p]
        lib.bmpc_set_robot.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.bmpc_get_robot.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.bmpc_stream.restype = ctypes.c_void_p
        lib.bmpc_stream.argtypes = [ctypes.c_void_p]
        lib.bmpc_active.argtypes = [ctypes.c_void_p]
        lib.bmpc_fk.argtypes = [ctypes.c_void_p, ctypes.c_int] + [_dp] * 7
        lib.bmpc_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        lib.bmpc_loop_last_error.restype = ctypes.c_char_p
        lib.bmpc_loop_last_error.argtypes = [ctypes.c_void_p]
        lib.bmpc_loop_field.argtypes = [ctypes.c_char_p, _ip, _ip]
        lib.bmpc_loop_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.bmpc_loop_destroy.argtypes = [ctypes.c_void_p]
        lib.bmpc_loop_set_obstacles.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, _dp, _ip, _dp, _ip]
        lib.bmpc_loop_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp]
        lib.bmpc_loop_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp]
        lib.bmpc_loop_run.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        lib.bmpc_loop_run_async.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_float)]
        lib.bmpc_loop_prepare.argtypes = [ctypes.c_void_p]
        lib.bmpc_loop_solve.argtypes = [ctypes.c_void_p]
        lib.bmpc_loop_finish.argtypes = [ctypes.c_void_p, _dp]
        lib.bmpc_loop_problem.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, _dp]
        lib.bmpc_loop_solution.argtypes = [ctypes.c_void_p, _dp, _ip, _ip, _dp]
        lib.bmpc_loop_set_solution.argtypes = [ctypes.c_void_p, _dp, _ip, _ip, _dp]
        lib.bmpc_debug_spin.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.bmpc_debug_inst_state.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp]
        lib.bmpc_debug_time_ric.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.bmpc_debug_ric_stats.argtypes = [ctypes.c_void_p, _dp]
        lib.bmpc_debug_lane_stats.argtypes = [ctypes.c_void_p, _dp]
        lib.bmpc_debug_ric_stats_full.argtypes = [ctypes.c_void_p, _dp]
        lib.bmpc_loop_set_record.argtypes = [ctypes.c_void_p, ctypes.c_int, _ip]
        lib.bmpc_loop_records.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _ip]
        lib.bmpc_loop_record_doubles.argtypes = [ctypes.c_int]
        _lib = lib
    return _lib


def _P(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class HipBoundMPC:
    """Owner of one C handle: batched solves + batched kinematics on one MI355X."""

    def __init__(self, N, dt=0.1, tol=1e-5, max_iter=100, device=0, robot=None, **kw):
        """robot: None (iiwa14), "iiwa14", "gen3" or a table of boundplanner_amd.robots"""
        lib = load_library()
        o = BmpcOpts()
        lib.bmpc_default_opts(ctypes.byref(o), N)
        o.dt, o.tol, o.max_iter, o.device = dt, tol, max_iter, device
        for k, v in kw.items():
            setattr(o, k, v)
        self._h = ctypes.c_void_p()
        rc = lib.bmpc_create(ctypes.byref(o), ctypes.byref(self._h))
        if rc != 0:
            msg = lib.bmpc_last_error(self._h).decode() if self._h else "invalid options"
            raise RuntimeError(f"bmpc_create failed ({rc}): {msg} -- the HIP path has no CPU fallback")
        self.lib, self.N, self.opts = lib, N, o
        nw, ng, npar = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.bmpc_dims(self._h, ctypes.byref(nw), ctypes.byref(ng), ctypes.byref(npar))
        self.n_w, self.n_g, self.n_p = nw.value, ng.value, npar.value
        self.lbg, self.ubg = np.zeros(self.n_g), np.zeros(self.n_g)
        lib.bmpc_gbounds(self._h, _P(self.lbg), _P(self.ubg))
        from . import robots
        self.robot = robots.IIWA14
        if robot is not None:
            self.set_robot(robot)

    def set_robot(self, robot):
        """Kinematic table, limits and collision-sphere radii of the handle (bmpc_set_robot)."""
        from . import robots
        table = {"iiwa14": robots.IIWA14, "gen3": robots.GEN3}[robot] if isinstance(robot, str) else robot
        r = robots.to_struct(table)
        self._chk(self.lib.bmpc_set_robot(self._h, ctypes.byref(r)), "bmpc_set_robot")
        self.robot = table

    def close(self):
        if getattr(self, "_h", None):
            self.lib.bmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc == 4:
            raise RuntimeError(f"{what} refused: the handle is in use by another thread (one handle per host thread)")
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.bmpc_last_error(self._h).decode()}")

    def solve_batch(self, x0, lbx, ubx, p, want_g=False, want_lam=False):
        x0, lbx, ubx, p = (np.ascontiguousarray(np.atleast_2d(a), float) for a in (x0, lbx, ubx, p))
        B = x0.shape[0]
        assert x0.shape == (B, self.n_w) and lbx.shape == x0.shape and ubx.shape == x0.shape and p.shape == (B, self.n_p)
        x = np.empty((B, self.n_w)); f = np.empty(B); viol = np.empty(B)
        g = np.empty((B, self.n_g)) if want_g else None
        iters = np.empty(B, np.int32); status = np.empty(B, np.int32)
        lam_g = np.empty((B, self.n_g)) if want_lam else None
        lam_x = np.empty((B, self.n_w)) if want_lam else None
        rc = self.lib.bmpc_solve(self._h, B, _P(x0), _P(lbx), _P(ubx), _P(p), _P(x), _P(g), _P(lam_g), _P(lam_x), _P(f),
                                 iters.ctypes.data_as(_ip), status.ctypes.data_as(_ip), _P(viol))
        self._chk(rc, "bmpc_solve")
        return dict(x=x, g=g, f=f, iters=iters, status=status, viol=viol, lam_g=lam_g, lam_x=lam_x)

    def stream(self):
        """The handle's own HIP stream as an integer (hipStream_t): the one bmpc_solve and bmpc_solve_dev_async run on."""
        return int(self.lib.bmpc_stream(self._h) or 0)

    INST_STATE_FIELDS = ("iters", "status", "mu", "alpha", "alpha_dual", "alpha_ftb", "delta_w", "hess_next", "retries", "backtracks",
                         "err_prev", "stall")

    def inst_state(self, B):
        """Diagnostic: [B][12] per-instance solver state of the most recent solve (bmpc_debug_inst_state; INST_STATE_FIELDS)."""
        out = np.zeros((B, 12))
        self._chk(self.lib.bmpc_debug_inst_state(self._h, int(B), _P(out)), "bmpc_debug_inst_state")
        return out

    def time_ric(self, on=True):
        """HIP events around every launch of the Riccati kernel, from the next solve on (bmpc_debug_time_ric)."""
        self._chk(self.lib.bmpc_debug_time_ric(self._h, int(bool(on))), "bmpc_debug_time_ric")

    def ric_stats(self):
        """{kernel: (summed launch ms, launches, instance-iterations)} of the most recent solve (bmpc_debug_ric_stats)."""
        out = np.zeros(6)
        self._chk(self.lib.bmpc_debug_ric_stats(self._h, _P(out)), "bmpc_debug_ric_stats")
        full = np.zeros(3)
        self._chk(self.lib.bmpc_debug_ric_stats_full(self._h, _P(full)), "bmpc_debug_ric_stats_full")
        return {"bmpc_k_ric": tuple(out[:3]), "bmpc_k_ric_lat": tuple(out[3:]), "bmpc_k_ric_full_batch": tuple(full)}

    def lane_stats(self):
        """The two lanes of the most recent closed-loop run without lock step on this handle (bmpc_debug_lane_stats)."""
        out = np.zeros(8)
        self._chk(self.lib.bmpc_debug_lane_stats(self._h, _P(out)), "bmpc_debug_lane_stats")
        return {"bursts": int(out[0]), "fast_super_steps": int(out[1]), "bulk_super_steps": int(out[2]),
                "fast_rounds": int(out[6]), "fast_lane_instances_mean": float(out[3] / max(out[6], 1.0))}

    def debug_spin(self, ms):
        """Diagnostic: occupy the handle's stream for `ms` milliseconds (bmpc_debug_spin)."""
        self._chk(self.lib.bmpc_debug_spin(self._h, int(ms)), "bmpc_debug_spin")

    def multipliers_dev(self, B, d_lam_g, d_lam_x, stream=0):
        """lam_g, lam_x (raw device pointers) of the most recent finished solve on this handle."""
        self._chk(self.lib.bmpc_multipliers_dev(self._h, B, d_lam_g, d_lam_x, stream or None), "bmpc_multipliers_dev")

    def solve_dev(self, B, d_x0, d_lbx, d_ubx, d_p, d_x, d_f, d_iters, d_status, d_viol, d_g=0, stream=0):
        """Raw device pointers (ints), asynchronous on `stream`."""
        rc = self.lib.bmpc_solve_dev(self._h, B, d_x0, d_lbx, d_ubx, d_p, d_x, d_g or None, d_f, d_iters, d_status,
                                     d_viol, stream or None)
        self._chk(rc, "bmpc_solve_dev")

    def solve_dev_async(self, B, d_x0, d_lbx, d_ubx, d_p, d_x, d_f, d_iters, d_status, d_viol, d_g=0):
        """Returns at once; the solve runs on the handle's own stream (one in flight per handle)."""
        rc = self.lib.bmpc_solve_dev_async(self._h, B, d_x0, d_lbx, d_ubx, d_p, d_x, d_g or None, d_f, d_iters,
                                           d_status, d_viol)
        self._chk(rc, "bmpc_solve_dev_async")

    def wait(self):
        self._chk(self.lib.bmpc_wait(self._h), "bmpc_wait")

    def active(self):
        """Unfinished instances of the solve in flight (0 when idle)."""
        return self.lib.bmpc_active(self._h)

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self.lib.bmpc_last_kernel_ms(self._h, ctypes.byref(ms))
        return ms.value

    def fk(self, q, dq=None):
        q = np.ascontiguousarray(q, float).reshape(-1, 7)
        B = q.shape[0]
        dq = None if dq is None else np.ascontiguousarray(dq, float).reshape(-1, 7)
        out = dict(ee_pos=np.empty((B, 3)), ee_rot=np.empty((B, 3, 3)), col_pts=np.empty((B, 6, 3)),
                   jac=np.empty((B, 6, 7)), dvdq=np.empty((B, 6, 7)))
        rc = self.lib.bmpc_fk(self._h, B, _P(q), _P(dq), _P(out["ee_pos"]), _P(out["ee_rot"]), _P(out["col_pts"]),
                              _P(out["jac"]), _P(out["dvdq"]))
        self._chk(rc, "bmpc_fk")
        return out


class _DM:
    """Minimal stand-in for casadi.DM results: `.full()` and numpy conversion."""

    def __init__(self, a):
        self._a = np.asarray(a, float)

    def full(self):
        return self._a.reshape(-1, 1) if self._a.ndim == 1 else self._a

    def __array__(self, dtype=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __float__(self):
        return float(self._a)


class HipNlpSolver:
    """Call-compatible replacement of the CasADi nlpsol function object of BoundMPC.py:240-246:
    sol = solver(x0=, lbx=, ubx=, lbg=, ubg=, p=) -> {"x","g","lam_g","lam_x","f"}; solver.stats()."""

    def __init__(self, N, dt=0.1, backend=None, **kw):
        self.backend = backend or HipBoundMPC(N, dt=dt, **kw)
        self.lbg, self.ubg = self.backend.lbg, self.backend.ubg
        self._stats = {}

    def __call__(self, x0, lbx, ubx, p, lbg=None, ubg=None, **_):
        inf2big = lambda a: np.nan_to_num(np.asarray(a, float), posinf=1e20, neginf=-1e20)
        r = self.backend.solve_batch(np.asarray(x0, float)[None], inf2big(lbx)[None], inf2big(ubx)[None],
                                     np.asarray(p, float)[None], want_g=True, want_lam=True)
        st = int(r["status"][0])
        self._stats = {"iter_count": int(r["iters"][0]), "success": st == 0,
                       "return_status": ["Solve_Succeeded", "Maximum_Iterations_Exceeded",
                                         "Search_Direction_Becomes_Too_Small", "Error_In_Step_Computation"][st],
                       "g_viol": float(r["viol"][0]), "t_kernel_ms": self.backend.last_kernel_ms()}
        n_w, n_g = self.backend.n_w, self.backend.n_g
        return {"x": _DM(r["x"][0]), "g": _DM(r["g"][0]), "f": _DM(r["f"][0]),
                "lam_g": _DM(r["lam_g"][0] if r["lam_g"] is not None else np.zeros(n_g)),
                "lam_x": _DM(r["lam_x"][0] if r["lam_x"] is not None else np.zeros(n_w))}

    def stats(self):
        return dict(self._stats)


def default_fk_fn():
    """Batched kinematics backed by the HIP library (used by RobotModel when no backend is given)."""
    be = HipBoundMPC(15)
    return lambda q, dq=None: be.fk(q, dq)
