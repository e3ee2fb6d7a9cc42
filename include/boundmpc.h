/*
 * boundmpc.h -- C ABI of libboundmpc_hip.so, the MI355X-native drop-in for the NLP solve of
 * BoundMPC's receding-horizon step.
 *
 * Reference interface replaced (all paths relative to /root/reference):
 *   bound_planner/BoundMPC/BoundMPC.py:594-603    sol = self.solver(x0=w0, lbx=lbx, ubx=ubx,
 *                                                 lbg=self.lbg, ubg=self.ubg, p=params)
 *   bound_planner/BoundMPC/BoundMPC.py:604-617    sol["x"], sol["g"], solver.stats()
 *   bound_planner/BoundMPC/BoundMPC.py:240-246    solver construction (setup_optimization_problem)
 *   bound_planner/RobotModel/RobotModel.py:146-267 fk_pos, fk_pos_col, hom_transform_endeffector,
 *                                                 jacobian_fk (the numeric Pinocchio path)
 * Vector layouts are the reference's own: decision vector w (44N+6, variable-major/time-minor,
 * casadi_ocp_formulation.py:89-101), parameter vector p (875, :383-415), constraint vector g
 * (147(N-1)+21, :106-380).  All host arrays are row-major [instance][index], FP64.
 *
 * Plain C, plain pointers and sizes; no torch/HIP types in the signatures (the `_dev` entry
 * takes a hipStream_t as void*).  A handle is not thread-safe; use one per host thread/stream.
 * Return value: 0 on success, nonzero on API misuse or a HIP error (see bmpc_last_error); 4 = the handle is busy with a
 * solve started from another host thread (nothing was done).
 */
#ifndef BOUNDMPC_H
#define BOUNDMPC_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bmpc_handle bmpc_handle;

typedef struct {
    int N;              /* horizon; reference default 15 (utils/util_functions.py:49) */
    int nr_segs;        /* must be 4 (utils/util_functions.py:49) */
    double dt;          /* 0.1 */
    double tol;         /* IPOPT "tol": 10e-6 = 1e-5 (BoundMPC.py:203) */
    int max_iter;       /* IPOPT "max_iter": 100 (BoundMPC.py:204) */
    int device;         /* HIP device ordinal */
    int hess;           /* 0 Gauss-Newton Hessian, 2 hybrid (second-order kinematic terms) */
    double hess_switch; /* hess==2: the exact Lagrangian Hessian (second-order kinematic and sigmoid terms) is tried once the KKT
                           error of the previous iterate is below this (default 1.0) */
    double mu_init, kappa_mu, theta_mu, kappa_eps; /* monotone barrier schedule */
    double mu_floor_k;  /* a barrier decrease stops at (scaled KKT error) / mu_floor_k (default 1e4; 0 = no floor) */
    int inertia;        /* exact Hessian not positive definite on the null space of the dynamics: 0 Gauss-Newton fallback,
                           1 IPOPT's inertia correction (delta_w I added, escalated), 2 (default) Gauss-Newton fallback while
                           the previous KKT error is above inertia_err and the error improved within the last stall_n
                           iterations, inertia correction otherwise */
    double dw0;         /* first delta_w (IPOPT delta_w^0 = 1e-4) */
    double inertia_err; /* 1e-2 */
    int stall_n;        /* 8 */
    int slack_reset;    /* 1 (default): a trial slack is never below the value that closes its row at the trial point,
                           t <- max(t + alpha dt, -h(trial)) (Byrd-Hribar-Nocedal slack reset); 0: off */
    double ls_alpha_mem; /* 0 (default): the filter line search always starts at the fraction-to-boundary length (IPOPT); m > 0: at
                           min(that, m x the step length the previous iteration ended with) -- fewer rejected trials on iterates that
                           crawl, measured neutral on configs[2] and +5 % on configs[4] with m = 4 */
    int gn_backoff;     /* 2: after a Gauss-Newton fallback the exact Hessian is tried again after 1, then 2 iterations
                           (a failed attempt costs a Riccati sweep); 0: every iteration */
    int trial_repeats;  /* 9 (default): a rejected line-search trial is repeated (half the step length) up to this many times by the
                           wavefront that evaluated it (bmpc_k_trial holds all pairs of its instances, so it runs the filter test
                           itself): the whole line search of an iteration in one super-step.  0: one trial per super-step
                           (rounds 1-2).  Scheduling only, results do not depend on it (bitwise) */
    int watchdog_ms;    /* > 0 (default 30000): a wait for the GPU gives up after this long and the call returns 5 with a message
                           (the handle is unusable afterwards: every entry point returns 5, and bmpc_destroy neither waits for the
                           stream nor frees device memory that queued kernels may still write -- it leaks them; end the process
                           with an error and let a fresh one take over); 0: plain hipStreamSynchronize */
    int max_batch;      /* capacity hint for host-pointer calls (device staging buffers) */
    int pool_slots;     /* 0 (default) = every instance of a call has its own workspace slot; > 0 = the workspace
                           holds this many instances and a call with more of them STREAMS them through it: a slot whose
                           instance has finished takes the next one, so a long call keeps the GPU on ~pool_slots instances
                           and pays a single straggler tail.  Results do not depend on it (bitwise). */
} bmpc_opts;

void bmpc_default_opts(bmpc_opts* o, int N);

/* Kinematic table of a 7-joint serial arm in the frame conventions of RobotModel.py:15-54: revolute joints about their
 * local z axes with the URDF <origin xyz rpy> of joint_1..joint_7 (fixed rotation R = Rz(yaw) Ry(pitch) Rx(roll)), the fixed
 * joints to end_effector_link and link4_col_link (child of joint_4's link), URDF limits, BoundMPC.py:171-191 acceleration /
 * jerk limits and the collision-sphere radii col_joint_sizes (RobotModel.py:37-40).  The six collision points are the
 * origins of joint_3..joint_7 and link4_col_link (RobotModel.py:27-35).  Infinite joint limits: +-1e20. */
typedef struct {
    double joint_xyz[7][3], joint_rpy[7][3];
    double ee_xyz[3], ee_rpy[3];
    double link4_col_xyz[3];
    double q_lower[7], q_upper[7], dq_max[7];
    double ddq_max, u_max;
    double col_joint_sizes[7];
} bmpc_robot;
void bmpc_robot_iiwa14(bmpc_robot* r);   /* RobotModel/iiwa.urdf (USE_IIWA = True, the default; RobotModel.py:10) */
void bmpc_robot_gen3(bmpc_robot* r);     /* RobotModel/gen3_arm.urdf (USE_IIWA = False): Kinova Gen3, joints 1/3/5/7 unlimited */

/* replaces setup_optimization_problem(...) + nlpsol construction (BoundMPC.py:240-246) */
int bmpc_create(const bmpc_opts* o, bmpc_handle** h);
void bmpc_destroy(bmpc_handle* h);
const char* bmpc_last_error(const bmpc_handle* h);

/* n_w = 44N+6, n_g = 147(N-1)+21, n_p = 875 */
int bmpc_dims(const bmpc_handle* h, int* n_w, int* n_g, int* n_p);

/* the handle's own HIP stream (a hipStream_t): the one bmpc_solve, bmpc_solve_dev_async and the device loop run on */
void* bmpc_stream(bmpc_handle* h);

/* Robot of the handle (default: iiwa14).  bmpc_set_robot must precede the solves / device loops that are to use it. */
int bmpc_set_robot(bmpc_handle* h, const bmpc_robot* r);
int bmpc_get_robot(const bmpc_handle* h, bmpc_robot* r);

/* the options the handle was created with */
int bmpc_get_opts(const bmpc_handle* h, bmpc_opts* o);

/* constant constraint bounds (self.lbg / self.ubg, casadi_ocp_formulation.py:145-380);
 * infinities are returned as +-1e20 */
int bmpc_gbounds(const bmpc_handle* h, double* lbg, double* ubg);

/* B independent solves = B calls of self.solver(...) (BoundMPC.py:594-603).  Host pointers.
 * x0/lbx/ubx/x: [B][n_w]; p: [B][875]; g: [B][n_g] or NULL; lam_g: [B][n_g] or NULL; lam_x: [B][n_w] or
 * NULL -- sol["lam_g"], sol["lam_x"] (BoundMPC.py:638-645) in CasADi's convention: grad f + J_g^T lam_g + lam_x = 0,
 * positive at an active upper bound, negative at an active lower bound; the entries of the variables that are fixed
 * by lbx == ubx follow from stationarity (IPOPT fixed_variable_treatment=make_parameter).  On a handle created with
 * pool_slots > 0 the multipliers need B <= pool_slots (a streamed call keeps no final iterates): otherwise the call is refused
 * up front with return code 1.
 * f/viol: [B]; iters/status: [B].
 * status: 0 converged, 1 max_iter, 2 stalled, 3 numerical.  viol = sum of constraint
 * violations exactly as BoundMPC.py:613-615, so the caller reproduces
 * `success = stats["success"] or g_viol < 1e-4`.  Infinite bounds may be passed as +-inf or
 * +-1e20. */
int bmpc_solve(bmpc_handle* h, int B, const double* x0, const double* lbx, const double* ubx,
               const double* p, double* x, double* g, double* lam_g, double* lam_x, double* f,
               int* iters, int* status, double* viol);

/* Same with DEVICE pointers; all work is enqueued on `stream` (a hipStream_t).  The call returns
 * when the batch is solved: the interior-point iteration count is data dependent, so the host
 * polls the number of unfinished instances between bursts of launches.  On return `stream`
 * has been synchronised: the outputs are complete and the handle's workspace is free for the next call
 * (on any stream).  A solve started with bmpc_solve_dev_async is waited for first. */
int bmpc_solve_dev(bmpc_handle* h, int B, const double* d_x0, const double* d_lbx,
                   const double* d_ubx, const double* d_p, double* d_x, double* d_g, double* d_f,
                   int* d_iters, int* d_status, double* d_viol, void* stream);

/* Multipliers lam_g [B][n_g], lam_x [B][n_w] (device pointers) of the most recent finished solve on this handle (any
 * entry point): the final iterate stays in the handle's workspace until the next solve.  Enqueued on
 * `stream` and waited for. */
int bmpc_multipliers_dev(bmpc_handle* h, int B, double* d_lam_g, double* d_lam_x, void* stream);

/* Asynchronous form: returns at once, the solve runs on the handle's own stream driven by a worker
 * thread; inputs must already be complete on the device.  One solve in flight per handle;
 * bmpc_wait() blocks until it has finished and returns its status.  Two handles used alternately
 * overlap the straggler tail of one batch with the bulk of the next (bench.py). */
int bmpc_solve_dev_async(bmpc_handle* h, int B, const double* d_x0, const double* d_lbx,
                         const double* d_ubx, const double* d_p, double* d_x, double* d_g, double* d_f,
                         int* d_iters, int* d_status, double* d_viol);
int bmpc_wait(bmpc_handle* h);
/* unfinished instances of the solve in flight on this handle (0 when idle) */
int bmpc_active(bmpc_handle* h);

/* Batched kinematics (RobotModel.py:146-267): ee_pos [B][3], ee_rot [B][9] row-major,
 * col_pts [B][18] (joint_3..joint_7 origins, link4_col_link), jac [B][42] (6x7 geometric,
 * LOCAL_WORLD_ALIGNED), dvdq [B][42] = d(J dq)/dq.  Host pointers; outputs may be NULL. */
int bmpc_fk(bmpc_handle* h, int B, const double* q, const double* dq, double* ee_pos,
            double* ee_rot, double* col_pts, double* jac, double* dvdq);

/* Duration (ms) of the most recent solve kernel measured with HIP events on its stream
 * (bmpc_solve: events around the launch; bmpc_solve_dev: caller must have synchronised). */
int bmpc_last_kernel_ms(bmpc_handle* h, float* ms);

/* Diagnostic builds (-DBMPC_PROFILE) only: per-phase shader-cycle sums of the last launches. */
int bmpc_debug_phase_cycles(bmpc_handle* h, double* out16);
/* Diagnostic: per-instance solver state of the most recent finished solve (B rows of 12 doubles, host memory): iterations, status,
 * mu, alpha (1e300: no acceptable step), alpha_dual, fraction-to-boundary alpha, delta_w, exact Hessian wanted next, factorisation
 * retries, rejected line-search trials, KKT error of the previous iterate, stall counter.  With max_iter = k: the decisions of
 * iteration k - 1, which the iterate-for-iterate parity test compares with the oracle's.  B <= workspace slots. */
int bmpc_debug_inst_state(bmpc_handle* h, int B, double* out);
/* Measurement: from the next solve on, HIP events bracket every launch of the Riccati kernel on the handle's stream
 * (bmpc_debug_time_ric(h, 1)); bmpc_debug_ric_stats then returns for the most recent solve {summed launch durations [ms], launches,
 * instance-iterations} of the throughput variant of that kernel in out6[0..2] and of its latency variant (nearly empty super-steps)
 * in out6[3..5].  bench.py derives its roofline line from these. */
int bmpc_debug_time_ric(bmpc_handle* h, int on);
int bmpc_debug_ric_stats(bmpc_handle* h, double* out6);
/* ... and of those launches of the throughput variant whose grid was the whole batch (the first super-steps of a solve, before anybody
 * has finished: the kernel at full occupancy): {summed durations [ms], launches, instance-iterations}. */
int bmpc_debug_ric_stats_full(bmpc_handle* h, double* out3);
/* Measurement: the two lanes of the most recent bmpc_loop_run_async on this handle (see there): {bursts, fast-lane super-steps,
 * bulk-lane super-steps, summed fast-lane instance counts at the ends of its rounds, 0, 0, fast-lane rounds, 0}; all zero when the
 * run had one lane. */
int bmpc_debug_lane_stats(bmpc_handle* h, double* out8);
/* Diagnostic: keeps the handle's stream busy for `ms` milliseconds (at most 10 s, then the kernel ends by itself), so that the
 * watchdog (bmpc_opts.watchdog_ms) can be exercised without a kernel that really hangs. */
int bmpc_debug_spin(bmpc_handle* h, int ms);

/* ---------------------------------------------------------------------------------------------
 * Device-resident closed loop: R rollouts advanced in lock step, one batched solve per MPC step,
 * per-rollout state in HBM.  Replaces, for every rollout and step, the host code around the solver
 * call: BoundMPC.step before the solve (BoundMPC.py:388-589), the acceptance test and
 * compute_return_data (BoundMPC.py:604-1040), ReferencePath.update (ReferencePath.py:187-207),
 * MPCNode.step's state advance with integrate_joint (MPCNode.py:106-160, util_functions.py:55-65).
 * Plan-time construction (ReferencePath.__init__, BoundMPC.update) stays with the caller, who
 * serialises it into the state vector: bmpc_loop_state_doubles() doubles per rollout, fields located by
 * name with bmpc_loop_field() (names = LP_FIELDS of boundplanner_amd/csrc/bmpc_loop.hpp).
 * Per-step collision sets (ConvexSetFinder.find_set_collision_avoidance, ConvexSetFinder.py:309-375) are computed on the
 * device too: boxes around the collision points, plus separating halfspaces of the scene obstacles set with
 * bmpc_loop_set_obstacles (shared by all rollouts of the loop).
 * All pointers below are HOST pointers.  The loop borrows the handle's solver and stream: do not use the handle
 * for other solves while a loop call is running.  The loop keeps the handle alive: a bmpc_destroy(handle)
 * issued while loops exist is deferred until the last bmpc_loop_destroy. */
typedef struct bmpc_loop bmpc_loop;
int bmpc_loop_state_doubles(void);
int bmpc_loop_log_doubles(void);
int bmpc_loop_field(const char* name, int* offset, int* count);
int bmpc_loop_create(bmpc_handle* h, int R, bmpc_loop** out);
void bmpc_loop_destroy(bmpc_loop* l);
const char* bmpc_loop_last_error(const bmpc_loop* l);
/* scene obstacles (BoundMPC.obstacles as polytopes A x <= b with their vertices; ConvexSetFinder.py:309-375):
 * A [n_obs][15][3] and b [n_obs][15] (first nrows[o] rows used), V [n_obs][32][3] (first nv[o] vertices used);
 * n_obs <= 16; n_obs = 0 clears the scene */
int bmpc_loop_set_obstacles(bmpc_loop* l, int n_obs, const double* A, const double* b, const int* nrows, const double* V,
                            const int* nv);
/* state: [count][state_doubles]; prev: [count][n_w] previous solutions (warm start) or NULL */
int bmpc_loop_upload(bmpc_loop* l, int first, int count, const double* state, const double* prev);
int bmpc_loop_download(bmpc_loop* l, int first, int count, double* state, double* prev);
/* nsteps MPC steps of all rollouts; log: [nsteps][R][log_doubles] or NULL -- per row: iters, status,
 * viol, error_count, dead, phi, phi_max, split_idx[1], sector, switch, p_lie(6), q(7).
 * ms_total: HIP-event time of the whole run on the loop's stream; ms_solve: host time inside the solves */
int bmpc_loop_run(bmpc_loop* l, int nsteps, double* log, float* ms_total, float* ms_solve);
/* Trace records with the content of the reference's message boundmpcmsg/msg/MPCData.msg:1-64, written by the finish kernel for
 * the rollouts selected with bmpc_loop_set_record (n = 0 switches them off; they cost 16 + 64 N + 600 doubles per rollout and
 * step).  bmpc_loop_records copies the records of the last bmpc_loop_run / bmpc_loop_finish: out [steps][n][record_doubles(N)] with room for
 * max_steps steps (rc 1 when more were recorded; out == NULL returns only *steps, to size the buffer; records of dead rollouts are zero),
 * each: header (iters, status, viol, error_count, valid stages n, sector, phi_max, split_idxs[5], next selector, 0, 0, 0);
 * N stage blocks of 64 (p 6, v 6, q 7, dq 7, ddq 7, dddq 7, phi, dphi, e_p 3, de_p 3, e_r 3, de_r 3, e_r_orth1, e_r_par,
 * e_r_orth2, p_ref 6, segment; zero beyond n); the a_set / b_set / a_set_joints / b_set_joints blocks of the step's parameter
 * vector (600).  boundplanner_amd.mpc_data.from_device_record decodes them.  Lock-step runs only. */
int bmpc_loop_record_doubles(int N);
int bmpc_loop_set_record(bmpc_loop* l, int n, const int* rollouts);
int bmpc_loop_records(bmpc_loop* l, double* out, int max_steps, int* steps);
/* The same nsteps MPC steps of all rollouts WITHOUT lock step: the rollouts are independent, so each one starts its next
 * step as soon as its own solve has retired (its workspace slot is re-admitted with the next problem, prepared on the
 * device) instead of waiting for the slowest solve of the batch at every step.  Same log as bmpc_loop_run (bitwise). */
int bmpc_loop_run_async(bmpc_loop* l, int nsteps, double* log, float* ms_total);
/* the three phases of one step separately, and access to the solver arguments / solution (tests) */
int bmpc_loop_prepare(bmpc_loop* l);
int bmpc_loop_solve(bmpc_loop* l);
int bmpc_loop_finish(bmpc_loop* l, double* log);
int bmpc_loop_problem(bmpc_loop* l, double* x0, double* lbx, double* ubx, double* p);
int bmpc_loop_solution(bmpc_loop* l, double* x, int* iters, int* status, double* viol);
int bmpc_loop_set_solution(bmpc_loop* l, const double* x, const int* iters, const int* status, const double* viol);

#ifdef __cplusplus
}
#endif
#endif
