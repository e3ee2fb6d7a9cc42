/*
 * bmpc_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's BoundMPC receding-horizon NLP
 * (bound_planner/BoundMPC/casadi_ocp_formulation.py:13-421, bound_mpc_functions.py:49-428,
 * mpc_utils_casadi.py:6-70, RobotModel/RobotModel.py:146-267 + iiwa.urdf) and of a
 * primal-dual interior-point solve of it.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the HIP product path never does.
 *
 * Parity status: the NLP functions (f, g, grad f, J_g) and the kinematics are PINNED against
 * golden vectors produced from the reference's own formulation code / serialized .ca tapes
 * (tests/golden/).  The solver boundary itself (CasADi 3.6.7's bundled IPOPT + MUMPS,
 * requirements.txt:1, call site BoundMPC.py:594-603) cannot be executed in this project
 * (no wheel, no network): PARITY AT THE IPOPT BOUNDARY IS UNPINNED.  The solver below follows
 * the published algorithm (Waechter & Biegler 2006: slack-based primal-dual IP,
 * fraction-to-boundary, filter line search, monotone barrier update, IPOPT's scaled termination error) on a
 * stage-condensed form of the same NLP, and is validated by KKT residuals of the pinned NLP and by an independent
 * SLSQP solve of the same NLP that reaches the same solutions (tests/test_independent_solver.py).
 */
#ifndef BMPC_ORACLE_H
#define BMPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define BMPC_NJ 7
#define BMPC_NSEG 4
#define BMPC_SET 15
#define BMPC_NP 875

typedef struct {
    int N;            /* horizon (reference default 15, util_functions.py:49) */
    double dt;        /* 0.1 */
    double tol;       /* IPOPT tol, BoundMPC.py:203 (10e-6 = 1e-5) */
    int max_iter;     /* BoundMPC.py:204 */
    int verbose;
    int hess;         /* 0 Gauss-Newton Hessian, 1 + second-order kinematic terms (default) */
    int mu_strategy;  /* 0 LOQO adaptive, 1 monotone Fiacco-McCormick (default), 2 / 3 probing (Mehrotra) rule */
    double hess_switch; /* hess==2: use second-order terms once the KKT error is below this */
    double mu_init, kappa_mu, theta_mu, kappa_eps; /* monotone barrier schedule (IPOPT names) */
    int inertia;      /* indefinite exact Hessian: 0 Gauss-Newton fallback (round 2), 1 delta_w inertia correction (IPOPT),
                         2 (default) Gauss-Newton fallback while the previous error is above inertia_err and the error improved
                         within the last stall_n iterations, delta_w otherwise */
    double dw0;       /* first delta_w (IPOPT delta_w^0 = 1e-4) */
    double inertia_err; /* 1e-2 */
    int stall_n;      /* 8 */
    int gn_backoff;   /* 2: after a Gauss-Newton fallback the exact Hessian is tried again after 1, then 2 iterations (0: every iteration) */
    int slack_reset;  /* 1: trial slacks t <- max(t + alpha dt, -h(trial point)) (0: round 2) */
    double ls_alpha_mem; /* m > 0: the line search starts at min(fraction-to-boundary length, m x the previous iteration's step length); 0 (default): off */
    double mu_floor_k; /* a barrier decrease stops at (scaled optimality error) / mu_floor_k; 0 = off (round 2); default 1e4 */
    /* oracle-only experiment switches (measured in DESIGN.md 2.2, no gain, so the kernels do not mirror them); all default 0 */
    int soc;           /* second-order corrections per iteration after a rejected first trial that raised theta (IPOPT: 4) */
    int soc_after;     /* ... only from this iteration on */
    int pi_shoot;      /* 1: the rotation-integral state follows its nonlinear dynamics exactly at every trial point */
} bmpc_oracle_opts;

void bmpc_oracle_default_opts(bmpc_oracle_opts* o, int N);

/* sizes: n_w = 44N+6, n_g = 147(N-1)+21, n_p = 875 */
void bmpc_oracle_dims(int N, int* n_w, int* n_g, int* n_p);

/* Kinematics (RobotModel.py:146-267): ee position (3), ee rotation (row-major 3x3),
 * 6 collision points (6x3 row-major: joint_3..joint_7 origins, link4_col_link),
 * geometric Jacobian LOCAL_WORLD_ALIGNED (6x7 row-major), dJ/dq . dq (6x7 row-major). */
void bmpc_oracle_fk(const double* q, const double* dq, double* ee_pos, double* ee_rot,
                    double* col_pts, double* jac, double* dvdq);

/* Kinematic table of another 7-joint arm (RobotModel.py:10-48, USE_IIWA = False: gen3_arm.urdf): joint <origin xyz> [7][3] and
 * <origin rpy> [7][3], the fixed joints to end_effector_link and link4_col_link.  Process-wide; NULL restores the iiwa14. */
void bmpc_oracle_set_robot(const double* joint_xyz21, const double* joint_rpy21, const double* ee_xyz, const double* ee_rpy,
                           const double* link4_col_xyz);

/* Full-space NLP functions in the reference's layout (casadi_ocp_formulation.py:89-101,
 * 383-417).  Any output pointer may be NULL.  jac_g is dense row-major n_g x n_w. */
int bmpc_oracle_eval(int N, double dt, const double* w, const double* p, double* f, double* g,
                     double* grad_f, double* jac_g);

/* Constant constraint bounds lbg/ubg (casadi_ocp_formulation.py:145-380); +-inf as +-1e20. */
void bmpc_oracle_gbounds(int N, double* lbg, double* ubg);

/* One solve = the call at BoundMPC.py:594-603.  status: 0 converged, 1 max_iter,
 * 2 stalled/infeasible, 3 numerical.  Outputs x (n_w), g (n_g), lam_g (n_g), lam_x (n_w), f,
 * iters, viol (sum of constraint violations as BoundMPC.py:613-615); NULL allowed for
 * g/lam_g/lam_x. */
int bmpc_oracle_solve(const bmpc_oracle_opts* o, const double* x0, const double* lbx,
                      const double* ubx, const double* p, double* x, double* g, double* lam_g,
                      double* lam_x, double* f, int* iters, int* status, double* viol);

/* Batched convenience (OpenMP over instances when built with -fopenmp). */
int bmpc_oracle_solve_batch(const bmpc_oracle_opts* o, int B, const double* x0,
                            const double* lbx, const double* ubx, const double* p, double* x,
                            double* f, int* iters, int* status, double* viol, int nthreads);

/* The same + info[B][BMPC_ORACLE_INFO]: decisions of the LAST iteration of each solve -- {iterations, status, mu, alpha (1e300: no
 * acceptable step), alpha_dual, fraction-to-boundary alpha, delta_w, exact Hessian wanted next, factorisation retries, rejected
 * trials, KKT error of the previous iterate, stall counter}; compared with the HIP path's per-instance state by
 * tests/test_iterate_parity.py, run by run with max_iter = 1, 2, 3, ... */
#define BMPC_ORACLE_INFO 12
int bmpc_oracle_solve_batch_info(const bmpc_oracle_opts* o, int B, const double* x0,
                                 const double* lbx, const double* ubx, const double* p, double* x,
                                 double* f, int* iters, int* status, double* viol, double* info, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
