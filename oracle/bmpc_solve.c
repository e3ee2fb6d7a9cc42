/* ORACLE (test infrastructure).  Primal-dual interior-point solve of the BoundMPC NLP.
 *
 * What it replaces: the call sol = self.solver(x0, lbx, ubx, lbg, ubg, p) at
 * /root/reference/bound_planner/BoundMPC/BoundMPC.py:594-603, i.e. CasADi 3.6.7's bundled
 * IPOPT (+MUMPS) applied to the NLP of casadi_ocp_formulation.py.  IPOPT is a third-party
 * dependency that is absent from /root/reference and from this image, so this file restates
 * the PUBLISHED algorithm family (Waechter & Biegler, Math. Prog. 106, 2006: slack-based
 * primal-dual barrier method, fraction-to-boundary rule, filter line search, monotone Fiacco-McCormick
 * barrier update (default; the LOQO rule of IPOPT's mu_oracle=loqo is available as mu_strategy=0),
 * scaled optimality error of eq. (5)-(6) with tol/dual_inf_tol/constr_viol_tol/compl_inf_tol)
 * on an equivalent stage-condensed form of the same NLP:
 *
 *   - p[:3]=fk(q) and v=J(q)dq (equalities :128,:121-125) are substituted;
 *   - stage 0 is pinned by lbx==ubx (BoundMPC.py:544-580) and removed (IPOPT's
 *     fixed_variable_treatment=make_parameter does the same);
 *   - hat-function dynamics are rewritten x~_{k+1} = A x~_k + b u_k with
 *     x_k = x~_k + B1 u_k, pi_k = p_rot_k - dt/2 w_k, rs~_k = rs_k - dt/2 drs_k (bijective);
 *   - the 6 global dslacks ride along as a constant state;
 *   - the Hessian is Gauss-Newton/convex (exact first derivatives): the KKT points are those
 *     of the reference NLP, the iteration path is not IPOPT's.
 * The banded KKT system is solved by a dense Riccati recursion (n_x=32, n_u=9).
 * PARITY AT THE IPOPT BOUNDARY IS UNPINNED (see bmpc_oracle.h).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "bmpc_internal.h"

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
/* diagnostic counters over all solves of the process (not thread-exact): the pass counts quoted in DESIGN.md 2.2 */
long bmpc_dbg_soc_try = 0, bmpc_dbg_soc_acc = 0, bmpc_dbg_soc_pass = 0;
long bmpc_dbg_iters = 0, bmpc_dbg_sweeps = 0, bmpc_dbg_retry_iters = 0, bmpc_dbg_trials = 0;   /* diagnostics (not thread-exact) */

#define NX 32
#define NU 9
#define NZ 41
#define MAXROWS 216
#define BIG 1e19

/* natural stage coordinates y (41) */
enum { Y_Q = 0, Y_DQ = 7, Y_DDQ = 14, Y_U = 21, Y_PI = 28, Y_RS = 31, Y_DRS = 32, Y_PS = 33, Y_DPS = 34, Y_D = 35 };
/* zeta = (x, w) coordinates */
enum { Z_Q = 0, Z_DQ = 7, Z_DDQ = 14, Z_PI = 21, Z_RS = 24, Z_PS = 25, Z_D = 26, Z_U = 32, Z_DRS = 39, Z_DPS = 40 };
/* pose-group local coordinates (15): p_pos 0..2, p_rot 3..5, v 6..11, ps 12, rs 13, d5 14 */
enum { L_PS = 12, L_RS = 13, L_D5 = 14, NLOC = 15 };

enum { KIND_POSE = 0, KIND_PT = 1, KIND_SPARSE = 2 };

typedef struct {
    int kind, grp;       /* grp: collision point index for KIND_PT */
    double a[NLOC];      /* local gradient (POSE: 15, PT: 4 = point3 + d coef) */
    int i0, i1;          /* KIND_SPARSE: y indices (-1 = none) */
    double c0, c1;
    double cst;          /* unused */
    int gidx;            /* index of the row in the stage's 112 (+21) inequality rows of g, or -1 */
    int gsign;           /* +1: row is g <= 0 (lam_g = z); -1: row is g >= 0 (lam_g = -z) */
    int xidx;            /* bound rows: index in w (lam_x += xsign * z), or -1 */
    int xsign;
} row_t;

typedef struct {
    double zeta[NZ];
    int nrows;
    row_t rows[MAXROWS];
    double h[MAXROWS], t[MAXROWS], z[MAXROWS], dt_[MAXROWS], dz_[MAXROWS];
    double Jpose[NLOC][NZ];   /* d(local pose coords)/dy */
    double Jpt[6][4][NZ];     /* d(point, d_i)/dy */
    double costH[12 * 12], costg[12]; /* output-space cost model */
    double costHx[6 * 6];             /* second-order part of the sigmoid-weighted error terms (pose block), hess mode only */
    double A[NX * NX], B[NX * NU], r[NX];
    double H[NZ * NZ], g[NZ], gdual[NZ];
    double P[NX * NX], pv[NX], K[NU * NX], kf[NU];
    double dzeta[NZ], lam[NX];
    double fval;              /* stage cost value */
    double prot[3];           /* p_rot_k = pi_k + dt/2 w_k */
    double ppos[3], v[6];
    bmpc_kin kin;
    double J[6][7];
} stage_t;

typedef struct {
    int N;
    double dt, c1, c2, c3, b1, b2, b3;
    const double* p;
    double T[NZ * NZ];        /* y = T zeta */
    double As[NX * NX], Bs[NX * NU];
    double *lbq, *ubq;        /* [N][28] */
    double x1fix[24];         /* required fixed part of x_1 (q~,dq~,ddq~,pi) */
    double r0[NX];            /* initial defect on the fixed part */
    double iw0[3];            /* p[0,3:] pinned */
    double sl0[6];
    int no_sigma;             /* debug: leave the barrier terms out of H */
    int hess;                 /* 0: Gauss-Newton, 1: + kinematic curvature terms */
    double hreg;              /* Levenberg regularisation added to every stage Hessian diagonal */
    stage_t* st;              /* index 1..N-1 */
} prob_t;

void bmpc_oracle_default_opts(bmpc_oracle_opts* o, int N) {
    o->N = N;
    o->dt = 0.1;
    o->tol = 1e-5;
    o->max_iter = 100;
    o->verbose = 0;
    o->hess = 2;
    o->mu_strategy = 1;
    o->hess_switch = 1.0;
    o->mu_init = 0.1; o->kappa_mu = 0.1; o->theta_mu = 2.0; o->kappa_eps = 1000.0;
    o->inertia = 2; o->dw0 = 1e-4; o->inertia_err = 1e-2; o->stall_n = 8; o->mu_floor_k = 1e4; o->gn_backoff = 2; o->slack_reset = 1; o->ls_alpha_mem = 0.0;
    o->soc = 0; o->soc_after = 0; o->pi_shoot = 0;
}

/* ---------------------------------------------------------------- small dense helpers */
static void matmul(const double* A, const double* B, double* C, int m, int k, int n) { /* C = A B */
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void matTmul(const double* A, const double* B, double* C, int m, int k, int n) { /* C = A^T B, A is k x m */
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int l = 0; l < k; l++) s += A[l * m + i] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static int chol(double* A, int n) { /* in-place lower Cholesky, row-major */
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int l = 0; l < j; l++) d -= A[j * n + l] * A[j * n + l];
        if (!(d > 0)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int l = 0; l < j; l++) s -= A[i * n + l] * A[j * n + l];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const double* L, int n, double* b, int nrhs) { /* b (n x nrhs) <- A^-1 b */
    for (int c = 0; c < nrhs; c++) {
        for (int i = 0; i < n; i++) {
            double s = b[i * nrhs + c];
            for (int l = 0; l < i; l++) s -= L[i * n + l] * b[l * nrhs + c];
            b[i * nrhs + c] = s / L[i * n + i];
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = b[i * nrhs + c];
            for (int l = i + 1; l < n; l++) s -= L[l * n + i] * b[l * nrhs + c];
            b[i * nrhs + c] = s / L[i * n + i];
        }
    }
}

/* ---------------------------------------------------------------- problem setup */
static void build_T(prob_t* pb) {
    double* T = pb->T;
    memset(T, 0, sizeof pb->T);
    double dt = pb->dt;
    for (int i = 0; i < 7; i++) {
        T[(Y_Q + i) * NZ + Z_Q + i] = 1;   T[(Y_Q + i) * NZ + Z_U + i] = pb->c3;
        T[(Y_DQ + i) * NZ + Z_DQ + i] = 1; T[(Y_DQ + i) * NZ + Z_U + i] = pb->c2;
        T[(Y_DDQ + i) * NZ + Z_DDQ + i] = 1; T[(Y_DDQ + i) * NZ + Z_U + i] = pb->c1;
        T[(Y_U + i) * NZ + Z_U + i] = 1;
    }
    for (int i = 0; i < 3; i++) T[(Y_PI + i) * NZ + Z_PI + i] = 1;
    T[Y_RS * NZ + Z_RS] = 1;  T[Y_RS * NZ + Z_DRS] = dt / 2;
    T[Y_DRS * NZ + Z_DRS] = 1;
    T[Y_PS * NZ + Z_PS] = 1;  T[Y_PS * NZ + Z_DPS] = dt / 2;
    T[Y_DPS * NZ + Z_DPS] = 1;
    for (int i = 0; i < 6; i++) T[(Y_D + i) * NZ + Z_D + i] = 1;
    /* structured linear dynamics x+ = As x + Bs w */
    memset(pb->As, 0, sizeof pb->As);
    memset(pb->Bs, 0, sizeof pb->Bs);
    for (int i = 0; i < NX; i++) pb->As[i * NX + i] = 1;
    for (int i = 0; i < 7; i++) {
        pb->As[(Z_Q + i) * NX + Z_DQ + i] = dt;
        pb->As[(Z_Q + i) * NX + Z_DDQ + i] = dt * dt / 2;
        pb->As[(Z_DQ + i) * NX + Z_DDQ + i] = dt;
        pb->Bs[(Z_Q + i) * NU + i] = pb->b3;
        pb->Bs[(Z_DQ + i) * NU + i] = pb->b2;
        pb->Bs[(Z_DDQ + i) * NU + i] = pb->b1;
    }
    pb->Bs[Z_RS * NU + 7] = dt;
    pb->Bs[Z_PS * NU + 8] = dt;
}

static void zeta_to_y(const prob_t* pb, const double* zeta, double* y) {
    for (int i = 0; i < NZ; i++) {
        double s = 0;
        for (int j = 0; j < NZ; j++) s += pb->T[i * NZ + j] * zeta[j];
        y[i] = s;
    }
}

static void add_row(stage_t* s, const row_t* r, double h) {
    if (s->nrows >= MAXROWS) abort();
    s->rows[s->nrows] = *r;
    s->h[s->nrows] = h;
    s->nrows++;
}

/* Evaluate stage k at its current zeta: cost model, rows, dynamics linearisation.
 * mode 0: full (derivatives); mode 1: values only (line search).  Returns stage cost. */
static void eval_stage(prob_t* pb, int k, int mode) {
    stage_t* s = &pb->st[k];
    const double* p = pb->p;
    const double* wts = p + P_W;
    int N = pb->N, term = (k == N - 1);
    double dt = pb->dt;
    double y[NZ];
    zeta_to_y(pb, s->zeta, y);
    bmpc_kin kin;
    double J[6][7], G[6][7];
    bmpc_kin_eval(y + Y_Q, &kin);
    bmpc_kin_jac(&kin, J);
    bmpc_kin_dvdq(&kin, J, y + Y_DQ, G);
    double pose[6], v[6];
    for (int a = 0; a < 6; a++) {
        v[a] = 0;
        for (int j = 0; j < 7; j++) v[a] += J[a][j] * y[Y_DQ + j];
    }
    for (int a = 0; a < 3; a++) {
        pose[a] = kin.pee[a];
        pose[3 + a] = y[Y_PI + a] + 0.5 * dt * v[3 + a];
        s->prot[a] = pose[3 + a];
        s->ppos[a] = pose[a];
    }
    memcpy(s->v, v, sizeof v);
    s->kin = kin;
    memcpy(s->J, J, sizeof J);
    bmpc_seg sc;
    bmpc_pose_eval pe;
    bmpc_seg_ctx(N, p, k, &sc);
    bmpc_pose_eval_fn(&sc, pose, v, pb->iw0, p[P_PHIMAX], &pe);
    double giw[3];
    double fv = bmpc_stage_cost_o(&sc, &pe, v, wts, p + P_XPHID, term, s->costg, giw, mode == 0 ? s->costH : NULL);
    if (mode == 0) {
        /* what the Gauss-Newton model of r = sig(phi) e leaves out of the Hessian of sig^2 (|e_r|^2 + |e_p|^2)
         * (casadi_ocp_formulation.py:272-276): 2 sig [ sig'' |e|^2 dphi dphi^T + sig' (ge dphi^T + dphi ge^T) ],
         * ge = De^T e; e is linear in the pose, phi = (p - p_ref) . dp */
        double e2 = dot3(pe.er, pe.er) + dot3(pe.ep, pe.ep);
        double d2sig = 60.0 * pe.dsig * (1.0 - 2.0 * pe.sig);
        double ge[6], dph[6];
        for (int b = 0; b < 6; b++) {
            double sm = 0;
            for (int a = 0; a < 3; a++) sm += pe.Der[a][b] * pe.er[a] + (b < 3 ? pe.Dep[a][b] * pe.ep[a] : 0.0);
            ge[b] = sm;
            dph[b] = b < 3 ? sc.dp[b] : 0.0;
        }
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++)
                s->costHx[6 * i + j] = 2 * pe.sig * (d2sig * e2 * dph[i] * dph[j] + pe.dsig * (ge[i] * dph[j] + dph[i] * ge[j]));
    }
    for (int j = 2; j <= 4; j++) fv += wts[6] * y[Y_DQ + j] * y[Y_DQ + j];
    for (int j = 0; j < 7; j++) fv += wts[7] * y[Y_U + j] * y[Y_U + j];
    fv += wts[9] * y[Y_RS] * y[Y_RS] + wts[10] * y[Y_DRS] * y[Y_DRS] + wts[9] * y[Y_PS] * y[Y_PS] + wts[10] * y[Y_DPS] * y[Y_DPS];
    double sl[6];
    for (int i = 0; i < 6; i++) sl[i] = pb->sl0[i] + y[Y_D + i];
    if (term)
        for (int i = 0; i < 6; i++) {
            if (i != 4) fv += wts[8] * sl[i] * sl[i];
            fv += wts[10] * y[Y_D + i] * y[Y_D + i];
        }
    s->fval = fv;

    /* ---- group Jacobians ---- */
    if (mode == 0) {
        memset(s->Jpose, 0, sizeof s->Jpose);
        memset(s->Jpt, 0, sizeof s->Jpt);
        for (int a = 0; a < 3; a++) {
            for (int j = 0; j < 7; j++) {
                s->Jpose[a][Y_Q + j] = J[a][j];
                s->Jpose[3 + a][Y_Q + j] = 0.5 * dt * G[3 + a][j];
                s->Jpose[3 + a][Y_DQ + j] = 0.5 * dt * J[3 + a][j];
            }
            s->Jpose[3 + a][Y_PI + a] = 1;
        }
        for (int a = 0; a < 6; a++)
            for (int j = 0; j < 7; j++) {
                s->Jpose[6 + a][Y_Q + j] = G[a][j];
                s->Jpose[6 + a][Y_DQ + j] = J[a][j];
            }
        s->Jpose[L_PS][Y_PS] = 1;
        s->Jpose[L_RS][Y_RS] = 1;
        s->Jpose[L_D5][Y_D + 5] = 1;
        for (int i = 0; i < 6; i++) {
            double Jp[3][7];
            bmpc_kin_point_jac(&kin, kin.pc[i], BMPC_COL_NJ[i], Jp);
            for (int a = 0; a < 3; a++)
                for (int j = 0; j < 7; j++) s->Jpt[i][a][Y_Q + j] = Jp[a][j];
            s->Jpt[i][3][Y_D + i] = 1;
        }
    }

    /* ---- rows: h_i(y) <= 0 ---- */
    s->nrows = 0;
    row_t r;
    /* box bounds on q, dq, ddq, u (BoundMPC.py:171-186, 544-589) */
    for (int j = 0; j < 28; j++) {
        double lb = pb->lbq[k * 28 + j], ub = pb->ubq[k * 28 + j];
        memset(&r, 0, sizeof r);
        r.kind = KIND_SPARSE; r.i1 = -1; r.gidx = -1;
        r.xidx = (j / 7) * 7 * N + (j % 7) * N + k;
        if (ub < BIG) { r.i0 = j; r.c0 = 1; r.xsign = 1; add_row(s, &r, y[j] - ub); }
        if (lb > -BIG) { r.i0 = j; r.c0 = -1; r.xsign = -1; add_row(s, &r, lb - y[j]); }
    }
    /* slack variables >= 0 (Q6: all four per stage) */
    {
        int idx[4] = {Y_RS, Y_DRS, Y_PS, Y_DPS};
        int widx[4] = {W_RS(N) + k, W_DRS(N) + k, W_PS(N) + k, W_DPS(N) + k};
        for (int m = 0; m < 4; m++) {
            memset(&r, 0, sizeof r);
            r.kind = KIND_SPARSE; r.i0 = idx[m]; r.c0 = -1; r.i1 = -1;
            r.gidx = -1; r.xidx = widx[m]; r.xsign = -1;
            add_row(s, &r, -y[idx[m]]);
        }
    }
    if (k == 1) {
        /* stage-0 slacks: rs_0, drs_0 >= 0 reach any rs~_1 = rs_0 + dt/2 drs_0 >= 0 */
        memset(&r, 0, sizeof r);
        r.kind = KIND_SPARSE; r.i0 = Y_RS; r.c0 = -1; r.i1 = Y_DRS; r.c1 = dt / 2;
        r.gidx = -1; r.xidx = -1;       /* multipliers of the stage-0 slacks follow from stationarity */
        add_row(s, &r, -(y[Y_RS] - dt / 2 * y[Y_DRS]));
        r.i0 = Y_PS; r.i1 = Y_DPS;
        add_row(s, &r, -(y[Y_PS] - dt / 2 * y[Y_DPS]));
        for (int i = 0; i < 6; i++) {
            memset(&r, 0, sizeof r);
            r.kind = KIND_SPARSE; r.i0 = Y_D + i; r.c0 = -1; r.i1 = -1;
            r.gidx = -1; r.xidx = W_DSL(N) + i; r.xsign = -1;
            add_row(s, &r, -y[Y_D + i]);
        }
    }
    /* EE in current set (:304); all-zero padding rows can never be active (b>0, ps>=0) */
    for (int rr = 0; rr < NSET; rr++) {
        double a0 = sc.a_cur[rr], a1 = sc.a_cur[rr + NSET], a2 = sc.a_cur[rr + 2 * NSET];
        if (a0 == 0 && a1 == 0 && a2 == 0 && sc.b_cur[rr] > 0) continue;
        memset(&r, 0, sizeof r);
        r.kind = KIND_POSE;
        r.a[0] = a0; r.a[1] = a1; r.a[2] = a2; r.a[L_PS] = -1;
        r.gidx = rr; r.gsign = 1; r.xidx = -1;
        add_row(s, &r, a0 * pose[0] + a1 * pose[1] + a2 * pose[2] - sc.b_cur[rr] - y[Y_PS]);
    }
    /* orientation bounds (:308-321) */
    {
        double nb[3] = {dot3(sc.br1, sc.br1), dot3(sc.dpn, sc.dpn), dot3(sc.br2, sc.br2)};
        double pr[3] = {pe.proj1, pe.projp, pe.proj2};
        const double* gs[3] = {pe.gsc1, pe.gscp, pe.gsc2};
        for (int m = 0; m < 3; m++) {
            memset(&r, 0, sizeof r);
            r.kind = KIND_POSE;
            for (int c = 0; c < 6; c++) r.a[c] = nb[m] * gs[m][c];
            r.a[L_RS] = -1;
            r.gidx = 15 + m; r.gsign = 1; r.xidx = -1;
            add_row(s, &r, pr[m] - sc.ub[m] - y[Y_RS]);
        }
        for (int m = 0; m < 3; m++) { /* lower: proj - lb + rs >= 0  ->  -(...) <= 0 */
            memset(&r, 0, sizeof r);
            r.kind = KIND_POSE;
            for (int c = 0; c < 6; c++) r.a[c] = -nb[m] * gs[m][c];
            r.a[L_RS] = -1;
            r.gidx = 18 + m; r.gsign = -1; r.xidx = -1;
            add_row(s, &r, -(pr[m] - sc.lb[m] + y[Y_RS]));
        }
    }
    /* collision points (:323-330) */
    for (int i = 0; i < 6; i++) {
        const double* aj = p + P_ASETJ + 45 * i;
        for (int rr = 0; rr < NSET; rr++) {
            double a0 = aj[rr], a1 = aj[rr + NSET], a2 = aj[rr + 2 * NSET];
            double b = p[P_BSETJ + rr * 6 + i];
            if (a0 == 0 && a1 == 0 && a2 == 0 && b + pb->sl0[i] > 0) continue;
            memset(&r, 0, sizeof r);
            r.kind = KIND_PT; r.grp = i;
            r.a[0] = a0; r.a[1] = a1; r.a[2] = a2; r.a[3] = -1;
            r.gidx = 21 + 15 * i + rr; r.gsign = 1; r.xidx = -1;
            add_row(s, &r, a0 * kin.pc[i][0] + a1 * kin.pc[i][1] + a2 * kin.pc[i][2] - b - sl[i]);
        }
    }
    /* phi cap (:332) */
    memset(&r, 0, sizeof r);
    r.kind = KIND_POSE;
    for (int c = 0; c < 3; c++) r.a[c] = sc.dp[c];
    r.gidx = 111; r.gsign = 1; r.xidx = -1;
    add_row(s, &r, pe.phi - (sc.phi_end_seg + 0.005));
    if (term) {
        double z1 = dot3(sc.bp1, pe.ep), z2 = dot3(sc.bp2, pe.ep);
        for (int rr = 0; rr < NSET; rr++) {
            double an[3] = {sc.a_next[rr], sc.a_next[rr + NSET], sc.a_next[rr + 2 * NSET]};
            if (an[0] == 0 && an[1] == 0 && an[2] == 0 && sc.b_next[rr] + pb->sl0[5] > 0) continue;
            double a1 = dot3(an, sc.bp1), a2 = dot3(an, sc.bp2);
            double bnew = sc.b_next[rr] - dot3(an, sc.p_end);
            memset(&r, 0, sizeof r);
            r.kind = KIND_POSE;
            for (int c = 0; c < 3; c++) {
                double tt = 0;
                for (int a = 0; a < 3; a++) tt += (a1 * sc.bp1[a] + a2 * sc.bp2[a]) * pe.Dep[a][c];
                r.a[c] = tt;
            }
            r.a[L_D5] = -1;
            r.gidx = 112 + rr; r.gsign = 1; r.xidx = -1;
            add_row(s, &r, a1 * z1 + a2 * z2 - bnew - sl[5]);
        }
        double cc[3] = {dot3(sc.br1n, sc.br1), dot3(sc.dpnn, sc.dpn), dot3(sc.br2n, sc.br2)};
        double pr[3] = {pe.proj1n, pe.projpn, pe.proj2n};
        const double* gs[3] = {pe.gsc1, pe.gscp, pe.gsc2};
        for (int m = 0; m < 3; m++) {
            memset(&r, 0, sizeof r);
            r.kind = KIND_POSE;
            for (int c = 0; c < 6; c++) r.a[c] = cc[m] * gs[m][c];
            r.a[L_D5] = -1;
            r.gidx = 127 + m; r.gsign = 1; r.xidx = -1;
            add_row(s, &r, pr[m] - sc.ubn[m] - sl[5]);
        }
        for (int m = 0; m < 3; m++) {
            memset(&r, 0, sizeof r);
            r.kind = KIND_POSE;
            for (int c = 0; c < 6; c++) r.a[c] = -cc[m] * gs[m][c];
            r.a[L_D5] = -1;
            r.gidx = 130 + m; r.gsign = -1; r.xidx = -1;
            add_row(s, &r, -(pr[m] - sc.lbn[m] + sl[5]));
        }
    }

    /* ---- dynamics k -> k+1 ---- */
    if (k < N - 1) {
        const stage_t* sn = &pb->st[k + 1];
        if (mode == 0) {
            memcpy(s->A, pb->As, sizeof s->A);
            memcpy(s->B, pb->Bs, sizeof s->B);
            for (int a = 0; a < 3; a++)
                for (int j = 0; j < 7; j++) {
                    double eq = dt * G[3 + a][j], ed = dt * J[3 + a][j];
                    s->A[(Z_PI + a) * NX + Z_Q + j] = eq;
                    s->A[(Z_PI + a) * NX + Z_DQ + j] = ed;
                    s->B[(Z_PI + a) * NU + j] = pb->c3 * eq + pb->c2 * ed;
                }
        }
        for (int i = 0; i < NX; i++) {
            double sm = 0;
            for (int j = 0; j < NX; j++) sm += pb->As[i * NX + j] * s->zeta[j];
            for (int j = 0; j < NU; j++) sm += pb->Bs[i * NU + j] * s->zeta[NX + j];
            s->r[i] = sm - sn->zeta[i];
        }
        for (int a = 0; a < 3; a++) s->r[Z_PI + a] = s->zeta[Z_PI + a] + dt * v[3 + a] - sn->zeta[Z_PI + a];
    }
}

/* chain a group's local (M, b) into the natural-coordinate Hessian/gradient */
static void chain_group(const double* Jg, int nloc, const double* M, const double* b, double* Hy, double* gy) {
    /* Hy += Jg^T M Jg ; gy += Jg^T b ; Jg is nloc x NZ */
    double MJ[NLOC * NZ];
    matmul(M, Jg, MJ, nloc, nloc, NZ);
    for (int i = 0; i < NZ; i++)
        for (int j = 0; j < NZ; j++) {
            double sm = 0;
            for (int l = 0; l < nloc; l++) sm += Jg[l * NZ + i] * MJ[l * NZ + j];
            Hy[i * NZ + j] += sm;
        }
    if (b && gy)
        for (int i = 0; i < NZ; i++) {
            double sm = 0;
            for (int l = 0; l < nloc; l++) sm += Jg[l * NZ + i] * b[l];
            gy[i] += sm;
        }
}

/* Assemble stage Hessian H (zeta coords), effective gradient g (barrier-modified) and dual
 * gradient gdual = grad f + sum z_i a_i, for barrier parameter mu. */
static void assemble_stage(prob_t* pb, int k, double mu) {
    stage_t* s = &pb->st[k];
    const double* wts = pb->p + P_W;
    int term = (k == pb->N - 1);
    double y[NZ];
    zeta_to_y(pb, s->zeta, y);
    double Hy[NZ * NZ], gy[NZ], gz[NZ];
    memset(Hy, 0, sizeof Hy);
    memset(gy, 0, sizeof gy);
    memset(gz, 0, sizeof gz);
    double Mp[NLOC * NLOC], bp[NLOC], bzp[NLOC];
    double Mt[6][16], bt[6][4], bzt[6][4];
    memset(Mp, 0, sizeof Mp); memset(bp, 0, sizeof bp); memset(bzp, 0, sizeof bzp);
    memset(Mt, 0, sizeof Mt); memset(bt, 0, sizeof bt); memset(bzt, 0, sizeof bzt);
    for (int i = 0; i < 12; i++) {
        bp[i] = s->costg[i];
        bzp[i] = s->costg[i];
        for (int j = 0; j < 12; j++) Mp[i * NLOC + j] = s->costH[i * 12 + j];
    }
    if (pb->hess)
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) Mp[i * NLOC + j] += s->costHx[6 * i + j];
    for (int i = 0; i < s->nrows; i++) {
        const row_t* r = &s->rows[i];
        double sg = pb->no_sigma ? 0.0 : s->z[i] / s->t[i];
        double rho = mu / s->t[i] + sg * (s->h[i] + s->t[i]);
        if (r->kind == KIND_POSE) {
            for (int a = 0; a < NLOC; a++) {
                if (r->a[a] == 0) continue;
                bp[a] += rho * r->a[a];
                bzp[a] += s->z[i] * r->a[a];
                for (int b = 0; b < NLOC; b++) Mp[a * NLOC + b] += sg * r->a[a] * r->a[b];
            }
        } else if (r->kind == KIND_PT) {
            for (int a = 0; a < 4; a++) {
                bt[r->grp][a] += rho * r->a[a];
                bzt[r->grp][a] += s->z[i] * r->a[a];
                for (int b = 0; b < 4; b++) Mt[r->grp][a * 4 + b] += sg * r->a[a] * r->a[b];
            }
        } else {
            int id[2] = {r->i0, r->i1};
            double cf[2] = {r->c0, r->c1};
            for (int a = 0; a < 2; a++) {
                if (id[a] < 0) continue;
                gy[id[a]] += rho * cf[a];
                gz[id[a]] += s->z[i] * cf[a];
                for (int b = 0; b < 2; b++)
                    if (id[b] >= 0) Hy[id[a] * NZ + id[b]] += sg * cf[a] * cf[b];
            }
        }
    }
    chain_group(&s->Jpose[0][0], NLOC, Mp, bp, Hy, gy);
    for (int i = 0; i < NZ; i++) { /* gz += Jpose^T bzp */
        double sm = 0;
        for (int l = 0; l < NLOC; l++) sm += s->Jpose[l][i] * bzp[l];
        gz[i] += sm;
    }
    for (int c = 0; c < 6; c++) {
        chain_group(&s->Jpt[c][0][0], 4, Mt[c], bt[c], Hy, gy);
        for (int i = 0; i < NZ; i++) {
            double sm = 0;
            for (int l = 0; l < 4; l++) sm += s->Jpt[c][l][i] * bzt[c][l];
            gz[i] += sm;
        }
    }
    if (pb->hess) {
        /* second-order kinematic terms of the Lagrangian Hessian:
         *   d2 p / dq_i dq_j = z_min x c_max  (c_j = z_j x (p - o_j)),  dz_j/dq_i = z_i x z_j (i<j)
         * weighted by the generalised forces on each point / on v. */
        double Fp[3] = {bzp[0], bzp[1], bzp[2]};
        double Fv[6];
        for (int a = 0; a < 6; a++) Fv[a] = bzp[6 + a];
        for (int a = 0; a < 3; a++) Fv[3 + a] += 0.5 * pb->dt * bzp[3 + a];
        if (k < pb->N - 1)
            for (int a = 0; a < 3; a++) Fv[3 + a] += pb->dt * pb->st[k + 1].lam[Z_PI + a];
        for (int i = 0; i < 7; i++)
            for (int j = i; j < 7; j++) {
                double cj[3] = {s->J[0][j], s->J[1][j], s->J[2][j]}, zc[3], zz[3];
                bmpc_cross(s->kin.z[i], cj, zc);
                double hq = dot3(Fp, zc);
                Hy[(Y_Q + i) * NZ + Y_Q + j] += hq;
                if (j != i) Hy[(Y_Q + j) * NZ + Y_Q + i] += hq;
                /* mixed q-dq terms from v = J(q) dq */
                double hm = dot3(Fv, zc);
                double hm_ij = hm, hm_ji = hm; /* d2 v / dq_i d(dq_j) and d2 v / dq_j d(dq_i) */
                if (j > i) {
                    bmpc_cross(s->kin.z[i], s->kin.z[j], zz);
                    hm_ij += dot3(Fv + 3, zz);
                }
                Hy[(Y_Q + i) * NZ + Y_DQ + j] += hm_ij;
                Hy[(Y_DQ + j) * NZ + Y_Q + i] += hm_ij;
                if (j != i) {
                    Hy[(Y_Q + j) * NZ + Y_DQ + i] += hm_ji;
                    Hy[(Y_DQ + i) * NZ + Y_Q + j] += hm_ji;
                }
            }
        /* q-q block of the v = J(q) dq curvature: sum_j dq_j d2 c_j/dq_a dq_b (linear part) and
         * sum_j dq_j d2 z_j/dq_a dq_b (angular part); dz_i/dq_b = z_b x z_i (b<i),
         * dc_j/dq_b = z_min x c_max. */
        {
            const double* dqv = y + Y_DQ;
            double cc[7][3];
            for (int j = 0; j < 7; j++) for (int a = 0; a < 3; a++) cc[j][a] = s->J[a][j];
            /* symmetric in (a, b) (third derivatives of the kinematics): upper triangle, mirrored */
            for (int a = 0; a < 7; a++)
                for (int b = a; b < 7; b++) {
                    double acc = 0;
                    for (int j = 0; j < 7; j++) {
                        if (dqv[j] == 0.0) continue;
                        int m = a < j ? a : j, M = a < j ? j : a;
                        double t1[3] = {0, 0, 0}, t2[3], dzm[3] = {0, 0, 0}, dcM[3], tmp[3];
                        if (b < m) { bmpc_cross(s->kin.z[b], s->kin.z[m], dzm); bmpc_cross(dzm, cc[M], t1); }
                        int m2 = b < M ? b : M, M2 = b < M ? M : b;
                        bmpc_cross(s->kin.z[m2], cc[M2], dcM);
                        bmpc_cross(s->kin.z[m], dcM, t2);
                        double lin = Fv[0] * (t1[0] + t2[0]) + Fv[1] * (t1[1] + t2[1]) + Fv[2] * (t1[2] + t2[2]);
                        double ang = 0;
                        if (a < j) { /* d/dq_b (z_a x z_j) */
                            double dza[3] = {0, 0, 0}, dzj[3] = {0, 0, 0}, u1[3] = {0, 0, 0}, u2[3] = {0, 0, 0};
                            if (b < a) { bmpc_cross(s->kin.z[b], s->kin.z[a], dza); bmpc_cross(dza, s->kin.z[j], u1); }
                            if (b < j) { bmpc_cross(s->kin.z[b], s->kin.z[j], dzj); bmpc_cross(s->kin.z[a], dzj, u2); }
                            ang = Fv[3] * (u1[0] + u2[0]) + Fv[4] * (u1[1] + u2[1]) + Fv[5] * (u1[2] + u2[2]);
                        }
                        (void)tmp;
                        acc += dqv[j] * (lin + ang);
                    }
                    Hy[(Y_Q + a) * NZ + Y_Q + b] += acc;
                    if (b != a) Hy[(Y_Q + b) * NZ + Y_Q + a] += acc;
                }
        }
        for (int c = 0; c < 6; c++) {
            double Fc[3] = {bzt[c][0], bzt[c][1], bzt[c][2]};
            int nj = BMPC_COL_NJ[c];
            for (int i = 0; i < nj; i++)
                for (int j = i; j < nj; j++) {
                    double cj[3] = {s->Jpt[c][0][Y_Q + j], s->Jpt[c][1][Y_Q + j], s->Jpt[c][2][Y_Q + j]}, zc[3];
                    bmpc_cross(s->kin.z[i], cj, zc);
                    double hq = dot3(Fc, zc);
                    Hy[(Y_Q + i) * NZ + Y_Q + j] += hq;
                    if (j != i) Hy[(Y_Q + j) * NZ + Y_Q + i] += hq;
                }
        }
    }
    for (int i = 0; i < NZ; i++) Hy[i * NZ + i] += pb->hreg;
    /* direct quadratic cost terms */
    for (int j = 2; j <= 4; j++) {
        Hy[(Y_DQ + j) * NZ + Y_DQ + j] += 2 * wts[6];
        gy[Y_DQ + j] += 2 * wts[6] * y[Y_DQ + j];
        gz[Y_DQ + j] += 2 * wts[6] * y[Y_DQ + j];
    }
    for (int j = 0; j < 7; j++) {
        Hy[(Y_U + j) * NZ + Y_U + j] += 2 * wts[7];
        gy[Y_U + j] += 2 * wts[7] * y[Y_U + j];
        gz[Y_U + j] += 2 * wts[7] * y[Y_U + j];
    }
    {
        int idx[4] = {Y_RS, Y_DRS, Y_PS, Y_DPS};
        double ww[4] = {wts[9], wts[10], wts[9], wts[10]};
        for (int m = 0; m < 4; m++) {
            Hy[idx[m] * NZ + idx[m]] += 2 * ww[m];
            gy[idx[m]] += 2 * ww[m] * y[idx[m]];
            gz[idx[m]] += 2 * ww[m] * y[idx[m]];
        }
    }
    if (term)
        for (int i = 0; i < 6; i++) {
            double h = 2 * wts[10] + (i != 4 ? 2 * wts[8] : 0);
            double gg = 2 * wts[10] * y[Y_D + i] + (i != 4 ? 2 * wts[8] * (pb->sl0[i] + y[Y_D + i]) : 0);
            Hy[(Y_D + i) * NZ + Y_D + i] += h;
            gy[Y_D + i] += gg;
            gz[Y_D + i] += gg;
        }
    /* to zeta coordinates: H = T^T Hy T, g = T^T gy */
    double HT[NZ * NZ];
    matmul(Hy, pb->T, HT, NZ, NZ, NZ);
    matTmul(pb->T, HT, s->H, NZ, NZ, NZ);
    for (int i = 0; i < NZ; i++) {
        double a = 0, b = 0;
        for (int l = 0; l < NZ; l++) {
            a += pb->T[l * NZ + i] * gy[l];
            b += pb->T[l * NZ + i] * gz[l];
        }
        s->g[i] = a;
        s->gdual[i] = b;
    }
}

/* Backward Riccati sweep (deviation form).  Returns 0 or -1 on a non-PD control block. */
static int riccati_backward(prob_t* pb, double reg) {
    int N = pb->N;
    for (int k = N - 1; k >= 1; k--) {
        stage_t* s = &pb->st[k];
        double F[NX * NX], Gm[NU * NX], Hm[NU * NU], fx[NX], fw[NU];
        for (int i = 0; i < NX; i++) {
            for (int j = 0; j < NX; j++) F[i * NX + j] = s->H[i * NZ + j];
            fx[i] = s->g[i];
        }
        for (int i = 0; i < NU; i++) {
            for (int j = 0; j < NX; j++) Gm[i * NX + j] = s->H[(NX + i) * NZ + j];
            for (int j = 0; j < NU; j++) Hm[i * NU + j] = s->H[(NX + i) * NZ + NX + j];
            fw[i] = s->g[NX + i];
        }
        if (k < N - 1) {
            const stage_t* sn = &pb->st[k + 1];
            double PA[NX * NX], PB[NX * NU], tmp[NX * NX], vv[NX];
            matmul(sn->P, s->A, PA, NX, NX, NX);
            matmul(sn->P, s->B, PB, NX, NX, NU);
            matTmul(s->A, PA, tmp, NX, NX, NX);
            for (int i = 0; i < NX * NX; i++) F[i] += tmp[i];
            matTmul(s->B, PA, tmp, NU, NX, NX);
            for (int i = 0; i < NU * NX; i++) Gm[i] += tmp[i];
            matTmul(s->B, PB, tmp, NU, NX, NU);
            for (int i = 0; i < NU * NU; i++) Hm[i] += tmp[i];
            for (int i = 0; i < NX; i++) {
                double sm = sn->pv[i];
                for (int j = 0; j < NX; j++) sm += sn->P[i * NX + j] * s->r[j];
                vv[i] = sm;
            }
            for (int i = 0; i < NX; i++) {
                double sm = 0;
                for (int j = 0; j < NX; j++) sm += s->A[j * NX + i] * vv[j];
                fx[i] += sm;
            }
            for (int i = 0; i < NU; i++) {
                double sm = 0;
                for (int j = 0; j < NX; j++) sm += s->B[j * NU + i] * vv[j];
                fw[i] += sm;
            }
        }
        for (int i = 0; i < NU; i++) Hm[i * NU + i] += reg;
        if (chol(Hm, NU)) return -1;
        /* [K kf] = -Hm^-1 [Gm fw] */
        double rhs[NU * (NX + 1)];
        for (int i = 0; i < NU; i++) {
            for (int j = 0; j < NX; j++) rhs[i * (NX + 1) + j] = Gm[i * NX + j];
            rhs[i * (NX + 1) + NX] = fw[i];
        }
        chol_solve(Hm, NU, rhs, NX + 1);
        for (int i = 0; i < NU; i++) {
            for (int j = 0; j < NX; j++) s->K[i * NX + j] = -rhs[i * (NX + 1) + j];
            s->kf[i] = -rhs[i * (NX + 1) + NX];
        }
        /* P = F + Gm^T K, p = fx + Gm^T kf */
        for (int i = 0; i < NX; i++) {
            for (int j = 0; j < NX; j++) {
                double sm = F[i * NX + j];
                for (int l = 0; l < NU; l++) sm += Gm[l * NX + i] * s->K[l * NX + j];
                s->P[i * NX + j] = sm;
            }
            double sm = fx[i];
            for (int l = 0; l < NU; l++) sm += Gm[l * NX + i] * s->kf[l];
            s->pv[i] = sm;
        }
        for (int i = 0; i < NX; i++)
            for (int j = i + 1; j < NX; j++) {
                double a = 0.5 * (s->P[i * NX + j] + s->P[j * NX + i]);
                s->P[i * NX + j] = s->P[j * NX + i] = a;
            }
    }
    return 0;
}

#define NFREE 8 /* rs~_1, ps~_1, d(6) are free at the first stage */

static int riccati_forward(prob_t* pb) {
    int N = pb->N;
    stage_t* s1 = &pb->st[1];
    double dx[NX];
    for (int i = 0; i < 24; i++) dx[i] = pb->r0[i];
    /* free part: P_ff dx_f = -(p_f + P_fc dx_c) */
    double Pff[NFREE * NFREE], rhs[NFREE];
    for (int i = 0; i < NFREE; i++) {
        double sm = s1->pv[24 + i];
        for (int j = 0; j < 24; j++) sm += s1->P[(24 + i) * NX + j] * dx[j];
        rhs[i] = -sm;
        for (int j = 0; j < NFREE; j++) Pff[i * NFREE + j] = s1->P[(24 + i) * NX + 24 + j];
    }
    if (chol(Pff, NFREE)) return -1;
    chol_solve(Pff, NFREE, rhs, 1);
    for (int i = 0; i < NFREE; i++) dx[24 + i] = rhs[i];
    for (int k = 1; k < N; k++) {
        stage_t* s = &pb->st[k];
        for (int i = 0; i < NX; i++) s->dzeta[i] = dx[i];
        for (int i = 0; i < NU; i++) {
            double sm = s->kf[i];
            for (int j = 0; j < NX; j++) sm += s->K[i * NX + j] * dx[j];
            s->dzeta[NX + i] = sm;
        }
        if (k < N - 1) {
            double dn[NX];
            for (int i = 0; i < NX; i++) {
                double sm = s->r[i];
                for (int j = 0; j < NX; j++) sm += s->A[i * NX + j] * dx[j];
                for (int j = 0; j < NU; j++) sm += s->B[i * NU + j] * s->dzeta[NX + j];
                dn[i] = sm;
            }
            memcpy(dx, dn, sizeof dx);
        }
    }
    return 0;
}

/* row-gradient . dy for every row of a stage */
static void row_dirs(const prob_t* pb, stage_t* s, double* adots) {
    double dy[NZ], dloc[NLOC], dpt[6][4];
    zeta_to_y(pb, s->dzeta, dy);
    for (int l = 0; l < NLOC; l++) {
        double sm = 0;
        for (int j = 0; j < NZ; j++) sm += s->Jpose[l][j] * dy[j];
        dloc[l] = sm;
    }
    for (int c = 0; c < 6; c++)
        for (int l = 0; l < 4; l++) {
            double sm = 0;
            for (int j = 0; j < NZ; j++) sm += s->Jpt[c][l][j] * dy[j];
            dpt[c][l] = sm;
        }
    for (int i = 0; i < s->nrows; i++) {
        const row_t* r = &s->rows[i];
        double sm = 0;
        if (r->kind == KIND_POSE)
            for (int a = 0; a < NLOC; a++) sm += r->a[a] * dloc[a];
        else if (r->kind == KIND_PT)
            for (int a = 0; a < 4; a++) sm += r->a[a] * dpt[r->grp][a];
        else {
            sm = r->c0 * dy[r->i0];
            if (r->i1 >= 0) sm += r->c1 * dy[r->i1];
        }
        adots[i] = sm;
    }
}

typedef struct {
    double err, dual, prim, compl, compl_mu, sd, sc, f, theta, avg_compl, min_compl, barrier_logsum;
    int nrows_total;
} kkt_t;

/* adjoint multipliers + IPOPT-style scaled optimality error at the current iterate */
static void kkt_error(prob_t* pb, kkt_t* kk, double mu) {
    int N = pb->N;
    double compl_mu = 0;
    double dual = 0, prim = 0, compl = 0, sum_lam = 0, sum_z = 0, f = 0, theta = 0, sumc = 0, minc = 1e300;
    int nrows = 0, neq = 0;
    double lam_next[NX];
    memset(lam_next, 0, sizeof lam_next);
    for (int k = N - 1; k >= 1; k--) {
        stage_t* s = &pb->st[k];
        double gl[NZ];
        memcpy(gl, s->gdual, sizeof gl);
        if (k < N - 1) {
            for (int i = 0; i < NX; i++) {
                double sm = 0;
                for (int j = 0; j < NX; j++) sm += s->A[j * NX + i] * lam_next[j];
                gl[i] += sm;
            }
            for (int i = 0; i < NU; i++) {
                double sm = 0;
                for (int j = 0; j < NX; j++) sm += s->B[j * NU + i] * lam_next[j];
                gl[NX + i] += sm;
            }
            for (int i = 0; i < NX; i++) {
                prim = fmax(prim, fabs(s->r[i]));
                theta += fabs(s->r[i]);
            }
            neq += NX;
        }
        for (int i = 0; i < NU; i++) dual = fmax(dual, fabs(gl[NX + i]));
        if (k == 1)
            for (int i = 24; i < NX; i++) dual = fmax(dual, fabs(gl[i]));
        for (int i = 0; i < NX; i++) {
            s->lam[i] = gl[i];
            lam_next[i] = gl[i];
            sum_lam += fabs(gl[i]);
        }
        for (int i = 0; i < s->nrows; i++) {
            double c = s->t[i] * s->z[i];
            compl = fmax(compl, c);
            compl_mu = fmax(compl_mu, fabs(c - mu));
            sumc += c;
            minc = fmin(minc, c);
            sum_z += s->z[i];
            prim = fmax(prim, fabs(s->h[i] + s->t[i]));
            theta += fabs(s->h[i] + s->t[i]);
        }
        nrows += s->nrows;
        f += s->fval;
    }
    for (int i = 0; i < 24; i++) {
        prim = fmax(prim, fabs(pb->r0[i]));
        theta += fabs(pb->r0[i]);
    }
    neq += 24;
    double smax = 100.0;
    kk->sd = fmax(smax, (sum_lam + sum_z) / (double)(neq + nrows)) / smax;
    kk->sc = fmax(smax, sum_z / (double)nrows) / smax;
    kk->dual = dual; kk->prim = prim; kk->compl = compl; kk->compl_mu = compl_mu;
    kk->err = fmax(fmax(dual / kk->sd, prim), compl / kk->sc);
    kk->f = f; kk->theta = theta;
    kk->avg_compl = sumc / nrows; kk->min_compl = minc;
    kk->nrows_total = nrows;
}

static double merit_parts(prob_t* pb, double* f, double* theta, double* logsum) {
    double ff = 0, th = 0, ls = 0;
    for (int k = 1; k < pb->N; k++) {
        stage_t* s = &pb->st[k];
        ff += s->fval;
        if (k < pb->N - 1)
            for (int i = 0; i < NX; i++) th += fabs(s->r[i]);
        for (int i = 0; i < s->nrows; i++) {
            th += fabs(s->h[i] + s->t[i]);
            ls += log(s->t[i]);
        }
    }
    for (int i = 0; i < 24; i++) th += fabs(pb->r0[i]);
    *f = ff; *theta = th; *logsum = ls;
    return 0;
}

static void setup_problem(const bmpc_oracle_opts* o, prob_t* pbp, const double* x0, const double* lbx,
                          const double* ubx, const double* p, double* pins) {
    int N = o->N;
    double dt = o->dt;
#define pb (*pbp)
    memset(&pb, 0, sizeof pb);
    pb.N = N; pb.dt = dt; pb.p = p;
    pb.c1 = dt / 2; pb.c2 = dt * dt / 6; pb.c3 = dt * dt * dt / 24;
    pb.b1 = dt; pb.b2 = dt * dt; pb.b3 = 7 * dt * dt * dt / 12;
    build_T(&pb);
    pb.hess = (o->hess == 2) ? 0 : o->hess;
    pb.hreg = 0.0;
    pb.st = (stage_t*)calloc(N, sizeof(stage_t));
    pb.lbq = (double*)malloc(sizeof(double) * N * 28);
    pb.ubq = (double*)malloc(sizeof(double) * N * 28);
    for (int i = 0; i < 6; i++) pb.sl0[i] = p[P_SLACKS0 + i];
    /* box bounds per stage, natural order q,dq,ddq,u */
    for (int k = 0; k < N; k++)
        for (int b = 0; b < 4; b++)
            for (int j = 0; j < 7; j++) {
                int wi = b * 7 * N + j * N + k;
                pb.lbq[k * 28 + b * 7 + j] = lbx[wi] <= -BIG ? -1e300 : lbx[wi];
                pb.ubq[k * 28 + b * 7 + j] = ubx[wi] >= BIG ? 1e300 : ubx[wi];
            }
    /* stage-0 pins (BoundMPC.py:551-556, 575-580): lbx == ubx */
    double q0[7], dq0[7], ddq0[7], u0[7], p0[6], v0[6];
    for (int j = 0; j < 7; j++) {
        q0[j] = lbx[W_Q(N) + j * N];
        dq0[j] = lbx[W_DQ(N) + j * N];
        ddq0[j] = lbx[W_DDQ(N) + j * N];
        u0[j] = lbx[W_U(N) + j * N];
    }
    for (int c = 0; c < 6; c++) {
        p0[c] = lbx[W_P(N) + c * N];
        v0[c] = lbx[W_V(N) + c * N];
    }
    for (int c = 0; c < 3; c++) pb.iw0[c] = p0[3 + c];
    /* required x~_1 = A x_0 + B0 u_0 (natural x_1 minus B1 u_1) and pi_1 = p_rot_0 + dt/2 w_0 */
    for (int j = 0; j < 7; j++) {
        pb.x1fix[Z_Q + j] = q0[j] + dt * dq0[j] + dt * dt / 2 * ddq0[j] + dt * dt * dt / 8 * u0[j];
        pb.x1fix[Z_DQ + j] = dq0[j] + dt * ddq0[j] + dt * dt / 3 * u0[j];
        pb.x1fix[Z_DDQ + j] = ddq0[j] + dt / 2 * u0[j];
    }
    for (int c = 0; c < 3; c++) pb.x1fix[Z_PI + c] = p0[3 + c] + dt / 2 * v0[3 + c];

    /* ---- initial iterate from x0 ---- */
    for (int k = 1; k < N; k++) {
        stage_t* s = &pb.st[k];
        double qk[7], dqk[7];
        for (int j = 0; j < 7; j++) {
            double uu = x0[W_U(N) + j * N + k];
            qk[j] = x0[W_Q(N) + j * N + k];
            dqk[j] = x0[W_DQ(N) + j * N + k];
            s->zeta[Z_Q + j] = qk[j] - pb.c3 * uu;
            s->zeta[Z_DQ + j] = dqk[j] - pb.c2 * uu;
            s->zeta[Z_DDQ + j] = x0[W_DDQ(N) + j * N + k] - pb.c1 * uu;
            s->zeta[Z_U + j] = uu;
        }
        bmpc_kin kin;
        double J[6][7];
        bmpc_kin_eval(qk, &kin);
        bmpc_kin_jac(&kin, J);
        for (int c = 0; c < 3; c++) {
            double om = 0;
            for (int j = 0; j < 7; j++) om += J[3 + c][j] * dqk[j];
            s->zeta[Z_PI + c] = x0[W_P(N) + (3 + c) * N + k] - dt / 2 * om;
        }
        double rs = x0[W_RS(N) + k], drs = x0[W_DRS(N) + k], ps = x0[W_PS(N) + k], dps = x0[W_DPS(N) + k];
        s->zeta[Z_RS] = rs - dt / 2 * drs;
        s->zeta[Z_PS] = ps - dt / 2 * dps;
        s->zeta[Z_DRS] = drs;
        s->zeta[Z_DPS] = dps;
        for (int i = 0; i < 6; i++) s->zeta[Z_D + i] = x0[W_DSL(N) + i];
    }
    for (int i = 0; i < 24; i++) pb.r0[i] = pb.x1fix[i] - pb.st[1].zeta[i];

    for (int j = 0; j < 7; j++) {
        pins[j] = q0[j]; pins[7 + j] = dq0[j]; pins[14 + j] = ddq0[j]; pins[21 + j] = u0[j];
    }
    for (int c = 0; c < 6; c++) { pins[28 + c] = p0[c]; pins[34 + c] = v0[c]; }
#undef pb
}

/* Multipliers of the FULL-SPACE NLP in CasADi's convention (grad f + J_g^T lam_g + lam_x = 0; lam > 0 at an active
 * upper bound, < 0 at an active lower bound; BoundMPC.py:638-645 reads sol["lam_g"], sol["lam_x"]) from the
 * interior-point iterate:
 *  - inequality rows of g and bound rows of x: the row multipliers z of the barrier method (sign by row type);
 *  - the 35 equality rows of block k (new_{k+1} - var_{k+1} = 0, casadi_ocp_formulation.py:145-164) have a -1 on
 *    exactly one variable of stage k+1 (q, dq, ddq, p, v, rslack, pslack): stationarity in that variable gives the
 *    multiplier, backwards over the stages (p before v before q, dq, ddq inside a stage: v_new - v and p_new - p couple
 *    them) -- the adjoint sweep in the full space, done here with the pinned dense Jacobian;
 *  - stage-0 variables (pinned by lbx == ubx, or the eliminated stage-0 slacks): lam_x = -(grad f + J_g^T lam_g),
 *    what IPOPT reports for fixed variables under fixed_variable_treatment=make_parameter.
 * What is left of the stationarity residual sits on u_k, drslack_k, dpslack_k (k >= 1) and dslacks: the dual
 * infeasibility the solver terminated with. */
static void recover_multipliers(prob_t* pb, const double* x, double* lam_g, double* lam_x) {
    const int N = pb->N, n_w = 44 * N + 6, n_g = 147 * (N - 1) + 21;
    double* J = (double*)malloc(sizeof(double) * (size_t)n_g * n_w);
    double* gr = (double*)malloc(sizeof(double) * n_w);
    double* res = (double*)malloc(sizeof(double) * n_w);
    bmpc_oracle_eval(N, pb->dt, x, pb->p, NULL, NULL, gr, J);
    memset(lam_g, 0, sizeof(double) * n_g);
    memset(lam_x, 0, sizeof(double) * n_w);
    for (int k = 1; k < N; k++) {
        const stage_t* s = &pb->st[k];
        const int g0 = 35 * (N - 1) + 112 * (k - 1);
        for (int i = 0; i < s->nrows; i++) {
            const row_t* r = &s->rows[i];
            if (r->gidx >= 0) lam_g[g0 + r->gidx] = r->gsign * s->z[i];
            if (r->xidx >= 0) lam_x[r->xidx] += r->xsign * s->z[i];
        }
    }
    /* residual with the known multipliers */
    for (int i = 0; i < n_w; i++) res[i] = gr[i] + lam_x[i];
    for (int r = 35 * (N - 1); r < n_g; r++)
        if (lam_g[r] != 0.0)
            for (int i = 0; i < n_w; i++) res[i] += J[(size_t)r * n_w + i] * lam_g[r];
    /* equality rows of block k = j - 1: offsets inside the block and the pivot variable of stage j */
    for (int j = N - 1; j >= 1; j--) {
        const int blk = 35 * (j - 1);
        /* order: p (21..26), v (27..32), rslack 33, pslack 34, then q, dq, ddq (0..20) */
        int order[35], piv[35], n = 0;
        for (int c = 0; c < 6; c++) { order[n] = 21 + c; piv[n++] = W_P(N) + c * N + j; }
        for (int c = 0; c < 6; c++) { order[n] = 27 + c; piv[n++] = W_V(N) + c * N + j; }
        order[n] = 33; piv[n++] = W_RS(N) + j;
        order[n] = 34; piv[n++] = W_PS(N) + j;
        for (int c = 0; c < 7; c++) { order[n] = c; piv[n++] = W_Q(N) + c * N + j; }
        for (int c = 0; c < 7; c++) { order[n] = 7 + c; piv[n++] = W_DQ(N) + c * N + j; }
        for (int c = 0; c < 7; c++) { order[n] = 14 + c; piv[n++] = W_DDQ(N) + c * N + j; }
        for (int m = 0; m < 35; m++) {
            const int row = blk + order[m];
            const double* Jr = J + (size_t)row * n_w;
            /* J[row][piv] = -1 (var_{k+1} enters its own defect only there) */
            const double lam = res[piv[m]] / (-Jr[piv[m]]);
            lam_g[row] = lam;
            for (int i = 0; i < n_w; i++)
                if (Jr[i] != 0.0) res[i] += Jr[i] * lam;
        }
    }
    /* stage-0 variables: multipliers of the fixed variables from stationarity */
    for (int f = 0; f < 40; f++) { lam_x[f * N] -= res[f * N]; }
    lam_x[W_RS(N)] -= res[W_RS(N)]; lam_x[W_DRS(N)] -= res[W_DRS(N)];
    lam_x[W_PS(N)] -= res[W_PS(N)]; lam_x[W_DPS(N)] -= res[W_DPS(N)];
    free(J); free(gr); free(res);
}

/* Decisions of the LAST iteration a solve took (tests/test_iterate_parity.py compares them with the HIP path's InstState, run by
 * run with max_iter = 1, 2, 3, ...): when non-NULL, bmpc_oracle_solve leaves
 *   {iterations, status, mu, alpha (1e300: no acceptable step), alpha_dual, alpha fraction-to-boundary, delta_w, exact Hessian
 *    wanted next, factorisation retries, backtracks, KKT error of the previous iterate, stall counter}   (BMPC_ORACLE_INFO = 12)
 * here; set per thread by bmpc_oracle_solve_batch_info. */
static _Thread_local double* g_info = NULL;

int bmpc_oracle_solve(const bmpc_oracle_opts* o, const double* x0, const double* lbx,
                      const double* ubx, const double* p, double* x, double* g, double* lam_g,
                      double* lam_x, double* f, int* iters, int* status, double* viol) {
    int N = o->N;
    int n_w = 44 * N + 6, n_g = 147 * (N - 1) + 21;
    double dt = o->dt;
    prob_t pb;
    double pins[40];
    setup_problem(o, &pb, x0, lbx, ubx, p, pins);
    const double *q0 = pins, *dq0 = pins + 7, *ddq0 = pins + 14, *u0 = pins + 21, *p0 = pins + 28, *v0 = pins + 34;
    /* ---- row slacks / multipliers ---- */
    const double t_push = 1e-2, z_init = 1.0;
    for (int k = N - 1; k >= 1; k--) eval_stage(&pb, k, 0);
    for (int k = 1; k < N; k++) {
        stage_t* s = &pb.st[k];
        for (int i = 0; i < s->nrows; i++) {
            s->t[i] = fmax(-s->h[i], t_push);
            s->z[i] = z_init;
        }
    }

    int st = 1, it = 0;
    double mu = o->mu_init, nu = 1.0;
#define MAXFILT 8
    double filt_th[MAXFILT], filt_phi[MAXFILT], filt_mu = -1, theta_max = 1e300, theta_min = 0;
    int nfilt = 0;
    kkt_t kk;
    memset(&kk, 0, sizeof kk);
    double reg = 1e-9, err_prev = 1e300, dw_last = 0.0, err_best = 1e300;
    int gn_skip = 0, gn_back = 0;
    double alpha_last = 1e300;
    int stall = 0;                   /* iterations since the optimality error last improved (by 10 %) */
    double info_ad = 0, info_ap = 0, info_dw = 0;      /* last iteration's decisions (g_info) */
    int info_tries = 0, info_bt = 0;
    for (it = 0;; it++) {
        /* gdual needs current z: assemble with a provisional mu (only H,g depend on mu) */
        for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
        kkt_error(&pb, &kk, mu);
        /* the second-order term of the pi dynamics in the exact Hessian is weighted by the adjoint multipliers lam_{k+1} that
         * kkt_error has just computed AT THIS ITERATE (the HIP kernels take them from the same backward sweep); the assembly above
         * still saw those of the previous iterate (round 4: found by the iterate-for-iterate comparison -- after the first
         * exact-Hessian iteration the two sides' steps differed by 1e-3 relative) */
        if (pb.hess)
            for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
        if (o->verbose)
            printf("it %3d f %.6e err %.2e (d %.2e p %.2e c %.2e) mu %.2e\n", it, kk.f, kk.err, kk.dual, kk.prim, kk.compl, mu);
        if (kk.err <= o->tol && kk.dual <= 1.0 && kk.prim <= 1e-4 && kk.compl <= 1e-4) { st = 0; break; }
        if (it >= o->max_iter) { st = 1; break; }
        if (o->mu_strategy == 0) {
            /* LOQO barrier update (IPOPT mu_oracle=loqo) */
            double xi = kk.min_compl / kk.avg_compl;
            double fac = 0.05 * (1 - xi) / xi;
            double sg = 0.1 * pow(fmin(fac, 2.0), 3);
            double mu_new = sg * kk.avg_compl;
            mu_new = fmax(mu_new, o->tol / 100.0);
            mu_new = fmin(mu_new, 1e3);
            if (mu_new != mu) {
                mu = mu_new;
                for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
            }
        } else if (o->mu_strategy == 1) {
            /* monotone Fiacco-McCormick update (IPOPT mu_strategy=monotone, eq. (7)) */
            double emu = fmax(fmax(kk.dual / kk.sd, kk.prim), kk.compl_mu / kk.sc);
            int changed = 0;
            while (emu <= o->kappa_eps * mu && mu > o->tol / 10.0) {
                double mu_before = mu;
                mu = fmax(o->tol / 10.0, fmin(o->kappa_mu * mu, pow(mu, o->theta_mu)));
                /* not below the optimality error it was released at (scaled): a barrier parameter far below the error
                 * jams the iterates against rows they have yet to identify as active */
                if (o->mu_floor_k > 0) {
                    double m2 = fmax(mu, fmin(emu / o->mu_floor_k, 0.1));
                    if (m2 >= mu_before) { mu = mu_before; break; }
                    mu = m2;
                }
                changed = 1;
                emu = fmax(fmax(kk.dual / kk.sd, kk.prim), fmax(kk.compl - mu, 0) / kk.sc);
            }
            if (changed)
                for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
        }
        int tries = 0;
        if (o->hess == 2) {
            /* hybrid: Gauss-Newton far from the solution, second-order kinematic terms once the
             * optimality error is small; fall back to Gauss-Newton when that is not convex */
            /* the exact Hessian close to a solution, or when the Gauss-Newton model has stopped making progress */
            int want = (err_prev < o->hess_switch) || (o->inertia == 2 && stall >= o->stall_n);   /* (error of the PREVIOUS iterate: fused-sweep friendly) */
            /* after a Gauss-Newton fallback the exact Hessian is not tried again for 1, 2, ... gn_backoff iterations (a
             * failed attempt costs a backward sweep); an exact step that goes through resets the back-off */
            if (o->gn_backoff > 0 && want && gn_skip > 0) { want = 0; gn_skip--; }
            if (want != pb.hess) {
                pb.hess = want;
                for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
            }
        }
        if (o->mu_strategy >= 2) {
            /* probing (Mehrotra) choice of mu from the affine-scaling direction (2: free, 3: never increasing); measured on
             * configs[2]: no fewer iterations than the monotone schedule, kept for reference */
            for (int k = 1; k < N; k++) assemble_stage(&pb, k, 0.0);
            if (!(riccati_backward(&pb, reg) || riccati_forward(&pb))) {
                double apa = 1, ada = 1;
                for (int k = 1; k < N; k++) {
                    stage_t* s = &pb.st[k];
                    double ad_[MAXROWS];
                    row_dirs(&pb, s, ad_);
                    for (int i = 0; i < s->nrows; i++) {
                        double dti = -(s->h[i] + s->t[i]) - ad_[i];
                        double dzi = (0.0 - s->t[i] * s->z[i] - s->z[i] * dti) / s->t[i];
                        s->dt_[i] = dti; s->dz_[i] = dzi;
                        if (dti < 0) apa = fmin(apa, -s->t[i] / dti);
                        if (dzi < 0) ada = fmin(ada, -s->z[i] / dzi);
                    }
                }
                double sm = 0; int nr = 0;
                for (int k = 1; k < N; k++) {
                    stage_t* s = &pb.st[k];
                    for (int i = 0; i < s->nrows; i++) { sm += (s->t[i] + apa * s->dt_[i]) * (s->z[i] + ada * s->dz_[i]); nr++; }
                }
                double mu_aff = sm / nr, sg = pow(mu_aff / kk.avg_compl, 3.0);
                double mu_new = fmin(fmax(sg * kk.avg_compl, o->tol / 10.0), 1e3);
                if (o->mu_strategy == 3) mu_new = fmin(mu_new, mu);      /* never increase */
                if (o->verbose > 1) printf("      probing: alpha_aff %.3g %.3g mu_aff %.2e avg %.2e sigma %.2e -> mu %.2e\n", apa, ada, mu_aff, kk.avg_compl, sg, mu_new);
                mu = mu_new;
            }
            for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
        }
        /* inertia correction (Waechter & Biegler 2006, Algorithm IC): the Riccati recursion's control blocks are positive
         * definite iff the Hessian is positive definite on the null space of the dynamics; when they are not, delta_w I is
         * added to the Hessian and escalated (first trial max(dw_min, dw_last / 3) or dw_0 = 1e-4, then x 100 while there is
         * no history, x 8 afterwards) */
        {
            double dw = 0.0;                 /* (pb.hreg is 0 here: reset at the end of every iteration) */
            while (riccati_backward(&pb, reg) || riccati_forward(&pb)) {
                /* far from a solution the Gauss-Newton model gives the better step (measured); close to one,
                 * or when the error has not improved for stall_n iterations, the corrected exact Hessian */
                int gn_fallback = o->inertia == 0 || (o->inertia == 2 && err_prev > o->inertia_err && stall < o->stall_n);
                if (gn_fallback && o->hess == 2 && pb.hess == 1) {
                    pb.hess = 0;
                    for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
                    tries++;
                    if (o->gn_backoff > 0) { gn_back = gn_back ? (2 * gn_back < o->gn_backoff ? 2 * gn_back : o->gn_backoff) : 1; gn_skip = gn_back; }
                    continue;
                }
                if (dw == 0.0) dw = (dw_last == 0.0) ? o->dw0 : fmax(1e-20, dw_last / 3.0);
                else dw *= (dw_last == 0.0) ? 100.0 : 8.0;
                if (++tries > 14 || dw > 1e20) { st = 3; goto done; }
                pb.hreg = dw;
                for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
            }
            if (dw > 0.0) dw_last = dw;
            if (tries == 0 && pb.hess == 1) gn_back = 0;
            bmpc_dbg_iters++; bmpc_dbg_sweeps += 1 + tries; bmpc_dbg_retry_iters += tries > 0;
        }
        /* row steps + fraction to boundary */
        double tau = fmax(0.99, 1 - mu), ap = 1, ad = 1, dphi_f = 0, dphi_bar = 0;
        int lim_k = -1, lim_i = -1;
for (int k = 1; k < N; k++) {
            stage_t* s = &pb.st[k];
            double ad_[MAXROWS];
            row_dirs(&pb, s, ad_);
            for (int i = 0; i < s->nrows; i++) {
                double dti = -(s->h[i] + s->t[i]) - ad_[i];
                double dzi = (mu - s->t[i] * s->z[i] - s->z[i] * dti) / s->t[i];
                s->dt_[i] = dti; s->dz_[i] = dzi;
                if (dti < 0 && -tau * s->t[i] / dti < ap) { ap = -tau * s->t[i] / dti; lim_k = k; lim_i = i; }
                if (dzi < 0) ad = fmin(ad, -tau * s->z[i] / dzi);
                dphi_bar -= mu * dti / s->t[i];
            }
        }
        /* directional derivative of f along dzeta via gdual - sum z a  ==  use finite pieces:
         * grad_f . d = (gdual . d) - sum_i z_i (a_i . d) */
        for (int k = 1; k < N; k++) {
            stage_t* s = &pb.st[k];
            double ad_[MAXROWS];
            row_dirs(&pb, s, ad_);
            double sm = 0;
            for (int i = 0; i < NZ; i++) sm += s->gdual[i] * s->dzeta[i];
            for (int i = 0; i < s->nrows; i++) sm -= s->z[i] * ad_[i];
            dphi_f += sm;
        }
        double f0, th0, ls0;
        merit_parts(&pb, &f0, &th0, &ls0);
        double dlin = dphi_f + dphi_bar;       /* directional derivative of the barrier objective */
        double phi0 = f0 - mu * ls0;
        double D = dlin;
        if (it == 0) { theta_max = 1e4 * fmax(1.0, th0); theta_min = 1e-4 * fmax(1.0, th0); }
        if (mu != filt_mu) { nfilt = 0; filt_mu = mu; }   /* the barrier objective changed */
        /* filter line search (Waechter & Biegler 2006, Sec. 2.3, without SOC/restoration) */
        double alpha = ap;
        /* the line search starts at ls_alpha_mem times the step length the previous iteration ended with (capped by the
         * fraction-to-boundary length): an iterate that needed six halvings is unlikely to take a full step next time, and every
         * rejected trial is an evaluation pass (a super-step on the GPU).  0: always from the fraction-to-boundary length (IPOPT) */
        if (o->ls_alpha_mem > 0 && it > 0) alpha = fmin(ap, o->ls_alpha_mem * alpha_last);
        int ls_ok = 0, armijo_case = 0, ls_bt = 0;   /* ls_bt: rejected trials (a failed search keeps its tenth trial: 9) */
        const int soc_on = o->soc, soc_after = o->soc_after;
        double* save = (double*)malloc(sizeof(double) * N * (NZ + MAXROWS));
        for (int k = 1; k < N; k++) {
            memcpy(save + k * (NZ + MAXROWS), pb.st[k].zeta, sizeof(double) * NZ);
            memcpy(save + k * (NZ + MAXROWS) + NZ, pb.st[k].t, sizeof(double) * MAXROWS);
        }
        for (int bt = 0; bt < 10; bt++) {
            ls_bt = bt;
            bmpc_dbg_trials++;
            for (int k = 1; k < N; k++) {
                stage_t* s = &pb.st[k];
                const double* sv = save + k * (NZ + MAXROWS);
                for (int i = 0; i < NZ; i++) s->zeta[i] = sv[i] + alpha * s->dzeta[i];
                for (int i = 0; i < s->nrows; i++) s->t[i] = sv[NZ + i] + alpha * s->dt_[i];
            }
            if (o->pi_shoot) {
                /* the rotation-integral state follows its (nonlinear) dynamics exactly at every trial point: pi_{k+1} = pi_k + dt w(q_k, dq_k) */
                for (int k = 1; k < N - 1; k++) {
                    stage_t* s = &pb.st[k];
                    double y[NZ], J[6][7];
                    bmpc_kin kin;
                    zeta_to_y(&pb, s->zeta, y);
                    bmpc_kin_eval(y + Y_Q, &kin);
                    bmpc_kin_jac(&kin, J);
                    for (int a = 0; a < 3; a++) {
                        double om = 0;
                        for (int j = 0; j < 7; j++) om += J[3 + a][j] * y[Y_DQ + j];
                        pb.st[k + 1].zeta[Z_PI + a] = s->zeta[Z_PI + a] + dt * om;
                    }
                }
            }
            for (int i = 0; i < 24; i++) pb.r0[i] = pb.x1fix[i] - pb.st[1].zeta[i];
            for (int k = N - 1; k >= 1; k--) eval_stage(&pb, k, 1);
            /* slack reset (Byrd, Hribar & Nocedal's interior-point method; KNITRO): a trial slack is never smaller than
             * the value that closes its row at the trial point, t <- max(t + alpha dt, -h(x + alpha dx)) -- it lowers the
             * infeasibility theta and the barrier term, and keeps the nonlinearity of the kinematic rows out of theta */
            if (o->slack_reset)
                for (int k = 1; k < N; k++) {
                    stage_t* s = &pb.st[k];
                    for (int i = 0; i < s->nrows; i++)
                        if (-s->h[i] > s->t[i]) s->t[i] = -s->h[i];
                }
            double f1, th1, ls1;
            merit_parts(&pb, &f1, &th1, &ls1);
            double phi1 = f1 - mu * ls1;
            int ok = (th1 <= theta_max);
            for (int j = 0; ok && j < nfilt; j++)
                if (th1 >= filt_th[j] && phi1 >= filt_phi[j]) ok = 0;
            if (ok) {
                int sw = (th0 <= theta_min) && (D < 0) && (alpha * pow(-D, 2.3) > pow(th0, 1.1));
                if (sw) {
                    ok = (phi1 <= phi0 + 1e-4 * alpha * D + 1e-12 * fabs(phi0));
                    armijo_case = ok;
                } else {
                    ok = (th1 <= (1 - 1e-5) * th0) || (phi1 <= phi0 - 1e-5 * th0);
                }
            }
            if (ok) { ls_ok = 1; break; }
            if (bt == 0 && soc_on && th1 >= th0 && it >= soc_after) {
                /* second-order correction (Waechter & Biegler 2006, Sec. 2.4 / Algorithm A-5.7-5.9) */
                const int W = NZ + 2 * MAXROWS;
                double* dsave = (double*)malloc(sizeof(double) * N * W);
                double* cs = (double*)malloc(sizeof(double) * N * (MAXROWS + NX));
                double r0s[24], ad_save = ad;
                for (int k = 1; k < N; k++) {
                    stage_t* s = &pb.st[k];
                    memcpy(dsave + k * W, s->dzeta, sizeof(double) * NZ);
                    memcpy(dsave + k * W + NZ, s->dt_, sizeof(double) * MAXROWS);
                    memcpy(dsave + k * W + NZ + MAXROWS, s->dz_, sizeof(double) * MAXROWS);
                }
                /* c(x_k): re-linearise at x_k */
                #define RESTORE_AND_LIN() do { \
                    for (int k = 1; k < N; k++) { stage_t* s = &pb.st[k]; const double* sv = save + k * (NZ + MAXROWS); \
                        memcpy(s->zeta, sv, sizeof(double) * NZ); memcpy(s->t, sv + NZ, sizeof(double) * MAXROWS); } \
                    for (int i = 0; i < 24; i++) pb.r0[i] = pb.x1fix[i] - pb.st[1].zeta[i]; \
                    for (int k = N - 1; k >= 1; k--) eval_stage(&pb, k, 0); } while (0)
                /* trial residuals first (the state is at the trial point now) */
                for (int k = 1; k < N; k++) {
                    stage_t* s = &pb.st[k];
                    for (int i = 0; i < s->nrows; i++) cs[k * (MAXROWS + NX) + i] = s->h[i] + s->t[i];
                    for (int i = 0; i < NX; i++) cs[k * (MAXROWS + NX) + MAXROWS + i] = (k < N - 1) ? s->r[i] : 0.0;
                }
                for (int i = 0; i < 24; i++) r0s[i] = pb.r0[i];
                RESTORE_AND_LIN();
                for (int k = 1; k < N; k++) {
                    stage_t* s = &pb.st[k];
                    for (int i = 0; i < s->nrows; i++) cs[k * (MAXROWS + NX) + i] += alpha * (s->h[i] + s->t[i]);
                    for (int i = 0; i < NX; i++) cs[k * (MAXROWS + NX) + MAXROWS + i] += (k < N - 1) ? alpha * s->r[i] : 0.0;
                }
                for (int i = 0; i < 24; i++) r0s[i] += alpha * pb.r0[i];
                double th_prev = th0, a_soc = alpha;
                int accepted = 0;
                bmpc_dbg_soc_try++;
                for (int ps = 0; ps < soc_on; ps++) {
                    bmpc_dbg_soc_pass++;
                    /* inject c_soc as the residuals of the linearisation at x_k */
                    for (int k = 1; k < N; k++) {
                        stage_t* s = &pb.st[k];
                        for (int i = 0; i < s->nrows; i++) s->h[i] = cs[k * (MAXROWS + NX) + i] - s->t[i];
                        if (k < N - 1) for (int i = 0; i < NX; i++) s->r[i] = cs[k * (MAXROWS + NX) + MAXROWS + i];
                    }
                    for (int i = 0; i < 24; i++) pb.r0[i] = r0s[i];
                    for (int k = 1; k < N; k++) assemble_stage(&pb, k, mu);
                    if (riccati_backward(&pb, reg) || riccati_forward(&pb)) break;
                    double aps = 1, ads = 1;
                    for (int k = 1; k < N; k++) {
                        stage_t* s = &pb.st[k];
                        double ad_[MAXROWS];
                        row_dirs(&pb, s, ad_);
                        for (int i = 0; i < s->nrows; i++) {
                            double dti = -(s->h[i] + s->t[i]) - ad_[i];
                            double dzi = (mu - s->t[i] * s->z[i] - s->z[i] * dti) / s->t[i];
                            s->dt_[i] = dti; s->dz_[i] = dzi;
                            if (dti < 0) aps = fmin(aps, -tau * s->t[i] / dti);
                            if (dzi < 0) ads = fmin(ads, -tau * s->z[i] / dzi);
                        }
                    }
                    a_soc = aps;
                    for (int k = 1; k < N; k++) {
                        stage_t* s = &pb.st[k];
                        const double* sv = save + k * (NZ + MAXROWS);
                        for (int i = 0; i < NZ; i++) s->zeta[i] = sv[i] + a_soc * s->dzeta[i];
                        for (int i = 0; i < s->nrows; i++) s->t[i] = sv[NZ + i] + a_soc * s->dt_[i];
                    }
                    for (int i = 0; i < 24; i++) pb.r0[i] = pb.x1fix[i] - pb.st[1].zeta[i];
                    for (int k = N - 1; k >= 1; k--) eval_stage(&pb, k, 1);
                    if (o->slack_reset)
                        for (int k = 1; k < N; k++) {
                            stage_t* s = &pb.st[k];
                            for (int i = 0; i < s->nrows; i++)
                                if (-s->h[i] > s->t[i]) s->t[i] = -s->h[i];
                        }
                    double f2, th2, ls2;
                    merit_parts(&pb, &f2, &th2, &ls2);
                    double phi2 = f2 - mu * ls2;
                    int ok2 = (th2 <= theta_max);
                    for (int j = 0; ok2 && j < nfilt; j++)
                        if (th2 >= filt_th[j] && phi2 >= filt_phi[j]) ok2 = 0;
                    if (ok2) {
                        int sw = (th0 <= theta_min) && (D < 0) && (alpha * pow(-D, 2.3) > pow(th0, 1.1));
                        if (sw) { ok2 = (phi2 <= phi0 + 1e-4 * alpha * D + 1e-12 * fabs(phi0)); if (ok2) armijo_case = 1; }
                        else ok2 = (th2 <= (1 - 1e-5) * th0) || (phi2 <= phi0 - 1e-5 * th0);
                    }
                    if (o->verbose) printf("      soc %d: a_soc %.3g th %.3e -> %.3e (th0 %.3e) phi %.6e vs %.6e ok %d\n", ps, a_soc, th_prev, th2, th0, phi2, phi0, ok2);
                    if (ok2) { accepted = 1; ad = ads; break; }
                    if (th2 > 0.99 * th_prev) break;
                    th_prev = th2;
                    /* next correction: c_soc <- a_soc c_soc + c(x_k + a_soc d_soc) */
                    for (int k = 1; k < N; k++) {
                        stage_t* s = &pb.st[k];
                        for (int i = 0; i < s->nrows; i++) cs[k * (MAXROWS + NX) + i] = a_soc * cs[k * (MAXROWS + NX) + i] + s->h[i] + s->t[i];
                        if (k < N - 1) for (int i = 0; i < NX; i++) cs[k * (MAXROWS + NX) + MAXROWS + i] = a_soc * cs[k * (MAXROWS + NX) + MAXROWS + i] + s->r[i];
                    }
                    for (int i = 0; i < 24; i++) r0s[i] = a_soc * r0s[i] + pb.r0[i];
                    RESTORE_AND_LIN();
                }
                if (!accepted) {
                    for (int k = 1; k < N; k++) {
                        stage_t* s = &pb.st[k];
                        memcpy(s->dzeta, dsave + k * W, sizeof(double) * NZ);
                        memcpy(s->dt_, dsave + k * W + NZ, sizeof(double) * MAXROWS);
                        memcpy(s->dz_, dsave + k * W + NZ + MAXROWS, sizeof(double) * MAXROWS);
                    }
                    ad = ad_save;
                }
                free(dsave); free(cs);
                if (accepted) { ls_ok = 1; bmpc_dbg_soc_acc++; alpha = a_soc; break; }
            }
            alpha *= 0.5;
        }
        free(save);
        if (!ls_ok) alpha *= 2.0;                /* (the last trial, which is kept) */
        if (!armijo_case) { /* augment the filter with the current point */
            if (nfilt == MAXFILT) { memmove(filt_th, filt_th + 1, sizeof(double) * (MAXFILT - 1)); memmove(filt_phi, filt_phi + 1, sizeof(double) * (MAXFILT - 1)); nfilt--; }
            filt_th[nfilt] = (1 - 1e-5) * th0;
            filt_phi[nfilt] = phi0 - 1e-5 * th0;
            nfilt++;
        }
        double ad_eff = ad;
        for (int k = 1; k < N; k++) {
            stage_t* s = &pb.st[k];
            for (int i = 0; i < s->nrows; i++) s->z[i] += ad_eff * s->dz_[i];
        }
        /* full re-linearisation at the accepted point */
        for (int k = N - 1; k >= 1; k--) {
            /* eval_stage rebuilds rows (same order) but must keep t,z */
            eval_stage(&pb, k, 0);
        }
        err_prev = kk.err;
        alpha_last = ls_ok ? alpha : 1e300;      /* a search that found no acceptable step leaves no memory */
        info_ad = ad; info_ap = ap; info_dw = pb.hreg; info_tries = tries;
        info_bt = ls_bt;
        pb.hreg = 0.0;                       /* the next iteration's model starts without a correction */
        if (kk.err < 0.9 * err_best) { err_best = kk.err; stall = 0; } else stall++;
        if (o->verbose > 1 && lim_k > 0) {
            const stage_t* s = &pb.st[lim_k]; const row_t* r = &s->rows[lim_i];
            printf("      ftb row: stage %d kind %d gidx %d xidx %d t %.2e z %.2e dt %.2e h+t %.2e\n", lim_k, r->kind, r->gidx, r->xidx >= 0 ? r->xidx / N : -1, s->t[lim_i] - alpha * s->dt_[lim_i], s->z[lim_i], s->dt_[lim_i], 0.0);
        }
        if (o->verbose > 1) {
            double mq = 0, mu_ = 0, mpi = 0, msl = 0;
            for (int k = 1; k < N; k++) {
                const stage_t* s = &pb.st[k];
                for (int i = 0; i < 7; i++) { mq = fmax(mq, fabs(s->dzeta[Z_Q + i])); mu_ = fmax(mu_, fabs(s->dzeta[Z_U + i])); }
                for (int i = 0; i < 3; i++) mpi = fmax(mpi, fabs(s->dzeta[Z_PI + i]));
                msl = fmax(msl, fmax(fabs(s->dzeta[Z_RS]), fabs(s->dzeta[Z_PS])));
            }
            printf("      step: |dq| %.2e |du| %.2e |dpi| %.2e |dslack| %.2e\n", mq, mu_, mpi, msl);
        }
        if (o->verbose > 1) printf("      alpha_p %.3g (ftb %.3g) alpha_d %.3g ls_ok %d nu %.2e D %.2e th %.2e hreg %.1e tries %d hess %d stall %d\n", alpha, ap, ad, ls_ok, nu, D, th0, pb.hreg, tries, pb.hess, stall);
    }
done:
    if (g_info) {
        int want = (o->hess == 2) && ((err_prev < o->hess_switch) || (o->inertia == 2 && stall >= o->stall_n));
        if (o->gn_backoff > 0 && want && gn_skip > 0) want = 0;
        g_info[0] = it; g_info[1] = st; g_info[2] = mu; g_info[3] = alpha_last; g_info[4] = info_ad; g_info[5] = info_ap;
        g_info[6] = info_dw; g_info[7] = want; g_info[8] = info_tries; g_info[9] = info_bt; g_info[10] = err_prev; g_info[11] = stall;
    }
    /* ---- outputs in the reference layout ---- */
    {
        memset(x, 0, sizeof(double) * n_w);
        for (int j = 0; j < 7; j++) {
            x[W_Q(N) + j * N] = q0[j];
            x[W_DQ(N) + j * N] = dq0[j];
            x[W_DDQ(N) + j * N] = ddq0[j];
            x[W_U(N) + j * N] = u0[j];
        }
        for (int c = 0; c < 6; c++) {
            x[W_P(N) + c * N] = p0[c];
            x[W_V(N) + c * N] = v0[c];
        }
        for (int k = 1; k < N; k++) {
            stage_t* s = &pb.st[k];
            double y[NZ];
            zeta_to_y(&pb, s->zeta, y);
            for (int j = 0; j < 7; j++) {
                x[W_Q(N) + j * N + k] = y[Y_Q + j];
                x[W_DQ(N) + j * N + k] = y[Y_DQ + j];
                x[W_DDQ(N) + j * N + k] = y[Y_DDQ + j];
                x[W_U(N) + j * N + k] = y[Y_U + j];
            }
            for (int c = 0; c < 3; c++) {
                x[W_P(N) + c * N + k] = s->ppos[c];
                x[W_P(N) + (3 + c) * N + k] = s->prot[c];
            }
            for (int c = 0; c < 6; c++) x[W_V(N) + c * N + k] = s->v[c];
            x[W_RS(N) + k] = y[Y_RS];
            x[W_DRS(N) + k] = y[Y_DRS];
            x[W_PS(N) + k] = y[Y_PS];
            x[W_DPS(N) + k] = y[Y_DPS];
            if (k == 1) {
                x[W_RS(N)] = s->zeta[Z_RS];  /* rs_0 = rs~_1, drs_0 = 0 (one admissible split) */
                x[W_PS(N)] = s->zeta[Z_PS];
            }
            if (k == N - 1)
                for (int i = 0; i < 6; i++) x[W_DSL(N) + i] = y[Y_D + i];
        }
        double fv = 0, *gg = g ? g : (double*)malloc(sizeof(double) * n_g);
        bmpc_oracle_eval(N, dt, x, p, &fv, gg, NULL, NULL);
        if (f) *f = fv;
        if (viol) {
            /* BoundMPC.py:613-615 */
            double* lb = (double*)malloc(sizeof(double) * n_g * 2);
            bmpc_oracle_gbounds(N, lb, lb + n_g);
            double vs = 0;
            for (int i = 0; i < n_g; i++) {
                if (gg[i] < lb[i] - 1e-6) vs -= gg[i];
                if (gg[i] > lb[n_g + i] + 1e-6) vs += gg[i];
            }
            free(lb);
            *viol = vs;
        }
        if (!g) free(gg);
        if (lam_g && lam_x) recover_multipliers(&pb, x, lam_g, lam_x);
        else {
            if (lam_g) memset(lam_g, 0, sizeof(double) * n_g);
            if (lam_x) memset(lam_x, 0, sizeof(double) * n_w);
        }
    }
    if (iters) *iters = it;
    if (status) *status = st;
    free(pb.st); free(pb.lbq); free(pb.ubq);
    return 0;
}

int bmpc_oracle_solve_batch(const bmpc_oracle_opts* o, int B, const double* x0,
                            const double* lbx, const double* ubx, const double* p, double* x,
                            double* f, int* iters, int* status, double* viol, int nthreads) {
    int n_w = 44 * o->N + 6;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic)
#endif
    for (int b = 0; b < B; b++)
        bmpc_oracle_solve(o, x0 + (size_t)b * n_w, lbx + (size_t)b * n_w, ubx + (size_t)b * n_w,
                          p + (size_t)b * BMPC_NP, x + (size_t)b * n_w, NULL, NULL, NULL, f + b,
                          iters + b, status + b, viol + b);
    return 0;
}

/* the same, and info[b][BMPC_ORACLE_INFO]: the decisions of the last iteration of every solve (see g_info) */
int bmpc_oracle_solve_batch_info(const bmpc_oracle_opts* o, int B, const double* x0,
                                 const double* lbx, const double* ubx, const double* p, double* x,
                                 double* f, int* iters, int* status, double* viol, double* info, int nthreads) {
    int n_w = 44 * o->N + 6;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic)
#endif
    for (int b = 0; b < B; b++) {
        g_info = info + (size_t)b * BMPC_ORACLE_INFO;
        bmpc_oracle_solve(o, x0 + (size_t)b * n_w, lbx + (size_t)b * n_w, ubx + (size_t)b * n_w,
                          p + (size_t)b * BMPC_NP, x + (size_t)b * n_w, NULL, NULL, NULL, f + b,
                          iters + b, status + b, viol + b);
        g_info = NULL;
    }
    return 0;
}

/* ---- debug: analytic stage Lagrangian Hessian vs central differences of the Lagrangian
 * gradient (tests/test_oracle_solver.py).  Hout/Hfd are NZ x NZ row-major. */
static void setup_problem(const bmpc_oracle_opts* o, prob_t* pb, const double* x0, const double* lbx,
                          const double* ubx, const double* p, double* pins);

int bmpc_oracle_debug_hess(const bmpc_oracle_opts* o, const double* x0, const double* lbx,
                           const double* ubx, const double* p, int k, double zval, double lamval,
                           double* Hout, double* Hfd) {
    prob_t pb;
    double pins[40];
    setup_problem(o, &pb, x0, lbx, ubx, p, pins);
    int N = pb.N;
    pb.no_sigma = 1;
    for (int kk = N - 1; kk >= 1; kk--) eval_stage(&pb, kk, 0);
    for (int kk = 1; kk < N; kk++)
        for (int i = 0; i < pb.st[kk].nrows; i++) {
            pb.st[kk].t[i] = 1.0;
            pb.st[kk].z[i] = zval * (1 + (i % 3));
        }
    if (k < N - 1)
        for (int i = 0; i < NX; i++) pb.st[k + 1].lam[i] = lamval * (1 + (i % 5)) * ((i % 2) ? 1 : -1);
    assemble_stage(&pb, k, 0.0);
    stage_t* s = &pb.st[k];
    memcpy(Hout, s->H, sizeof s->H);
    double base[NZ];
    memcpy(base, s->zeta, sizeof base);
    double eps = 1e-6;
    for (int j = 0; j < NZ; j++) {
        double gp[NZ], gm[NZ];
        for (int sgn = 0; sgn < 2; sgn++) {
            memcpy(s->zeta, base, sizeof base);
            s->zeta[j] += sgn ? -eps : eps;
            eval_stage(&pb, k, 0);
            assemble_stage(&pb, k, 0.0);
            double* gg = sgn ? gm : gp;
            memcpy(gg, s->gdual, sizeof s->gdual);
            if (k < N - 1) {
                const double* lam = pb.st[k + 1].lam;
                for (int i = 0; i < NX; i++) {
                    double sm = 0;
                    for (int l = 0; l < NX; l++) sm += s->A[l * NX + i] * lam[l];
                    gg[i] += sm;
                }
                for (int i = 0; i < NU; i++) {
                    double sm = 0;
                    for (int l = 0; l < NX; l++) sm += s->B[l * NU + i] * lam[l];
                    gg[NX + i] += sm;
                }
            }
        }
        for (int i = 0; i < NZ; i++) Hfd[i * NZ + j] = (gp[i] - gm[i]) / (2 * eps);
    }
    free(pb.st); free(pb.lbq); free(pb.ubq);
    return 0;
}
