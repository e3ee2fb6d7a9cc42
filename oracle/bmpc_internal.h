/* ORACLE (test infrastructure) -- internal declarations shared by bmpc_kin.c / bmpc_nlp.c /
 * bmpc_solve.c.  See bmpc_oracle.h for the status of this code. */
#ifndef BMPC_INTERNAL_H
#define BMPC_INTERNAL_H

#include "bmpc_oracle.h"

#define NJ 7
#define NSEG 4
#define NSET 15

typedef struct {
    double o[7][3];  /* joint origins, world */
    double z[7][3];  /* joint axes, world */
    double pee[3];   /* end_effector_link origin */
    double Ree[9];   /* end_effector_link rotation, row-major */
    double pc[6][3]; /* collision points (RobotModel.py:27-35, indices 0..5) */
} bmpc_kin;

extern const int BMPC_COL_NJ[6]; /* joints moving each collision point */

void bmpc_cross(const double a[3], const double b[3], double c[3]);
void bmpc_kin_eval(const double q[7], bmpc_kin* k);
void bmpc_kin_point_jac(const bmpc_kin* k, const double pt[3], int nj, double Jp[3][7]);
void bmpc_kin_jac(const bmpc_kin* k, double J[6][7]);
void bmpc_kin_dvdq(const bmpc_kin* k, const double J[6][7], const double dq[7], double G[6][7]);

/* ---- parameter vector offsets (casadi_ocp_formulation.py:383-415) ---- */
enum {
    P_SPLIT = 0, P_SLACKS0 = 5, P_IWREF = 11, P_DTAU = 14, P_DTAU_PAR = 26, P_DTAU_O1 = 38,
    P_DTAU_O2 = 50, P_XPHID = 62, P_PHISW = 65, P_JACR = 70, P_JACL = 79, P_PREF = 88,
    P_DPREF = 112, P_DPN = 136, P_BP1 = 148, P_BP2 = 160, P_BR1 = 172, P_BR2 = 184,
    P_ERB = 196, P_W = 220, P_PHIMAX = 231, P_V1 = 232, P_V2 = 244, P_V3 = 256, P_QD = 268,
    P_ASET = 275, P_BSET = 455, P_ASETJ = 515, P_BSETJ = 785
};

/* decision-vector block offsets (casadi_ocp_formulation.py:89-101): variable-major, time-minor */
#define W_Q(N) 0
#define W_DQ(N) (7 * (N))
#define W_DDQ(N) (14 * (N))
#define W_U(N) (21 * (N))
#define W_P(N) (28 * (N))
#define W_V(N) (34 * (N))
#define W_DSL(N) (40 * (N))
#define W_RS(N) (40 * (N) + 6)
#define W_DRS(N) (41 * (N) + 6)
#define W_PS(N) (42 * (N) + 6)
#define W_DPS(N) (43 * (N) + 6)

/* per-stage segment context: everything reference_function()/error_function() select by
 * split_idx (bound_mpc_functions.py:49-82, 85-253, 256-390) */
typedef struct {
    int s;             /* current segment 0..2 */
    int n;             /* "next" index 1..3 (bound_mpc_functions.py:177-182, 313-314) */
    double dp[6], pref[6], dpn_pos_unused;
    double phi_start, phi_end_seg;
    double dpn[3], dpnn[3];      /* dp_normed current / next */
    double bp1[3], bp2[3];
    double br1[3], br2[3], br1n[3], br2n[3];
    double v1[3], v2[3], v3[3];
    double e_init[3], e_par0[3], e_o10[3], e_o20[3];
    double iwref0[3];            /* i_w_ref_0 selected by idx <= split_idx[1] */
    int iwref_is_param;          /* 1: i_omega_ref_0 parameter, 0: p_ref_cur[3:] */
    double ub[3], lb[3], ubn[3], lbn[3];
    const double* a_cur;         /* 15x3 column-major */
    double b_cur[NSET];
    const double* a_next;        /* a_set[n] */
    double b_next[NSET];
    double p_end[3];             /* p_ref[s+1,:3] */
    double jl[9], jr[9];         /* row-major 3x3 */
} bmpc_seg;

void bmpc_seg_ctx(int N, const double* p, int k, bmpc_seg* sc);

/* stage evaluation in "output space" o = (p_pos 3, p_rot 3, v 6), plus iw0 = p[0,3:] */
typedef struct {
    double phi, dphi, sig, dsig; /* dsig = d sig / d phi */
    double ep[3], er[3], epar[3], eo1[3], eo2[3];
    double sc1, scp, sc2;
    double proj1, projp, proj2;       /* current-basis projections */
    double proj1n, projpn, proj2n;    /* next-basis projections (Q4: of the CURRENT errors) */
    double Dep[3][3];                 /* d e_p / d p_pos */
    double Der[3][6];                 /* d e_r / d pose(6) ; d e_r / d iw0 = -jl */
    double gsc1[6], gscp[6], gsc2[6]; /* gradients of sc1, scp, sc2 wrt pose(6) */
    double gsc1_w[3], gscp_w[3], gsc2_w[3]; /* ... wrt iw0 (3) */
} bmpc_pose_eval;

void bmpc_pose_eval_fn(const bmpc_seg* sc, const double pose[6], const double v[6],
                       const double iw0[3], double phi_max, bmpc_pose_eval* pe);

/* stage cost in output space; returns value, fills gradient (12 = pose6 + v6), gradient wrt iw0
 * (3), and the Gauss-Newton/convex Hessian (12x12, row-major) when H != NULL */
double bmpc_stage_cost_o(const bmpc_seg* sc, const bmpc_pose_eval* pe, const double v[6],
                         const double* wts, const double* x_phi_d, int terminal, double g12[12],
                         double g_iw0[3], double* H);

#endif
