/* ORACLE (test infrastructure).  Restatement of the BoundMPC NLP functions:
 *   segment selection    bound_mpc_functions.py:49-82
 *   reference_function   bound_mpc_functions.py:85-253   (CasADi branch)
 *   error_function       bound_mpc_functions.py:256-390  + mpc_utils_casadi.py:6-70
 *   objective_function   bound_mpc_functions.py:393-428
 *   cost / constraints   casadi_ocp_formulation.py:106-380
 * in the reference's own variable/parameter/constraint layout, with analytic first
 * derivatives.  Pinned against tests/golden/nlp_*.npz (f, g, grad f, J_g produced by running
 * the reference's formulation code numerically, see tests/golden/gen/). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bmpc_internal.h"

#define TAB(p, off, seg, c) ((p)[(off) + (c) * NSEG + (seg)])

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void bmpc_oracle_dims(int N, int* n_w, int* n_g, int* n_p) {
    if (n_w) *n_w = 44 * N + 6;
    if (n_g) *n_g = 147 * (N - 1) + 21;
    if (n_p) *n_p = BMPC_NP;
}

void bmpc_seg_ctx(int N, const double* p, int k, bmpc_seg* sc) {
    const double* split = p + P_SPLIT;
    int s = 0;
    /* get_current_segments_split: later conditions override (bound_mpc_functions.py:58-62) */
    if ((double)k > split[1]) s = 1;
    if ((double)k > split[2]) s = 2;
    int n = (split[1] == (double)N) ? 1 : ((split[2] == (double)N) ? 2 : 3);
    sc->s = s;
    sc->n = n;
    for (int c = 0; c < 6; c++) {
        sc->dp[c] = TAB(p, P_DPREF, s, c);
        sc->pref[c] = TAB(p, P_PREF, s, c);
    }
    sc->phi_start = p[P_PHISW + s];
    sc->phi_end_seg = p[P_PHISW + n];
    for (int c = 0; c < 3; c++) {
        sc->dpn[c] = TAB(p, P_DPN, s, c);
        sc->dpnn[c] = TAB(p, P_DPN, s + 1, c);
        sc->bp1[c] = TAB(p, P_BP1, s, c);
        sc->bp2[c] = TAB(p, P_BP2, s, c);
        sc->br1[c] = TAB(p, P_BR1, s, c);
        sc->br2[c] = TAB(p, P_BR2, s, c);
        sc->br1n[c] = TAB(p, P_BR1, s + 1, c);
        sc->br2n[c] = TAB(p, P_BR2, s + 1, c);
        sc->v1[c] = TAB(p, P_V1, s, c);
        sc->v2[c] = TAB(p, P_V2, s, c);
        sc->v3[c] = TAB(p, P_V3, s, c);
        sc->e_init[c] = p[P_DTAU + 3 * s + c];
        sc->e_par0[c] = p[P_DTAU_PAR + 3 * s + c];
        sc->e_o10[c] = p[P_DTAU_O1 + 3 * s + c];
        sc->e_o20[c] = p[P_DTAU_O2 + 3 * s + c];
        sc->ub[c] = TAB(p, P_ERB, s, c);
        sc->lb[c] = TAB(p, P_ERB, s, 3 + c);
        sc->ubn[c] = TAB(p, P_ERB, s + 1, c);
        sc->lbn[c] = TAB(p, P_ERB, s + 1, 3 + c);
        sc->p_end[c] = TAB(p, P_PREF, s + 1, c);
    }
    /* error_function: i_w_ref_0 = i_omega_ref_0 if idx <= split_idx[1] else p_ref_cur[3:] */
    sc->iwref_is_param = ((double)k <= split[1]);
    for (int c = 0; c < 3; c++) sc->iwref0[c] = sc->iwref_is_param ? p[P_IWREF + c] : sc->pref[3 + c];
    sc->a_cur = p + P_ASET + 45 * s;
    sc->a_next = p + P_ASET + 45 * n;
    for (int r = 0; r < NSET; r++) {
        sc->b_cur[r] = p[P_BSET + r * NSEG + s];
        sc->b_next[r] = p[P_BSET + r * NSEG + n];
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            sc->jr[3 * r + c] = p[P_JACR + 3 * c + r];
            sc->jl[3 * r + c] = p[P_JACL + 3 * c + r];
        }
}

void bmpc_pose_eval_fn(const bmpc_seg* sc, const double pose[6], const double v[6],
                       const double iw0[3], double phi_max, bmpc_pose_eval* pe) {
    const double* dpp = sc->dp;      /* position direction */
    const double* dpr = sc->dp + 3;  /* angular velocity of the reference per unit phi */
    double d[3], pdr[3], tmp[3], delta[3], jrdpr[3];
    for (int a = 0; a < 3; a++) d[a] = pose[a] - sc->pref[a];
    double phil = dot3(d, dpp);
    pe->phi = phil + sc->phi_start;
    pe->dphi = dot3(v, dpp);
    for (int a = 0; a < 3; a++) {
        pe->ep[a] = d[a] - dpp[a] * phil;
        pdr[a] = dpr[a] * phil + sc->pref[3 + a];
    }
    /* e_r = e_init + J_l (p_rot - iw0) - J_r (p_d_rot - i_w_ref_0)   (mpc_utils_casadi.py:6-14) */
    for (int a = 0; a < 3; a++) tmp[a] = pose[3 + a] - iw0[a];
    for (int a = 0; a < 3; a++) delta[a] = dot3(sc->jl + 3 * a, tmp);
    for (int a = 0; a < 3; a++) tmp[a] = pdr[a] - sc->iwref0[a];
    for (int a = 0; a < 3; a++) delta[a] -= dot3(sc->jr + 3 * a, tmp);
    for (int a = 0; a < 3; a++) {
        pe->er[a] = sc->e_init[a] + delta[a];
        jrdpr[a] = dot3(sc->jr + 3 * a, dpr);
    }
    pe->sc1 = dot3(delta, sc->v1);
    pe->scp = dot3(delta, sc->v2);
    pe->sc2 = dot3(delta, sc->v3);
    for (int a = 0; a < 3; a++) {
        pe->eo1[a] = sc->e_o10[a] + pe->sc1 * sc->br1[a];
        pe->epar[a] = sc->e_par0[a] + pe->scp * sc->dpn[a];
        pe->eo2[a] = sc->e_o20[a] + pe->sc2 * sc->br2[a];
    }
    pe->proj1 = dot3(sc->br1, pe->eo1);
    pe->projp = dot3(sc->dpn, pe->epar);
    pe->proj2 = dot3(sc->br2, pe->eo2);
    pe->proj1n = dot3(sc->br1n, pe->eo1);
    pe->projpn = dot3(sc->dpnn, pe->epar);
    pe->proj2n = dot3(sc->br2n, pe->eo2);
    double e = exp(-60.0 * (pe->phi - (phi_max - 0.05)));
    pe->sig = 1.0 / (1.0 + e);
    pe->dsig = 60.0 * pe->sig * (1.0 - pe->sig);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            pe->Dep[a][b] = (a == b ? 1.0 : 0.0) - dpp[a] * dpp[b];
            pe->Der[a][b] = -jrdpr[a] * dpp[b];
            pe->Der[a][3 + b] = sc->jl[3 * a + b];
        }
    const double* vv[3] = {sc->v1, sc->v2, sc->v3};
    double* gs[3] = {pe->gsc1, pe->gscp, pe->gsc2};
    double* gw[3] = {pe->gsc1_w, pe->gscp_w, pe->gsc2_w};
    for (int m = 0; m < 3; m++) {
        double c = dot3(vv[m], jrdpr);
        for (int b = 0; b < 3; b++) {
            gs[m][b] = -c * dpp[b];
            double jt = sc->jl[b] * vv[m][0] + sc->jl[3 + b] * vv[m][1] + sc->jl[6 + b] * vv[m][2];
            gs[m][3 + b] = jt;
            gw[m][b] = -jt;
        }
    }
}

/* weights (util_functions.py:34-48): 0 w_p 1 w_r 2 w_v_p 3 w_v_r 4 w_phi 5 w_dphi 6 w_dq 7 w_jerk
 * 8 w_term 9 w_slack 10 w_dslack */
double bmpc_stage_cost_o(const bmpc_seg* sc, const bmpc_pose_eval* pe, const double v[6],
                         const double* wts, const double* x_phi_d, int terminal, double g12[12],
                         double g_iw0[3], double* H) {
    const double* dpp = sc->dp;
    double w_p = wts[0], w_r = wts[1], w_vp = wts[2], w_vr = wts[3], w_phi = wts[4], w_dphi = wts[5];
    double sig = pe->sig;
    double er2 = dot3(pe->er, pe->er), ep2 = dot3(pe->ep, pe->ep);
    double vo[6], Wvo[6];
    for (int a = 0; a < 6; a++) vo[a] = v[a] - pe->dphi * sc->dp[a];
    for (int a = 0; a < 6; a++) Wvo[a] = (a < 3 ? w_vp : w_vr) * vo[a];
    double dphid = x_phi_d[0] - pe->phi;
    double rt = sqrt(dphid * dphid + 0.01);
    double val = sig * sig * (er2 + ep2);                                    /* :272-276 */
    val += w_r * dot3(pe->epar, pe->epar);                                   /* :407 */
    val += w_vp * (vo[0] * vo[0] + vo[1] * vo[1] + vo[2] * vo[2]);            /* :410 */
    val += w_vr * (vo[3] * vo[3] + vo[4] * vo[4] + vo[5] * vo[5]);            /* :411 */
    val += w_phi * (rt - 0.1);                                               /* :421,427-428 */
    val += w_dphi * (x_phi_d[1] - pe->dphi) * (x_phi_d[1] - pe->dphi);        /* :422 */
    val += w_p * ep2;                                                        /* ocp :288 */
    val += w_r / 50.0 * (dot3(pe->eo1, pe->eo1) + dot3(pe->eo2, pe->eo2));   /* ocp :289-290 */
    if (terminal)
        for (int a = 0; a < 6; a++) val += 100.0 * v[a] * v[a];              /* ocp :360 */

    double dpsi = -w_phi * dphid / rt;
    double ddpsi = w_phi * 0.01 / (rt * rt * rt);
    /* gradient wrt pose */
    for (int b = 0; b < 6; b++) {
        double s1 = 0;
        for (int a = 0; a < 3; a++) s1 += pe->Der[a][b] * pe->er[a];
        double gp = 2 * sig * sig * s1;
        if (b < 3) {
            double s2 = 0;
            for (int a = 0; a < 3; a++) s2 += pe->Dep[a][b] * pe->ep[a];
            gp += 2 * (sig * sig + w_p) * s2;
            gp += (2 * sig * pe->dsig * (er2 + ep2) + dpsi) * dpp[b];
        }
        gp += 2 * w_r * pe->projp * pe->gscp[b];
        gp += 2 * (w_r / 50.0) * (pe->proj1 * pe->gsc1[b] + pe->proj2 * pe->gsc2[b]);
        g12[b] = gp;
    }
    /* gradient wrt v */
    double dWvo = 0;
    for (int a = 0; a < 6; a++) dWvo += sc->dp[a] * Wvo[a];
    for (int b = 0; b < 6; b++) {
        double gv = 2 * Wvo[b];
        if (b < 3) gv += (-2 * dWvo - 2 * w_dphi * (x_phi_d[1] - pe->dphi)) * dpp[b];
        if (terminal) gv += 200.0 * v[b];
        g12[6 + b] = gv;
    }
    /* gradient wrt iw0 = p[0, 3:] */
    for (int b = 0; b < 3; b++) {
        double s1 = 0;
        for (int a = 0; a < 3; a++) s1 += -sc->jl[3 * a + b] * pe->er[a];
        g_iw0[b] = 2 * sig * sig * s1 + 2 * w_r * pe->projp * pe->gscp_w[b] +
                   2 * (w_r / 50.0) * (pe->proj1 * pe->gsc1_w[b] + pe->proj2 * pe->gsc2_w[b]);
    }
    if (H) {
        memset(H, 0, 144 * sizeof(double));
        /* residual Jacobians of r1 = sig e_r, r2 = sig e_p wrt pose */
        double R1[3][6], R2[3][6];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 6; b++) {
                double dphib = (b < 3) ? dpp[b] : 0.0;
                R1[a][b] = sig * pe->Der[a][b] + pe->er[a] * pe->dsig * dphib;
                R2[a][b] = (b < 3 ? sig * pe->Dep[a][b] : 0.0) + pe->ep[a] * pe->dsig * dphib;
            }
        double n_dpn = dot3(sc->dpn, sc->dpn), n_b1 = dot3(sc->br1, sc->br1), n_b2 = dot3(sc->br2, sc->br2);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
                double h = 0;
                for (int a = 0; a < 3; a++) h += R1[a][i] * R1[a][j] + R2[a][i] * R2[a][j];
                h *= 2;
                h += 2 * w_r * n_dpn * pe->gscp[i] * pe->gscp[j];
                h += 2 * (w_r / 50.0) * (n_b1 * pe->gsc1[i] * pe->gsc1[j] + n_b2 * pe->gsc2[i] * pe->gsc2[j]);
                if (i < 3 && j < 3) {
                    double dd = 0;
                    for (int a = 0; a < 3; a++) dd += pe->Dep[a][i] * pe->Dep[a][j];
                    h += 2 * w_p * dd + ddpsi * dpp[i] * dpp[j];
                }
                H[12 * i + j] = h;
            }
        /* v block: 2 Dvo^T W Dvo + 2 w_dphi dp dp^T (+ 200 I), Dvo = I - dp6 [dpp^T 0] */
        double dWd = 0;
        for (int a = 0; a < 6; a++) dWd += sc->dp[a] * sc->dp[a] * (a < 3 ? w_vp : w_vr);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
                double wi = (i < 3 ? w_vp : w_vr), wj = (j < 3 ? w_vp : w_vr);
                double di = (i < 3) ? dpp[i] : 0.0, dj = (j < 3) ? dpp[j] : 0.0;
                double h = (i == j ? wi : 0.0) - wi * sc->dp[i] * dj - di * wj * sc->dp[j] + di * dj * dWd;
                h *= 2;
                h += 2 * w_dphi * di * dj;
                if (terminal && i == j) h += 200.0;
                H[12 * (6 + i) + 6 + j] = h;
            }
    }
    return val;
}

/* ------------------------------------------------------------------------------------------ */
/* full-space evaluation in the reference layout                                               */
/* ------------------------------------------------------------------------------------------ */

#define WQ(j, k) w[W_Q(N) + (j) * N + (k)]
#define WDQ(j, k) w[W_DQ(N) + (j) * N + (k)]
#define WDDQ(j, k) w[W_DDQ(N) + (j) * N + (k)]
#define WU(j, k) w[W_U(N) + (j) * N + (k)]
#define WP(c, k) w[W_P(N) + (c) * N + (k)]
#define WV(c, k) w[W_V(N) + (c) * N + (k)]

void bmpc_oracle_gbounds(int N, double* lbg, double* ubg) {
    const double INF = 1e20;
    int r = 0;
    for (int k = 0; k < N - 1; k++)
        for (int i = 0; i < 35; i++) { lbg[r] = 0; ubg[r] = 0; r++; }
    for (int k = 1; k < N; k++) {
        for (int i = 0; i < 15; i++) { lbg[r] = -INF; ubg[r] = 0; r++; }
        for (int i = 0; i < 3; i++) { lbg[r] = -INF; ubg[r] = 0; r++; }
        for (int i = 0; i < 3; i++) { lbg[r] = 0; ubg[r] = INF; r++; }
        for (int i = 0; i < 90; i++) { lbg[r] = -INF; ubg[r] = 0; r++; }
        lbg[r] = -INF; ubg[r] = 0; r++;
        if (k == N - 1) {
            for (int i = 0; i < 15; i++) { lbg[r] = -INF; ubg[r] = 0; r++; }
            for (int i = 0; i < 3; i++) { lbg[r] = -INF; ubg[r] = 0; r++; }
            for (int i = 0; i < 3; i++) { lbg[r] = 0; ubg[r] = INF; r++; }
        }
    }
}

int bmpc_oracle_eval(int N, double dt, const double* w, const double* p, double* f, double* g,
                     double* grad_f, double* jac_g) {
    int n_w = 44 * N + 6, n_g = 147 * (N - 1) + 21;
    const double* wts = p + P_W;
    double phi_max = p[P_PHIMAX];
    const double* xphid = p + P_XPHID;
    double fval = 0;
    if (grad_f) memset(grad_f, 0, n_w * sizeof(double));
    if (jac_g) memset(jac_g, 0, (size_t)n_g * n_w * sizeof(double));
#define JG(r, c) jac_g[(size_t)(r) * n_w + (c)]
    double iw0[3] = {WP(3, 0), WP(4, 0), WP(5, 0)};
    double c3a = dt * dt * dt / 8.0, c3b = dt * dt * dt / 24.0, c2a = dt * dt / 3.0, c2b = dt * dt / 6.0;

    /* ---- dynamics rows (casadi_ocp_formulation.py:106-164) ---- */
    int row = 0;
    for (int k = 0; k < N - 1; k++) {
        double qn[7], dqn[7];
        for (int j = 0; j < 7; j++) {
            qn[j] = WQ(j, k + 1);
            dqn[j] = WDQ(j, k + 1);
        }
        bmpc_kin kin;
        double J[6][7], G[6][7];
        bmpc_kin_eval(qn, &kin);
        bmpc_kin_jac(&kin, J);
        bmpc_kin_dvdq(&kin, J, dqn, G);
        for (int j = 0; j < 7; j++) {
            int r = row + j;
            if (g) g[r] = WDDQ(j, k) * dt * dt / 2 + WDQ(j, k) * dt + WQ(j, k) + WU(j, k) * c3a + WU(j, k + 1) * c3b - WQ(j, k + 1);
            if (jac_g) {
                JG(r, W_DDQ(N) + j * N + k) = dt * dt / 2;
                JG(r, W_DQ(N) + j * N + k) = dt;
                JG(r, W_Q(N) + j * N + k) = 1;
                JG(r, W_U(N) + j * N + k) = c3a;
                JG(r, W_U(N) + j * N + k + 1) = c3b;
                JG(r, W_Q(N) + j * N + k + 1) = -1;
            }
            r = row + 7 + j;
            if (g) g[r] = WDDQ(j, k) * dt + WDQ(j, k) + WU(j, k) * c2a + WU(j, k + 1) * c2b - WDQ(j, k + 1);
            if (jac_g) {
                JG(r, W_DDQ(N) + j * N + k) = dt;
                JG(r, W_DQ(N) + j * N + k) = 1;
                JG(r, W_U(N) + j * N + k) = c2a;
                JG(r, W_U(N) + j * N + k + 1) = c2b;
                JG(r, W_DQ(N) + j * N + k + 1) = -1;
            }
            r = row + 14 + j;
            if (g) g[r] = WDDQ(j, k) + WU(j, k) * dt / 2 + WU(j, k + 1) * dt / 2 - WDDQ(j, k + 1);
            if (jac_g) {
                JG(r, W_DDQ(N) + j * N + k) = 1;
                JG(r, W_U(N) + j * N + k) = dt / 2;
                JG(r, W_U(N) + j * N + k + 1) = dt / 2;
                JG(r, W_DDQ(N) + j * N + k + 1) = -1;
            }
        }
        for (int a = 0; a < 3; a++) {
            int r = row + 21 + a;
            if (g) g[r] = kin.pee[a] - WP(a, k + 1);
            if (jac_g) {
                for (int j = 0; j < 7; j++) JG(r, W_Q(N) + j * N + k + 1) = J[a][j];
                JG(r, W_P(N) + a * N + k + 1) = -1;
            }
            r = row + 24 + a;
            if (g) g[r] = WP(3 + a, k) + 0.5 * dt * (WV(3 + a, k) + WV(3 + a, k + 1)) - WP(3 + a, k + 1);
            if (jac_g) {
                JG(r, W_P(N) + (3 + a) * N + k) = 1;
                JG(r, W_V(N) + (3 + a) * N + k) = 0.5 * dt;
                JG(r, W_V(N) + (3 + a) * N + k + 1) = 0.5 * dt;
                JG(r, W_P(N) + (3 + a) * N + k + 1) = -1;
            }
        }
        for (int a = 0; a < 6; a++) {
            int r = row + 27 + a;
            double s = 0;
            for (int j = 0; j < 7; j++) s += J[a][j] * dqn[j];
            if (g) g[r] = s - WV(a, k + 1);
            if (jac_g) {
                for (int j = 0; j < 7; j++) {
                    JG(r, W_Q(N) + j * N + k + 1) = G[a][j];
                    JG(r, W_DQ(N) + j * N + k + 1) = J[a][j];
                }
                JG(r, W_V(N) + a * N + k + 1) = -1;
            }
        }
        {
            int r = row + 33;
            if (g) g[r] = w[W_RS(N) + k] + 0.5 * dt * (w[W_DRS(N) + k] + w[W_DRS(N) + k + 1]) - w[W_RS(N) + k + 1];
            if (jac_g) {
                JG(r, W_RS(N) + k) = 1;
                JG(r, W_DRS(N) + k) = 0.5 * dt;
                JG(r, W_DRS(N) + k + 1) = 0.5 * dt;
                JG(r, W_RS(N) + k + 1) = -1;
            }
            r = row + 34;
            if (g) g[r] = w[W_PS(N) + k] + 0.5 * dt * (w[W_DPS(N) + k] + w[W_DPS(N) + k + 1]) - w[W_PS(N) + k + 1];
            if (jac_g) {
                JG(r, W_PS(N) + k) = 1;
                JG(r, W_DPS(N) + k) = 0.5 * dt;
                JG(r, W_DPS(N) + k + 1) = 0.5 * dt;
                JG(r, W_PS(N) + k + 1) = -1;
            }
        }
        row += 35;
    }

    /* ---- stage cost + inequality rows (casadi_ocp_formulation.py:167-380) ---- */
    double sl[6];
    for (int i = 0; i < 6; i++) sl[i] = p[P_SLACKS0 + i] + w[W_DSL(N) + i];
    for (int k = 1; k < N; k++) {
        bmpc_seg sc;
        bmpc_pose_eval pe;
        bmpc_seg_ctx(N, p, k, &sc);
        double pose[6], v[6], qk[7];
        for (int c = 0; c < 6; c++) {
            pose[c] = WP(c, k);
            v[c] = WV(c, k);
        }
        for (int j = 0; j < 7; j++) qk[j] = WQ(j, k);
        bmpc_pose_eval_fn(&sc, pose, v, iw0, phi_max, &pe);
        int term = (k == N - 1);
        double g12[12], giw[3];
        fval += bmpc_stage_cost_o(&sc, &pe, v, wts, xphid, term, g12, giw, NULL);
        double rs = w[W_RS(N) + k], drs = w[W_DRS(N) + k], ps = w[W_PS(N) + k], dps = w[W_DPS(N) + k];
        for (int j = 2; j <= 4; j++) fval += wts[6] * WDQ(j, k) * WDQ(j, k);
        for (int j = 0; j < 7; j++) fval += wts[7] * WU(j, k) * WU(j, k);
        fval += wts[9] * rs * rs + wts[10] * drs * drs + wts[9] * ps * ps + wts[10] * dps * dps;
        if (term) {
            for (int i = 0; i < 6; i++) {
                if (i != 4) fval += wts[8] * sl[i] * sl[i]; /* slacks[:-2] and slacks[-1] (Q2) */
                fval += wts[10] * w[W_DSL(N) + i] * w[W_DSL(N) + i];
            }
        }
        if (grad_f) {
            for (int c = 0; c < 6; c++) {
                grad_f[W_P(N) + c * N + k] += g12[c];
                grad_f[W_V(N) + c * N + k] += g12[6 + c];
            }
            for (int c = 0; c < 3; c++) grad_f[W_P(N) + (3 + c) * N + 0] += giw[c];
            for (int j = 2; j <= 4; j++) grad_f[W_DQ(N) + j * N + k] += 2 * wts[6] * WDQ(j, k);
            for (int j = 0; j < 7; j++) grad_f[W_U(N) + j * N + k] += 2 * wts[7] * WU(j, k);
            grad_f[W_RS(N) + k] += 2 * wts[9] * rs;
            grad_f[W_DRS(N) + k] += 2 * wts[10] * drs;
            grad_f[W_PS(N) + k] += 2 * wts[9] * ps;
            grad_f[W_DPS(N) + k] += 2 * wts[10] * dps;
            if (term)
                for (int i = 0; i < 6; i++) {
                    if (i != 4) grad_f[W_DSL(N) + i] += 2 * wts[8] * sl[i];
                    grad_f[W_DSL(N) + i] += 2 * wts[10] * w[W_DSL(N) + i];
                }
        }
        /* EE in current set (:304) */
        for (int r = 0; r < NSET; r++) {
            double s = 0;
            for (int c = 0; c < 3; c++) s += sc.a_cur[r + NSET * c] * pose[c];
            if (g) g[row + r] = s - sc.b_cur[r] - ps;
            if (jac_g) {
                for (int c = 0; c < 3; c++) JG(row + r, W_P(N) + c * N + k) = sc.a_cur[r + NSET * c];
                JG(row + r, W_PS(N) + k) = -1;
            }
        }
        row += NSET;
        /* orientation bounds (:308-321) */
        {
            double n_b1 = dot3(sc.br1, sc.br1), n_dp = dot3(sc.dpn, sc.dpn), n_b2 = dot3(sc.br2, sc.br2);
            double pr[3] = {pe.proj1, pe.projp, pe.proj2};
            double sc_[3] = {n_b1, n_dp, n_b2};
            const double* gs[3] = {pe.gsc1, pe.gscp, pe.gsc2};
            const double* gw[3] = {pe.gsc1_w, pe.gscp_w, pe.gsc2_w};
            for (int m = 0; m < 3; m++) {
                if (g) {
                    g[row + m] = pr[m] - sc.ub[m] - rs;
                    g[row + 3 + m] = pr[m] - sc.lb[m] + rs;
                }
                if (jac_g)
                    for (int h = 0; h < 2; h++) {
                        int r = row + 3 * h + m;
                        for (int c = 0; c < 6; c++) JG(r, W_P(N) + c * N + k) += sc_[m] * gs[m][c];
                        for (int c = 0; c < 3; c++) JG(r, W_P(N) + (3 + c) * N + 0) += sc_[m] * gw[m][c];
                        JG(r, W_RS(N) + k) = h ? 1 : -1;
                    }
            }
        }
        row += 6;
        /* collision points (:323-330) */
        {
            bmpc_kin kin;
            bmpc_kin_eval(qk, &kin);
            for (int i = 0; i < 6; i++) {
                const double* aj = p + P_ASETJ + 45 * i;
                double Jp[3][7];
                if (jac_g) bmpc_kin_point_jac(&kin, kin.pc[i], BMPC_COL_NJ[i], Jp);
                for (int r = 0; r < NSET; r++) {
                    double s = 0;
                    for (int c = 0; c < 3; c++) s += aj[r + NSET * c] * kin.pc[i][c];
                    int rr = row + NSET * i + r;
                    if (g) g[rr] = s - p[P_BSETJ + r * 6 + i] - sl[i];
                    if (jac_g) {
                        for (int j = 0; j < 7; j++) {
                            double t = 0;
                            for (int c = 0; c < 3; c++) t += aj[r + NSET * c] * Jp[c][j];
                            JG(rr, W_Q(N) + j * N + k) = t;
                        }
                        JG(rr, W_DSL(N) + i) = -1;
                    }
                }
            }
        }
        row += 6 * NSET;
        /* phi cap (:332) */
        if (g) g[row] = pe.phi - (sc.phi_end_seg + 0.005);
        if (jac_g)
            for (int c = 0; c < 3; c++) JG(row, W_P(N) + c * N + k) = sc.dp[c];
        row += 1;
        if (term) {
            /* terminal next-set rows in the (bp1,bp2) plane (:346-358) */
            double z1 = dot3(sc.bp1, pe.ep), z2 = dot3(sc.bp2, pe.ep);
            for (int r = 0; r < NSET; r++) {
                double an[3] = {sc.a_next[r], sc.a_next[r + NSET], sc.a_next[r + 2 * NSET]};
                double a1 = dot3(an, sc.bp1), a2 = dot3(an, sc.bp2);
                double bnew = sc.b_next[r] - dot3(an, sc.p_end);
                if (g) g[row + r] = a1 * z1 + a2 * z2 - bnew - sl[5];
                if (jac_g) {
                    for (int c = 0; c < 3; c++) {
                        double t = 0;
                        for (int a = 0; a < 3; a++) t += (a1 * sc.bp1[a] + a2 * sc.bp2[a]) * pe.Dep[a][c];
                        JG(row + r, W_P(N) + c * N + k) = t;
                    }
                    JG(row + r, W_DSL(N) + 5) = -1;
                }
            }
            row += NSET;
            /* terminal next-segment orientation rows reuse the CURRENT errors (Q4, :365-380) */
            double c1 = dot3(sc.br1n, sc.br1), cp = dot3(sc.dpnn, sc.dpn), c2 = dot3(sc.br2n, sc.br2);
            double pr[3] = {pe.proj1n, pe.projpn, pe.proj2n};
            double sc_[3] = {c1, cp, c2};
            const double* gs[3] = {pe.gsc1, pe.gscp, pe.gsc2};
            const double* gw[3] = {pe.gsc1_w, pe.gscp_w, pe.gsc2_w};
            for (int m = 0; m < 3; m++) {
                if (g) {
                    g[row + m] = pr[m] - sc.ubn[m] - sl[5];
                    g[row + 3 + m] = pr[m] - sc.lbn[m] + sl[5];
                }
                if (jac_g)
                    for (int h = 0; h < 2; h++) {
                        int r = row + 3 * h + m;
                        for (int c = 0; c < 6; c++) JG(r, W_P(N) + c * N + k) += sc_[m] * gs[m][c];
                        for (int c = 0; c < 3; c++) JG(r, W_P(N) + (3 + c) * N + 0) += sc_[m] * gw[m][c];
                        JG(r, W_DSL(N) + 5) = h ? 1 : -1;
                    }
            }
            row += 6;
        }
    }
    if (f) *f = fval;
    return (row == n_g) ? 0 : -1;
#undef JG
}
