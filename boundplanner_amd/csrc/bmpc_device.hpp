// bmpc_device.hpp -- device code of the batched BoundMPC interior-point solver (gfx950).
//
// One 64-lane wavefront (= one 64-thread workgroup) solves one problem instance at a time:
// the receding-horizon NLP that /root/reference/bound_planner/BoundMPC/BoundMPC.py:594-603
// hands to CasADi/IPOPT (formulation casadi_ocp_formulation.py:13-421).  The stage-banded KKT
// system is factorised by a Riccati recursion whose value-function Hessian P (32x32), stage
// Hessian W (41x41) and all stage linearisation data live in LDS; only the per-stage gains,
// the iterate and the row slacks/multipliers go to a per-workgroup scratch slab in HBM/L2
// (coalesced, lane = consecutive double).  No MFMA: the largest dense contraction is 41x41 by
// rank-9, far below a matrix-core tile.
//
// Cross-lane traffic goes through LDS + workgroup barriers only (free for a one-wave
// workgroup), so the same source runs under the CPU thread emulation in tests/emu/.
//
// Formulation (identical to oracle/bmpc_solve.c, which documents the algebra):
//   stage k = 1..N-1, state x = (q~7, dq~7, ddq~7, pi3, rs~, ps~, d6), input w = (u7, drs, dps)
//   natural variables  q = q~ + c3 u, dq = dq~ + c2 u, ddq = ddq~ + c1 u,
//                      p_rot = pi + dt/2 w(q,dq), rs = rs~ + dt/2 drs, ps = ps~ + dt/2 dps.
#pragma once
#include <math.h>

namespace bmpc {

constexpr int NX = 32, NU = 9, NZ = 41, LDW = 42, LDP = 33;
constexpr int NSLOT = 208;      // inequality-row slots per stage
constexpr int NPOSE = 43;       // rows living in pose space
constexpr int ZPAD = 48;        // padded stage vector length in the scratch slab
constexpr double BIGB = 1e19;

// zeta = (x, w) coordinates
constexpr int Z_Q = 0, Z_DQ = 7, Z_DDQ = 14, Z_PI = 21, Z_RS = 24, Z_PS = 25, Z_D = 26, Z_U = 32,
              Z_DRS = 39, Z_DPS = 40;

// parameter vector offsets (casadi_ocp_formulation.py:383-415)
constexpr int P_SPLIT = 0, P_SLACKS0 = 5, P_IWREF = 11, P_DTAU = 14, P_DTAU_PAR = 26,
              P_DTAU_O1 = 38, P_DTAU_O2 = 50, P_XPHID = 62, P_PHISW = 65, P_JACR = 70, P_JACL = 79,
              P_PREF = 88, P_DPREF = 112, P_DPN = 136, P_BP1 = 148, P_BP2 = 160, P_BR1 = 172,
              P_BR2 = 184, P_ERB = 196, P_W = 220, P_PHIMAX = 231, P_V1 = 232, P_V2 = 244,
              P_V3 = 256, P_ASET = 275, P_BSET = 455, P_ASETJ = 515, P_BSETJ = 785, NPAR = 875;

// row slots
constexpr int S_BOX = 0, S_NONNEG = 56, S_RS1 = 60, S_D1 = 62, S_EE = 68, S_ROTU = 83, S_ROTL = 86,
              S_COL = 89, S_PHI = 179, S_TSET = 180, S_TROTU = 195, S_TROTL = 198, S_END = 201;

struct SolverOpts {
    int N;
    double dt, tol;
    int max_iter;
    int hess;                 // 0 Gauss-Newton, 2 hybrid (second-order kinematic terms when convex)
    double hess_switch, mu_init, kappa_mu, theta_mu, kappa_eps;
};

struct RobotConst {           // iiwa14 chain constants (iiwa.urdf), rotations precomputed on the host
    double jxyz[7][3];
    double jrot[7][9];
    double ee_xyz[3];
    double ee_rot[9];
    double l4c_xyz[3];
};

struct KernelArgs {
    int B;
    SolverOpts o;
    const RobotConst* rc;
    const double *x0, *lbx, *ubx, *p;   // [B][n_w] x3, [B][875]
    double *x, *f, *viol, *g;           // [B][n_w], [B], [B], [B][n_g] or null
    int *iters, *status;
    double* ws;                         // scratch: nblocks * ws_doubles(N)
};

BMPC_HD int ws_doubles(int N) { return N * (3 * ZPAD + 5 * NSLOT + NU * NX + 32); }

// ------------------------------------------------------------------------------------------
// LDS carve-up (doubles)
// ------------------------------------------------------------------------------------------
struct Lds {
    double *P, *W, *sp, *zeta, *znext, *yz, *g0, *g1, *gz, *lam, *pv0, *pv1, *vt0, *vt1, *rdef,
        *J, *G, *Jp, *zax, *pc, *Op, *Ov, *T1, *T2, *Hp, *Hv, *mS, *sS, *bp0, *bp1, *bpz, *bv,
        *bS0, *bS1, *bSz, *M3, *mc, *sc, *b30, *b31, *b3z, *bc0, *bc1, *bcz, *rowS, *rowA, *rowSl,
        *rc, *Kl, *kf, *Y, *Et, *red, *dx, *dxn, *dloc, *dpt, *x1fix, *r0, *misc;
};
constexpr int LDS_DOUBLES =
    NX * LDP + NZ * LDW + NPAR + 2 * ZPAD + ZPAD +            // P W sp zeta znext yz
    3 * ZPAD + 5 * NX + 2 * NX + NX +                         // g0 g1 gz | lam pv0 pv1 vt0 vt1 | rdef..
    42 + 42 + 126 + 21 + 18 +                                 // J G Jp zax pc
    4 * 102 + 36 + 36 + 18 + 3 + 18 + 6 + 9 +                 // Op Ov T1 T2 Hp Hv mS sS bp* bv bS*
    54 + 18 + 6 + 54 + 18 +                                   // M3 mc sc b3* bc*
    4 * NSLOT + NPOSE * 6 + NPOSE +                           // rowS rowA rowSl
    160 + NU * NX + 32 + NZ * 3 + 3 * NZ + 64 +               // rc Kl kf Y Et red
    2 * NX + 16 + 24 + 24 + NX + 64;                          // dx dxn dloc dpt x1fix r0 misc

BMPC_INL void lds_carve(double* b, Lds& L) {
    auto take = [&](int n) { double* r = b; b += n; return r; };
    L.P = take(NX * LDP); L.W = take(NZ * LDW); L.sp = take(NPAR);
    L.zeta = take(ZPAD); L.znext = take(ZPAD); L.yz = take(ZPAD);
    L.g0 = take(ZPAD); L.g1 = take(ZPAD); L.gz = take(ZPAD);
    L.lam = take(NX); L.pv0 = take(NX); L.pv1 = take(NX); L.vt0 = take(NX); L.vt1 = take(NX);
    L.rdef = take(NX); take(2 * NX);
    L.J = take(42); L.G = take(42); L.Jp = take(126); L.zax = take(21); L.pc = take(18);
    L.Op = take(102); L.Ov = take(102); L.T1 = take(102); L.T2 = take(102);
    L.Hp = take(36); L.Hv = take(36); L.mS = take(18); L.sS = take(3);
    L.bp0 = take(6); L.bp1 = take(6); L.bpz = take(6); L.bv = take(6);
    L.bS0 = take(3); L.bS1 = take(3); L.bSz = take(3);
    L.M3 = take(54); L.mc = take(18); L.sc = take(6);
    L.b30 = take(18); L.b31 = take(18); L.b3z = take(18);
    L.bc0 = take(6); L.bc1 = take(6); L.bcz = take(6);
    L.rowS = take(4 * NSLOT); L.rowA = take(NPOSE * 6); L.rowSl = take(NPOSE);
    L.rc = take(160); L.Kl = take(NU * NX); L.kf = take(32); L.Y = take(NZ * 3); L.Et = take(3 * NZ);
    L.red = take(64); L.dx = take(NX); L.dxn = take(NX); L.dloc = take(16); L.dpt = take(24);
    L.x1fix = take(24); L.r0 = take(NX); L.misc = take(64);
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
BMPC_INL double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
BMPC_INL void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
BMPC_INL void mat3mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
BMPC_INL void mat3vec(const double* A, const double* v, double* r) {
    for (int i = 0; i < 3; i++) r[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

// workgroup-wide reductions through LDS (fixed summation order -> reproducible)
BMPC_DEV double wg_sum(double v, double* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = 0;
    for (int i = 0; i < 64; i++) s += red[i];
    return s;
}
BMPC_DEV double wg_max(double v, double* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = red[0];
    for (int i = 1; i < 64; i++) s = fmax(s, red[i]);
    return s;
}
BMPC_DEV double wg_min(double v, double* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = red[0];
    for (int i = 1; i < 64; i++) s = fmin(s, red[i]);
    return s;
}

// sparse column structure of [As Bs] (constant part of the stage dynamics) for zeta column c:
// returns number of (x-row, coefficient) pairs.
struct DynC { double dt, b1, b2, b3, c1, c2, c3; };
BMPC_INL int phi_col(int c, const DynC& d, int* idx, double* cf) {
    if (c < Z_DQ) { idx[0] = c; cf[0] = 1.0; return 1; }
    if (c < Z_DDQ) { idx[0] = c - 7; cf[0] = d.dt; idx[1] = c; cf[1] = 1.0; return 2; }
    if (c < Z_PI) { idx[0] = c - 14; cf[0] = 0.5 * d.dt * d.dt; idx[1] = c - 7; cf[1] = d.dt; idx[2] = c; cf[2] = 1.0; return 3; }
    if (c < Z_U) { idx[0] = c; cf[0] = 1.0; return 1; }
    if (c < Z_DRS) { int j = c - Z_U; idx[0] = Z_Q + j; cf[0] = d.b3; idx[1] = Z_DQ + j; cf[1] = d.b2; idx[2] = Z_DDQ + j; cf[2] = d.b1; return 3; }
    if (c == Z_DRS) { idx[0] = Z_RS; cf[0] = d.dt; return 1; }
    idx[0] = Z_PS; cf[0] = d.dt; return 1;
}

// ------------------------------------------------------------------------------------------
// per-stage context computed redundantly by every lane (registers)
// ------------------------------------------------------------------------------------------
struct Seg {
    int s, n;
    double dp[6], pref[6], phi_start, phi_end_seg;
    double dpn[3], dpnn[3], bp1[3], bp2[3], br1[3], br2[3], br1n[3], br2n[3];
    double v1[3], v2[3], v3[3], e_init[3], e_par0[3], e_o10[3], e_o20[3], iwref0[3];
    double ub[3], lb[3], ubn[3], lbn[3], p_end[3], jl[9], jr[9];
};

#define TABP(off, seg, c) sp[(off) + (c) * 4 + (seg)]

BMPC_DEV void seg_ctx(int N, const double* sp, int k, Seg& sc) {
    // bound_mpc_functions.py:49-82 (segment selection), :85-253, :256-390
    int s = 0;
    if ((double)k > sp[P_SPLIT + 1]) s = 1;
    if ((double)k > sp[P_SPLIT + 2]) s = 2;
    int n = (sp[P_SPLIT + 1] == (double)N) ? 1 : ((sp[P_SPLIT + 2] == (double)N) ? 2 : 3);
    sc.s = s; sc.n = n;
    for (int c = 0; c < 6; c++) { sc.dp[c] = TABP(P_DPREF, s, c); sc.pref[c] = TABP(P_PREF, s, c); }
    sc.phi_start = sp[P_PHISW + s];
    sc.phi_end_seg = sp[P_PHISW + n];
    bool iw_param = ((double)k <= sp[P_SPLIT + 1]);
    for (int c = 0; c < 3; c++) {
        sc.dpn[c] = TABP(P_DPN, s, c);   sc.dpnn[c] = TABP(P_DPN, s + 1, c);
        sc.bp1[c] = TABP(P_BP1, s, c);   sc.bp2[c] = TABP(P_BP2, s, c);
        sc.br1[c] = TABP(P_BR1, s, c);   sc.br2[c] = TABP(P_BR2, s, c);
        sc.br1n[c] = TABP(P_BR1, s + 1, c); sc.br2n[c] = TABP(P_BR2, s + 1, c);
        sc.v1[c] = TABP(P_V1, s, c); sc.v2[c] = TABP(P_V2, s, c); sc.v3[c] = TABP(P_V3, s, c);
        sc.e_init[c] = sp[P_DTAU + 3 * s + c];
        sc.e_par0[c] = sp[P_DTAU_PAR + 3 * s + c];
        sc.e_o10[c] = sp[P_DTAU_O1 + 3 * s + c];
        sc.e_o20[c] = sp[P_DTAU_O2 + 3 * s + c];
        sc.ub[c] = TABP(P_ERB, s, c);     sc.lb[c] = TABP(P_ERB, s, 3 + c);
        sc.ubn[c] = TABP(P_ERB, s + 1, c); sc.lbn[c] = TABP(P_ERB, s + 1, 3 + c);
        sc.p_end[c] = TABP(P_PREF, s + 1, c);
        sc.iwref0[c] = iw_param ? sp[P_IWREF + c] : sc.pref[3 + c];
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { sc.jr[3 * r + c] = sp[P_JACR + 3 * c + r]; sc.jl[3 * r + c] = sp[P_JACL + 3 * c + r]; }
}

struct Kin {
    double o[7][3], z[7][3], pee[3], Ree[9], pc[6][3];
};

BMPC_DEV void kin_eval(const RobotConst* rc, const double* q, Kin& k) {
    // RobotModel.py:146-231 restated from the URDF chain (oracle/bmpc_kin.c)
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}, Rn[9], tmp[3];
    for (int i = 0; i < 7; i++) {
        mat3vec(R, rc->jxyz[i], tmp);
        for (int a = 0; a < 3; a++) t[a] += tmp[a];
        mat3mul(R, rc->jrot[i], Rn);
        for (int a = 0; a < 3; a++) { k.o[i][a] = t[a]; k.z[i][a] = Rn[3 * a + 2]; }
        double c = cos(q[i]), s = sin(q[i]);
        double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
        mat3mul(Rn, Rz, R);
        if (i == 3) {
            mat3vec(R, rc->l4c_xyz, tmp);
            for (int a = 0; a < 3; a++) k.pc[5][a] = t[a] + tmp[a];
        }
    }
    mat3vec(R, rc->ee_xyz, tmp);
    for (int a = 0; a < 3; a++) k.pee[a] = t[a] + tmp[a];
    mat3mul(R, rc->ee_rot, k.Ree);
    for (int c = 0; c < 5; c++)
        for (int a = 0; a < 3; a++) k.pc[c][a] = k.o[c + 2][a];
}

// Jacobian J (6x7), v = J dq, G = d(J dq)/dq (6x7); row-major into registers
BMPC_DEV void kin_jac(const Kin& k, const double* dq, double J[6][7], double G[6][7], double v[6]) {
    for (int i = 0; i < 7; i++) {
        double r[3], c[3];
        for (int a = 0; a < 3; a++) r[a] = k.pee[a] - k.o[i][a];
        cross3(k.z[i], r, c);
        for (int a = 0; a < 3; a++) { J[a][i] = c[a]; J[3 + a][i] = k.z[i][a]; }
    }
    for (int a = 0; a < 6; a++) {
        double s = 0;
        for (int j = 0; j < 7; j++) s += J[a][j] * dq[j];
        v[a] = s;
    }
    double sufc[8][3], sufz[8][3], prez[8][3];
    for (int a = 0; a < 3; a++) { sufc[7][a] = 0; sufz[7][a] = 0; prez[0][a] = 0; }
    for (int j = 6; j >= 0; j--)
        for (int a = 0; a < 3; a++) {
            sufc[j][a] = sufc[j + 1][a] + J[a][j] * dq[j];
            sufz[j][a] = sufz[j + 1][a] + k.z[j][a] * dq[j];
        }
    for (int j = 0; j < 7; j++)
        for (int a = 0; a < 3; a++) prez[j + 1][a] = prez[j][a] + k.z[j][a] * dq[j];
    for (int i = 0; i < 7; i++) {
        double ci[3] = {J[0][i], J[1][i], J[2][i]}, t1[3], t2[3], t3[3];
        cross3(k.z[i], sufc[i], t1);
        cross3(prez[i], ci, t2);
        cross3(k.z[i], sufz[i + 1], t3);
        for (int a = 0; a < 3; a++) { G[a][i] = t1[a] + t2[a]; G[3 + a][i] = t3[a]; }
    }
}

struct Pose {   // bound_mpc_functions.error_function + reference_function at one stage
    double phi, dphi, sig, dsig, ep[3], er[3], epar[3], eo1[3], eo2[3];
    double proj[3], projn[3];
    double Dep[3][3], Der[3][6], gs[3][6];
};

BMPC_DEV void pose_eval(const Seg& sc, const double* pose, const double* v, const double* iw0, double phi_max, Pose& pe) {
    const double* dpp = sc.dp;
    const double* dpr = sc.dp + 3;
    double d[3], pdr[3], tmp[3], delta[3], jrdpr[3];
    for (int a = 0; a < 3; a++) d[a] = pose[a] - sc.pref[a];
    double phil = dot3(d, dpp);
    pe.phi = phil + sc.phi_start;
    pe.dphi = dot3(v, dpp);
    for (int a = 0; a < 3; a++) { pe.ep[a] = d[a] - dpp[a] * phil; pdr[a] = dpr[a] * phil + sc.pref[3 + a]; }
    for (int a = 0; a < 3; a++) tmp[a] = pose[3 + a] - iw0[a];
    for (int a = 0; a < 3; a++) delta[a] = dot3(sc.jl + 3 * a, tmp);
    for (int a = 0; a < 3; a++) tmp[a] = pdr[a] - sc.iwref0[a];
    for (int a = 0; a < 3; a++) delta[a] -= dot3(sc.jr + 3 * a, tmp);
    for (int a = 0; a < 3; a++) { pe.er[a] = sc.e_init[a] + delta[a]; jrdpr[a] = dot3(sc.jr + 3 * a, dpr); }
    double sc1 = dot3(delta, sc.v1), scp = dot3(delta, sc.v2), sc2 = dot3(delta, sc.v3);
    for (int a = 0; a < 3; a++) {
        pe.eo1[a] = sc.e_o10[a] + sc1 * sc.br1[a];
        pe.epar[a] = sc.e_par0[a] + scp * sc.dpn[a];
        pe.eo2[a] = sc.e_o20[a] + sc2 * sc.br2[a];
    }
    pe.proj[0] = dot3(sc.br1, pe.eo1); pe.proj[1] = dot3(sc.dpn, pe.epar); pe.proj[2] = dot3(sc.br2, pe.eo2);
    pe.projn[0] = dot3(sc.br1n, pe.eo1); pe.projn[1] = dot3(sc.dpnn, pe.epar); pe.projn[2] = dot3(sc.br2n, pe.eo2);
    double e = exp(-60.0 * (pe.phi - (phi_max - 0.05)));
    pe.sig = 1.0 / (1.0 + e);
    pe.dsig = 60.0 * pe.sig * (1.0 - pe.sig);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            pe.Dep[a][b] = (a == b ? 1.0 : 0.0) - dpp[a] * dpp[b];
            pe.Der[a][b] = -jrdpr[a] * dpp[b];
            pe.Der[a][3 + b] = sc.jl[3 * a + b];
        }
    const double* vv[3] = {sc.v1, sc.v2, sc.v3};
    for (int m = 0; m < 3; m++) {
        double c = dot3(vv[m], jrdpr);
        for (int b = 0; b < 3; b++) {
            pe.gs[m][b] = -c * dpp[b];
            pe.gs[m][3 + b] = sc.jl[b] * vv[m][0] + sc.jl[3 + b] * vv[m][1] + sc.jl[6 + b] * vv[m][2];
        }
    }
}

// stage cost in output space (pose6, v6): value, gradient, Gauss-Newton/convex Hessian blocks
BMPC_DEV double stage_cost_o(const Seg& sc, const Pose& pe, const double* v, const double* wts, const double* xphid,
                             bool terminal, double* g12, double* Hp, double* Hv, bool want_h) {
    const double* dpp = sc.dp;
    double w_p = wts[0], w_r = wts[1], w_vp = wts[2], w_vr = wts[3], w_phi = wts[4], w_dphi = wts[5];
    double sig = pe.sig;
    double er2 = dot3(pe.er, pe.er), ep2 = dot3(pe.ep, pe.ep);
    double vo[6], Wvo[6];
    for (int a = 0; a < 6; a++) { vo[a] = v[a] - pe.dphi * sc.dp[a]; Wvo[a] = (a < 3 ? w_vp : w_vr) * vo[a]; }
    double dphid = xphid[0] - pe.phi;
    double rt = sqrt(dphid * dphid + 0.01);
    double val = sig * sig * (er2 + ep2) + w_r * dot3(pe.epar, pe.epar);
    val += w_vp * (vo[0] * vo[0] + vo[1] * vo[1] + vo[2] * vo[2]) + w_vr * (vo[3] * vo[3] + vo[4] * vo[4] + vo[5] * vo[5]);
    val += w_phi * (rt - 0.1) + w_dphi * (xphid[1] - pe.dphi) * (xphid[1] - pe.dphi);
    val += w_p * ep2 + w_r / 50.0 * (dot3(pe.eo1, pe.eo1) + dot3(pe.eo2, pe.eo2));
    if (terminal)
        for (int a = 0; a < 6; a++) val += 100.0 * v[a] * v[a];
    double dpsi = -w_phi * dphid / rt, ddpsi = w_phi * 0.01 / (rt * rt * rt);
    for (int b = 0; b < 6; b++) {
        double s1 = 0;
        for (int a = 0; a < 3; a++) s1 += pe.Der[a][b] * pe.er[a];
        double gp = 2 * sig * sig * s1;
        if (b < 3) {
            double s2 = 0;
            for (int a = 0; a < 3; a++) s2 += pe.Dep[a][b] * pe.ep[a];
            gp += 2 * (sig * sig + w_p) * s2 + (2 * sig * pe.dsig * (er2 + ep2) + dpsi) * dpp[b];
        }
        gp += 2 * w_r * pe.proj[1] * pe.gs[1][b] + 2 * (w_r / 50.0) * (pe.proj[0] * pe.gs[0][b] + pe.proj[2] * pe.gs[2][b]);
        g12[b] = gp;
    }
    double dWvo = 0;
    for (int a = 0; a < 6; a++) dWvo += sc.dp[a] * Wvo[a];
    for (int b = 0; b < 6; b++) {
        double gv = 2 * Wvo[b];
        if (b < 3) gv += (-2 * dWvo - 2 * w_dphi * (xphid[1] - pe.dphi)) * dpp[b];
        if (terminal) gv += 200.0 * v[b];
        g12[6 + b] = gv;
    }
    if (want_h) {
        double R1[3][6], R2[3][6];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 6; b++) {
                double dphib = (b < 3) ? dpp[b] : 0.0;
                R1[a][b] = sig * pe.Der[a][b] + pe.er[a] * pe.dsig * dphib;
                R2[a][b] = (b < 3 ? sig * pe.Dep[a][b] : 0.0) + pe.ep[a] * pe.dsig * dphib;
            }
        double n_dpn = dot3(sc.dpn, sc.dpn), n_b1 = dot3(sc.br1, sc.br1), n_b2 = dot3(sc.br2, sc.br2);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
                double h = 0;
                for (int a = 0; a < 3; a++) h += R1[a][i] * R1[a][j] + R2[a][i] * R2[a][j];
                h *= 2;
                h += 2 * w_r * n_dpn * pe.gs[1][i] * pe.gs[1][j];
                h += 2 * (w_r / 50.0) * (n_b1 * pe.gs[0][i] * pe.gs[0][j] + n_b2 * pe.gs[2][i] * pe.gs[2][j]);
                if (i < 3 && j < 3) {
                    double dd = 0;
                    for (int a = 0; a < 3; a++) dd += pe.Dep[a][i] * pe.Dep[a][j];
                    h += 2 * w_p * dd + ddpsi * dpp[i] * dpp[j];
                }
                Hp[6 * i + j] = h;
            }
        double dWd = 0;
        for (int a = 0; a < 6; a++) dWd += sc.dp[a] * sc.dp[a] * (a < 3 ? w_vp : w_vr);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
                double wi = (i < 3 ? w_vp : w_vr), wj = (j < 3 ? w_vp : w_vr);
                double di = (i < 3) ? dpp[i] : 0.0, dj = (j < 3) ? dpp[j] : 0.0;
                double h = (i == j ? wi : 0.0) - wi * sc.dp[i] * dj - di * wj * sc.dp[j] + di * dj * dWd;
                h = 2 * h + 2 * w_dphi * di * dj;
                if (terminal && i == j) h += 200.0;
                Hv[6 * i + j] = h;
            }
    }
    return val;
}

// ------------------------------------------------------------------------------------------
// stage evaluation shared by every pass
// ------------------------------------------------------------------------------------------
// rc[] layout (per-stage row context in LDS, written by lane 0)
constexpr int RC_POSE = 0, RC_PROJ = 6, RC_PROJN = 9, RC_GS = 12, RC_GSN = 30, RC_PHI = 48,
              RC_UB = 49, RC_LB = 52, RC_UBN = 55, RC_LBN = 58, RC_DPP = 61, RC_PHIEND = 64,
              RC_TZ = 65 /*z1,z2*/, RC_BP1 = 67, RC_BP2 = 70, RC_DEP = 73, RC_PEND = 82,
              RC_SL = 85 /*sl0+d (6)*/, RC_FVAL = 91, RC_V = 92 /*v6*/, RC_PROT = 98, RC_SEG = 101 /*s,n*/;

struct StageEval {
    Seg sc;
    Kin kin;
    Pose pe;
    double J[6][7], G[6][7], v[6], pose[6];
    double g12[12];
    double fval;
};

// natural values (in zeta index positions) from zeta
BMPC_INL double nat_from_zeta(const double* z, int i, const DynC& d) {
    if (i < Z_DQ) return z[i] + d.c3 * z[Z_U + i];
    if (i < Z_DDQ) return z[i] + d.c2 * z[Z_U + i - 7];
    if (i < Z_PI) return z[i] + d.c1 * z[Z_U + i - 14];
    if (i == Z_RS) return z[i] + 0.5 * d.dt * z[Z_DRS];
    if (i == Z_PS) return z[i] + 0.5 * d.dt * z[Z_DPS];
    return z[i];
}

// Phase 1 of every pass: lanes fill yz; every lane evaluates kinematics/pose/cost redundantly;
// lane 0 publishes the row context and the Jacobians to LDS.
BMPC_DEV void stage_eval(const KernelArgs& A, const Lds& L, const DynC& dc, int k, int lane, bool want_h,
                         const double* iw0, StageEval& E) {
    const int N = A.o.N;
    const double* sp = L.sp;
    if (lane < NZ) L.yz[lane] = nat_from_zeta(L.zeta, lane, dc);
    BMPC_SYNC();
    double q[7], dq[7];
    for (int j = 0; j < 7; j++) { q[j] = L.yz[Z_Q + j]; dq[j] = L.yz[Z_DQ + j]; }
    kin_eval(A.rc, q, E.kin);
    kin_jac(E.kin, dq, E.J, E.G, E.v);
    for (int a = 0; a < 3; a++) {
        E.pose[a] = E.kin.pee[a];
        E.pose[3 + a] = L.yz[Z_PI + a] + 0.5 * dc.dt * E.v[3 + a];
    }
    seg_ctx(N, sp, k, E.sc);
    pose_eval(E.sc, E.pose, E.v, iw0, sp[P_PHIMAX], E.pe);
    bool term = (k == N - 1);
    double Hp[36], Hv[36];
    double fv = stage_cost_o(E.sc, E.pe, E.v, sp + P_W, sp + P_XPHID, term, E.g12, Hp, Hv, want_h);
    const double* wts = sp + P_W;
    for (int j = 2; j <= 4; j++) fv += wts[6] * dq[j] * dq[j];
    for (int j = 0; j < 7; j++) fv += wts[7] * L.yz[Z_U + j] * L.yz[Z_U + j];
    fv += wts[9] * L.yz[Z_RS] * L.yz[Z_RS] + wts[10] * L.yz[Z_DRS] * L.yz[Z_DRS] +
          wts[9] * L.yz[Z_PS] * L.yz[Z_PS] + wts[10] * L.yz[Z_DPS] * L.yz[Z_DPS];
    if (term)
        for (int i = 0; i < 6; i++) {
            double sl = sp[P_SLACKS0 + i] + L.yz[Z_D + i];
            if (i != 4) fv += wts[8] * sl * sl;
            fv += wts[10] * L.yz[Z_D + i] * L.yz[Z_D + i];
        }
    E.fval = fv;
    BMPC_SYNC();
    if (lane == 0) {
        double* rc = L.rc;
        double nb[3] = {dot3(E.sc.br1, E.sc.br1), dot3(E.sc.dpn, E.sc.dpn), dot3(E.sc.br2, E.sc.br2)};
        double cc[3] = {dot3(E.sc.br1n, E.sc.br1), dot3(E.sc.dpnn, E.sc.dpn), dot3(E.sc.br2n, E.sc.br2)};
        for (int a = 0; a < 6; a++) rc[RC_POSE + a] = E.pose[a];
        for (int m = 0; m < 3; m++) {
            rc[RC_PROJ + m] = E.pe.proj[m]; rc[RC_PROJN + m] = E.pe.projn[m];
            for (int c = 0; c < 6; c++) { rc[RC_GS + 6 * m + c] = nb[m] * E.pe.gs[m][c]; rc[RC_GSN + 6 * m + c] = cc[m] * E.pe.gs[m][c]; }
            rc[RC_UB + m] = E.sc.ub[m]; rc[RC_LB + m] = E.sc.lb[m]; rc[RC_UBN + m] = E.sc.ubn[m]; rc[RC_LBN + m] = E.sc.lbn[m];
            rc[RC_DPP + m] = E.sc.dp[m]; rc[RC_BP1 + m] = E.sc.bp1[m]; rc[RC_BP2 + m] = E.sc.bp2[m];
            rc[RC_PEND + m] = E.sc.p_end[m];
            for (int c = 0; c < 3; c++) rc[RC_DEP + 3 * m + c] = E.pe.Dep[m][c];
        }
        rc[RC_PHI] = E.pe.phi; rc[RC_PHIEND] = E.sc.phi_end_seg;
        rc[RC_TZ] = dot3(E.sc.bp1, E.pe.ep); rc[RC_TZ + 1] = dot3(E.sc.bp2, E.pe.ep);
        for (int i = 0; i < 6; i++) rc[RC_SL + i] = sp[P_SLACKS0 + i] + L.yz[Z_D + i];
        rc[RC_FVAL] = fv;
        for (int a = 0; a < 6; a++) rc[RC_V + a] = E.v[a];
        for (int a = 0; a < 3; a++) rc[RC_PROT + a] = E.pose[3 + a];
        rc[RC_SEG] = (double)E.sc.s; rc[RC_SEG + 1] = (double)E.sc.n;
        for (int c = 0; c < 6; c++)
            for (int a = 0; a < 3; a++) L.pc[3 * c + a] = E.kin.pc[c][a];
        for (int a = 0; a < 6; a++)
            for (int j = 0; j < 7; j++) { L.J[7 * a + j] = E.J[a][j]; L.G[7 * a + j] = E.G[a][j]; }
        for (int i = 0; i < 7; i++)
            for (int a = 0; a < 3; a++) L.zax[3 * i + a] = E.kin.z[i][a];
        const int nj[6] = {2, 3, 4, 5, 6, 4};
        for (int c = 0; c < 6; c++)
            for (int i = 0; i < 7; i++) {
                double r[3], cr[3] = {0, 0, 0};
                if (i < nj[c]) {
                    for (int a = 0; a < 3; a++) r[a] = E.kin.pc[c][a] - E.kin.o[i][a];
                    cross3(E.kin.z[i], r, cr);
                }
                for (int a = 0; a < 3; a++) L.Jp[21 * c + 7 * a + i] = cr[a];
            }
        if (want_h)
            for (int i = 0; i < 36; i++) { L.Hp[i] = Hp[i]; L.Hv[i] = Hv[i]; }
        for (int a = 0; a < 6; a++) { L.bp0[a] = E.g12[a]; L.bpz[a] = E.g12[a]; L.bp1[a] = 0; L.bv[a] = E.g12[6 + a]; }
    }
    BMPC_SYNC();
}

// One inequality row (slot) of stage k: h value, and a compact description of its gradient.
// kind: 0 inactive, 1 natural-diagonal (pos, coef), 2 zeta-diagonal (pos, coef), 3 pose row
// (a6 + slack selector 0 none / 1 ps / 2 rs / 3 d5, coefficient -1), 4 point row (point c, a3, -d_c)
struct Row { int kind, pos, sel; double coef, h, a[6]; };

BMPC_DEV void row_eval(const KernelArgs& A, const Lds& L, int b, int k, int s, Row& r) {
    const int N = A.o.N;
    const double* sp = L.sp;
    const double* rc = L.rc;
    r.kind = 0; r.pos = 0; r.sel = 0; r.coef = 0; r.h = 0;
    for (int c = 0; c < 6; c++) r.a[c] = 0;
    size_t xb = (size_t)b * (44 * N + 6);
    if (s < S_NONNEG) {                       // box bounds on q,dq,ddq,u (BoundMPC.py:171-186,544-589)
        int j = s >> 1, blk = j / 7, jj = j - 7 * blk;
        int pos = (blk == 0 ? Z_Q : blk == 1 ? Z_DQ : blk == 2 ? Z_DDQ : Z_U) + jj;
        size_t wi = xb + (size_t)blk * 7 * N + (size_t)jj * N + k;
        if ((s & 1) == 0) {
            double ub = A.ubx[wi];
            if (ub < BIGB) { r.kind = 1; r.pos = pos; r.coef = 1.0; r.h = L.yz[pos] - ub; }
        } else {
            double lb = A.lbx[wi];
            if (lb > -BIGB) { r.kind = 1; r.pos = pos; r.coef = -1.0; r.h = lb - L.yz[pos]; }
        }
    } else if (s < S_RS1) {                   // rs, drs, ps, dps >= 0 (Q6)
        int m = s - S_NONNEG;
        int pos = (m == 0 ? Z_RS : m == 1 ? Z_DRS : m == 2 ? Z_PS : Z_DPS);
        r.kind = 1; r.pos = pos; r.coef = -1.0; r.h = -L.yz[pos];
    } else if (s < S_D1) {                    // stage-0 slacks: rs~_1, ps~_1 >= 0
        if (k == 1) { int pos = (s == S_RS1) ? Z_RS : Z_PS; r.kind = 2; r.pos = pos; r.coef = -1.0; r.h = -L.zeta[pos]; }
    } else if (s < S_EE) {                    // dslacks >= 0
        if (k == 1) { int pos = Z_D + (s - S_D1); r.kind = 1; r.pos = pos; r.coef = -1.0; r.h = -L.yz[pos]; }
    } else if (s < S_ROTU) {                  // EE in current set (ocp :304)
        int rr = s - S_EE, sg = (int)rc[RC_SEG];
        const double* a = sp + P_ASET + 45 * sg;
        double a0 = a[rr], a1 = a[rr + 15], a2 = a[rr + 30], bb = sp[P_BSET + rr * 4 + sg];
        if (!(a0 == 0 && a1 == 0 && a2 == 0 && bb > 0)) {
            r.kind = 3; r.sel = 1; r.a[0] = a0; r.a[1] = a1; r.a[2] = a2;
            r.h = a0 * rc[RC_POSE] + a1 * rc[RC_POSE + 1] + a2 * rc[RC_POSE + 2] - bb - L.yz[Z_PS];
        }
    } else if (s < S_COL) {                   // orientation bounds (ocp :308-321)
        int m = s - S_ROTU;
        bool lower = m >= 3;
        if (lower) m -= 3;
        r.kind = 3; r.sel = 2;
        double sgn = lower ? -1.0 : 1.0;
        for (int c = 0; c < 6; c++) r.a[c] = sgn * rc[RC_GS + 6 * m + c];
        r.h = lower ? -(rc[RC_PROJ + m] - rc[RC_LB + m] + L.yz[Z_RS]) : (rc[RC_PROJ + m] - rc[RC_UB + m] - L.yz[Z_RS]);
    } else if (s < S_PHI) {                   // collision points (ocp :323-330)
        int c = (s - S_COL) / 15, rr = (s - S_COL) - 15 * c;
        const double* a = sp + P_ASETJ + 45 * c;
        double a0 = a[rr], a1 = a[rr + 15], a2 = a[rr + 30], bb = sp[P_BSETJ + rr * 6 + c];
        if (!(a0 == 0 && a1 == 0 && a2 == 0 && bb + sp[P_SLACKS0 + c] > 0)) {
            r.kind = 4; r.pos = c; r.a[0] = a0; r.a[1] = a1; r.a[2] = a2;
            r.h = a0 * L.pc[3 * c] + a1 * L.pc[3 * c + 1] + a2 * L.pc[3 * c + 2] - bb - rc[RC_SL + c];
        }
    } else if (s == S_PHI) {                  // phi cap (ocp :332)
        r.kind = 3; r.sel = 0;
        for (int c = 0; c < 3; c++) r.a[c] = rc[RC_DPP + c];
        r.h = rc[RC_PHI] - (rc[RC_PHIEND] + 0.005);
    } else if (s < S_TROTU) {                 // terminal next-set rows (ocp :346-358)
        if (k == N - 1) {
            int rr = s - S_TSET, nn = (int)rc[RC_SEG + 1];
            const double* a = sp + P_ASET + 45 * nn;
            double an[3] = {a[rr], a[rr + 15], a[rr + 30]};
            double bn = sp[P_BSET + rr * 4 + nn];
            if (!(an[0] == 0 && an[1] == 0 && an[2] == 0 && bn + sp[P_SLACKS0 + 5] > 0)) {
                double a1 = dot3(an, rc + RC_BP1), a2 = dot3(an, rc + RC_BP2);
                double bnew = bn - dot3(an, rc + RC_PEND);
                r.kind = 3; r.sel = 3;
                for (int c = 0; c < 3; c++) {
                    double tt = 0;
                    for (int a_ = 0; a_ < 3; a_++) tt += (a1 * rc[RC_BP1 + a_] + a2 * rc[RC_BP2 + a_]) * rc[RC_DEP + 3 * a_ + c];
                    r.a[c] = tt;
                }
                r.h = a1 * rc[RC_TZ] + a2 * rc[RC_TZ + 1] - bnew - rc[RC_SL + 5];
            }
        }
    } else if (s < S_END) {                   // terminal next-segment orientation rows (Q4, ocp :365-380)
        if (k == N - 1) {
            int m = s - S_TROTU;
            bool lower = m >= 3;
            if (lower) m -= 3;
            r.kind = 3; r.sel = 3;
            double sgn = lower ? -1.0 : 1.0;
            for (int c = 0; c < 6; c++) r.a[c] = sgn * rc[RC_GSN + 6 * m + c];
            r.h = lower ? -(rc[RC_PROJN + m] - rc[RC_LBN + m] + rc[RC_SL + 5]) : (rc[RC_PROJN + m] - rc[RC_UBN + m] - rc[RC_SL + 5]);
        }
    }
}

BMPC_INL int pose_row_index(int s) {   // slot -> 0..42 for rows living in pose space, else -1
    if (s >= S_EE && s < S_COL) return s - S_EE;
    if (s == S_PHI) return 21;
    if (s >= S_TSET && s < S_END) return 22 + (s - S_TSET);
    return -1;
}

}  // namespace bmpc
