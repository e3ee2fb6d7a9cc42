// bmpc_ric_kernel.hpp -- the sequential part of one interior-point iteration: ONE WAVEFRONT PER
// INSTANCE runs the Riccati recursion over the horizon (the stage-banded KKT factorisation, what
// IPOPT hands to MUMPS) with the value-function Hessian P (32x32) and the stage matrix W (41x41)
// in LDS, then the convergence test / barrier update (monotone Fiacco-McCormick) and the forward
// recursion for the Newton direction dz.  Stage records come from k_eval (bmpc_pair_kernels.hpp),
// gains go to a per-pair scratch record that the forward recursion reads back.
// Also here: the per-instance control kernels of the filter line search.
#pragma once
#include "bmpc_pipeline.hpp"

namespace bmpc {

// ---- LDS layout of k_ric (doubles) ----
constexpr int R_P = 0;
constexpr int R_W = R_P + NX * LDP;
constexpr int R_g0 = R_W + NZ * LDW;
constexpr int R_g1 = R_g0 + ZPAD;
constexpr int R_gz = R_g1 + ZPAD;
constexpr int R_lam = R_gz + ZPAD;
constexpr int R_pv0 = R_lam + NX;
constexpr int R_pv1 = R_pv0 + NX;
constexpr int R_vt0 = R_pv1 + NX;
constexpr int R_vt1 = R_vt0 + NX;
constexpr int R_rdef = R_vt1 + NX;
constexpr int R_Kl = R_rdef + NX;           // 9 x 32
constexpr int R_kf = R_Kl + NU * NX;        // 2 x 16
constexpr int R_Et = R_kf + 32;             // 3 x 41
constexpr int R_Y = R_Et + 3 * NZ;          // 41 x 3
constexpr int R_ew = R_Y + 3 * NZ;          // G_ang[3][7], J_ang[3][7]
constexpr int R_sufz = R_ew + 42;           // [7][3]
constexpr int R_dz2 = R_sufz + 24;          // sigma[2], r0[2], r1[2], zz[2]
constexpr int R_dx = R_dz2 + 8;
constexpr int R_dxn = R_dx + NX;
constexpr int R_r0 = R_dxn + NX;
constexpr int R_x1fix = R_r0 + NX;
constexpr int R_dzeta = R_x1fix + NX;       // 48
constexpr int R_dy = R_dzeta + ZPAD;        // 48
constexpr int R_red = R_dy + ZPAD;          // 64
constexpr int R_misc = R_red + 64;          // 64
constexpr int RIC_LDS_DOUBLES = R_misc + 64;

// scatter table entry of one record field: pass (0 none, 1 store, 2 add, 3 add when hess_mode),
// LDS offsets of the target and of its symmetric mirror (-1 = none)
struct TblEntry { int pass, o1, o2; };
inline void build_scatter_table(int* tbl /* 3*HREC */) {
    auto set = [&](int f, int pass, int o1, int o2) { tbl[3 * f] = pass; tbl[3 * f + 1] = o1; tbl[3 * f + 2] = o2; };
    for (int f = 0; f < HREC; f++) set(f, 0, -1, -1);
    auto Wo = [](int i, int j) { return R_W + i * LDW + j; };
    for (int c = 0; c < 5; c++)
        for (int i = 0; i < 7; i++) set(F_CD + c * 7 + i, 1, Wo(Z_Q + i, Z_D + c), Wo(Z_D + c, Z_Q + i));
    {
        int f = F_H17;
        for (int j = 0; j < 17; j++)
            for (int i = 0; i <= j; i++, f++) set(f, 1, Wo(pos17(i), pos17(j)), i == j ? -1 : Wo(pos17(j), pos17(i)));
    }
    const int spos[3] = {Z_PS, Z_RS, Z_D + 5};
    for (int sl = 0; sl < 3; sl++)
        for (int i = 0; i < 17; i++) set(F_C3 + sl * 17 + i, 2, Wo(pos17(i), spos[sl]), Wo(spos[sl], pos17(i)));
    for (int a = 0; a < 7; a++)
        for (int b = 0; b < 7; b++) {
            set(F_CQQ + a * 7 + b, 3, Wo(Z_Q + a, Z_Q + b), -1);
            set(F_CQD + a * 7 + b, 3, Wo(Z_Q + a, Z_DQ + b), Wo(Z_DQ + b, Z_Q + a));
        }
    for (int I = 0; I < 41; I++) {
        int pos = dg_pos(I);
        set(F_DG + 4 * I, 2, Wo(pos, pos), -1);
        set(F_DG + 4 * I + 1, 1, R_g0 + pos, -1);
        set(F_DG + 4 * I + 2, 1, R_g1 + pos, -1);
        set(F_DG + 4 * I + 3, 1, R_gz + pos, -1);
    }
    for (int i = 0; i < 8; i++) set(F_DZ2 + i, 1, R_dz2 + i, -1);
    for (int i = 0; i < 42; i++) set(F_EW + i, 1, R_ew + i, -1);
    for (int i = 0; i < 21; i++) set(F_SUFZ + i, 1, R_sufz + i, -1);
    for (int i = 0; i < 32; i++) set(F_RDEF + i, 1, R_rdef + i, -1);
}

BMPC_INL bool chol9r(const LDSD* W, double reg, double* Lc /*45 packed lower*/) {
    bool ok = true;
#define LI(i, j) Lc[(i) * ((i) + 1) / 2 + (j)]
    BMPC_UNROLL
    for (int j = 0; j < NU; j++) {
        double d = W[(NX + j) * LDW + NX + j] + reg;
        BMPC_UNROLL
        for (int l = 0; l < j; l++) d -= LI(j, l) * LI(j, l);
        if (!(d > 0)) { ok = false; d = 1.0; }
        d = sqrt(d);
        LI(j, j) = d;
        BMPC_UNROLL
        for (int i = j + 1; i < NU; i++) {
            double s = W[(NX + i) * LDW + NX + j];
            BMPC_UNROLL
            for (int l = 0; l < j; l++) s -= LI(i, l) * LI(j, l);
            LI(i, j) = s / d;
        }
    }
    return ok;
}
BMPC_INL void chol9r_solve(const double* Lc, double* b) {
    BMPC_UNROLL
    for (int i = 0; i < NU; i++) {
        double s = b[i];
        BMPC_UNROLL
        for (int l = 0; l < i; l++) s -= LI(i, l) * b[l];
        b[i] = s / LI(i, i);
    }
    BMPC_UNROLL
    for (int i = NU - 1; i >= 0; i--) {
        double s = b[i];
        BMPC_UNROLL
        for (int l = i + 1; l < NU; l++) s -= LI(l, i) * b[l];
        b[i] = s / LI(i, i);
    }
#undef LI
}

#define RL(x) (lds + (x))

// backward recursion over the horizon; returns false if a control block is not positive definite
BMPC_DEV bool ric_backward(const PipeArgs& A, LDSD* lds, int b, int lane, int hess_mode, double reg, double hreg,
                           double& lamsum_o, double& dual_o) {
    const int N = A.N;
    const DynC dc = make_dync(A.o.dt);
    bool ok = true;
    double lamsum = 0, dual = 0;
    if (lane < NX) { RL(R_lam)[lane] = 0; RL(R_pv0)[lane] = 0; RL(R_pv1)[lane] = 0; }
    constexpr int NF = (HREC + 63) / 64;
    int tps[NF], to1[NF], to2[NF];     // this lane's scatter-table entries (same for every stage)
    BMPC_UNROLL
    for (int i = 0; i < NF; i++) {
        int f = lane + 64 * i;
        tps[i] = (f < HREC) ? A.tbl[3 * f] : 0; to1[i] = (f < HREC) ? A.tbl[3 * f + 1] : 0; to2[i] = (f < HREC) ? A.tbl[3 * f + 2] : -1;
    }
    for (int k = N - 1; k >= 1; k--) {
        const bool term = (k == N - 1);
        const size_t pi = pair_of(A, b, k);
        const double* rec = A.hrec + pi * HREC;
        // ---- stage matrix from the record (natural coordinates) ----
        double rv[NF];
        BMPC_UNROLL
        for (int i = 0; i < NF; i++) { int f = lane + 64 * i; rv[i] = (f < HREC) ? rec[f] : 0.0; }
        for (int e = lane; e < NZ * LDW; e += 64) RL(R_W)[e] = (e / LDW == e % LDW) ? hreg : 0.0;
        BMPC_SYNC();
        BMPC_UNROLL
        for (int pass = 1; pass <= 3; pass++) {
            if (pass < 3 || hess_mode) {
                BMPC_UNROLL
                for (int i = 0; i < NF; i++) {
                    if (tps[i] == pass) {
                        if (pass == 1) { lds[to1[i]] = rv[i]; if (to2[i] >= 0) lds[to2[i]] = rv[i]; }
                        else { lds[to1[i]] += rv[i]; if (to2[i] >= 0) lds[to2[i]] += rv[i]; }
                    }
                }
            }
            BMPC_SYNC();
        }
        // ---- second-order term of the pi dynamics: multiplier lam_pi(k+1) times d2(dt w)/d(q,dq)2 ----
        if (hess_mode && !term) {
            const double l0 = dc.dt * RL(R_lam)[Z_PI], l1 = dc.dt * RL(R_lam)[Z_PI + 1], l2 = dc.dt * RL(R_lam)[Z_PI + 2];
            const double lamv[3] = {l0, l1, l2};
            auto zax = [&](int i, double* z) { z[0] = RL(R_ew)[21 + i]; z[1] = RL(R_ew)[28 + i]; z[2] = RL(R_ew)[35 + i]; };
            auto suf = [&](int m, double* s) {   // sufz[m], m = 1..7
                if (m >= 7) { s[0] = 0; s[1] = 0; s[2] = 0; }
                else { s[0] = RL(R_sufz)[3 * (m - 1)]; s[1] = RL(R_sufz)[3 * (m - 1) + 1]; s[2] = RL(R_sufz)[3 * (m - 1) + 2]; }
            };
            for (int e = lane; e < 98; e += 64) {
                if (e < 49) {
                    int a = e / 7, bq = e % 7;
                    double za[3], zb[3], sa[3], sm[3], u1[3] = {0, 0, 0}, u2[3], tmp[3];
                    zax(a, za); zax(bq, zb);
                    suf(a + 1, sa); suf((a > bq ? a : bq) + 1, sm);
                    if (bq < a) { cross3r(zb, za, tmp); cross3r(tmp, sa, u1); }
                    cross3r(zb, sm, tmp); cross3r(za, tmp, u2);
                    double acc = lamv[0] * (u1[0] + u2[0]) + lamv[1] * (u1[1] + u2[1]) + lamv[2] * (u1[2] + u2[2]);
                    RL(R_W)[(Z_Q + a) * LDW + Z_Q + bq] += acc;
                } else {
                    int i = (e - 49) / 7, j = (e - 49) % 7;
                    if (i < j) {
                        double zi[3], zj[3], zz[3];
                        zax(i, zi); zax(j, zj);
                        cross3r(zi, zj, zz);
                        double acc = lamv[0] * zz[0] + lamv[1] * zz[1] + lamv[2] * zz[2];
                        RL(R_W)[(Z_Q + i) * LDW + Z_DQ + j] += acc;
                        RL(R_W)[(Z_DQ + j) * LDW + Z_Q + i] += acc;
                    }
                }
            }
            BMPC_SYNC();
        }
        // ---- natural -> zeta coordinates: H = T^T Hy T (column pass, then row pass + vectors) ----
        for (int e = lane; e < NZ * 9; e += 64) {
            int i = e / 9, t = e % 9;
            LDSD* row = RL(R_W) + i * LDW;
            if (t < 7) row[Z_U + t] += dc.c3 * row[Z_Q + t] + dc.c2 * row[Z_DQ + t] + dc.c1 * row[Z_DDQ + t];
            else if (t == 7) row[Z_DRS] += 0.5 * dc.dt * row[Z_RS];
            else row[Z_DPS] += 0.5 * dc.dt * row[Z_PS];
        }
        BMPC_SYNC();
        for (int e = lane; e < NZ * 9 + 27; e += 64) {
            if (e < NZ * 9) {
                int j = e / 9, t = e % 9;
                LDSD* W = RL(R_W);
                if (t < 7) W[(Z_U + t) * LDW + j] += dc.c3 * W[(Z_Q + t) * LDW + j] + dc.c2 * W[(Z_DQ + t) * LDW + j] + dc.c1 * W[(Z_DDQ + t) * LDW + j];
                else if (t == 7) W[Z_DRS * LDW + j] += 0.5 * dc.dt * W[Z_RS * LDW + j];
                else W[Z_DPS * LDW + j] += 0.5 * dc.dt * W[Z_PS * LDW + j];
            } else {
                int vsel = (e - NZ * 9) / 9, t = (e - NZ * 9) % 9;
                LDSD* g = vsel == 0 ? RL(R_g0) : vsel == 1 ? RL(R_g1) : RL(R_gz);
                if (t < 7) g[Z_U + t] += dc.c3 * g[Z_Q + t] + dc.c2 * g[Z_DQ + t] + dc.c1 * g[Z_DDQ + t];
                else if (t == 7) g[Z_DRS] += 0.5 * dc.dt * g[Z_RS];
                else g[Z_DPS] += 0.5 * dc.dt * g[Z_PS];
            }
        }
        BMPC_SYNC();
        if (k == 1 && lane < 2) {   // zeta-diagonal rows rs~_1, ps~_1 >= 0
            int pos = lane ? Z_PS : Z_RS;
            RL(R_W)[pos * LDW + pos] += RL(R_dz2)[lane];
            RL(R_g0)[pos] -= RL(R_dz2)[2 + lane]; RL(R_g1)[pos] -= RL(R_dz2)[4 + lane]; RL(R_gz)[pos] -= RL(R_dz2)[6 + lane];
        }
        // ---- coupling with stage k+1 ----
        if (!term) {
            if (lane < NZ) {
                int c = lane;
                for (int a = 0; a < 3; a++) {
                    double v = 0;
                    if (c < Z_DQ) v = dc.dt * RL(R_ew)[7 * a + c];
                    else if (c < Z_DDQ) v = dc.dt * RL(R_ew)[21 + 7 * a + c - 7];
                    else if (c >= Z_U && c < Z_DRS) v = dc.dt * (dc.c3 * RL(R_ew)[7 * a + c - Z_U] + dc.c2 * RL(R_ew)[21 + 7 * a + c - Z_U]);
                    RL(R_Et)[a * NZ + c] = v;
                }
                PhiCol pc = phi_col(c, dc);
                for (int a = 0; a < 3; a++)
                    RL(R_Y)[c * 3 + a] = pc.c0 * RL(R_P)[pc.i0 * LDP + Z_PI + a] + pc.c1 * RL(R_P)[pc.i1 * LDP + Z_PI + a] +
                                         pc.c2 * RL(R_P)[pc.i2 * LDP + Z_PI + a];
            }
            BMPC_SYNC();
            if (lane < NX) {
                double v = RL(R_pv0)[lane];
                for (int j = 0; j < NX; j++) v += RL(R_P)[lane * LDP + j] * RL(R_rdef)[j];
                RL(R_vt0)[lane] = v;
            } else if (lane < 2 * NX) {
                RL(R_vt1)[lane - NX] = RL(R_pv1)[lane - NX];
            }
            BMPC_SYNC();
            {
                const double al[4][3] = {{1.0, 0.0, 0.0}, {dc.dt, 1.0, 0.0}, {0.5 * dc.dt * dc.dt, dc.dt, 1.0}, {dc.b3, dc.b2, dc.b1}};
                const int gpos[4] = {Z_Q, Z_DQ, Z_DDQ, Z_U};
                for (int e = lane; e < 49; e += 64) {
                    int a = e / 7, bq = e - 7 * a;
                    double Pb[3][3];
                    BMPC_UNROLL
                    for (int r = 0; r < 3; r++)
                        BMPC_UNROLL
                        for (int s = 0; s < 3; s++) Pb[r][s] = RL(R_P)[(7 * r + a) * LDP + 7 * s + bq];
                    BMPC_UNROLL
                    for (int gi = 0; gi < 4; gi++) {
                        double t0 = al[gi][0] * Pb[0][0] + al[gi][1] * Pb[1][0] + al[gi][2] * Pb[2][0];
                        double t1 = al[gi][0] * Pb[0][1] + al[gi][1] * Pb[1][1] + al[gi][2] * Pb[2][1];
                        double t2 = al[gi][0] * Pb[0][2] + al[gi][1] * Pb[1][2] + al[gi][2] * Pb[2][2];
                        BMPC_UNROLL
                        for (int gj = 0; gj < 4; gj++)
                            RL(R_W)[(gpos[gi] + a) * LDW + gpos[gj] + bq] += t0 * al[gj][0] + t1 * al[gj][1] + t2 * al[gj][2];
                    }
                }
                for (int e = lane; e < 7 * 11; e += 64) {
                    int a = e / 11, c = Z_PI + (e - 11 * a);
                    double p0 = RL(R_P)[a * LDP + c], p1 = RL(R_P)[(7 + a) * LDP + c], p2 = RL(R_P)[(14 + a) * LDP + c];
                    BMPC_UNROLL
                    for (int gi = 0; gi < 4; gi++) {
                        double v = al[gi][0] * p0 + al[gi][1] * p1 + al[gi][2] * p2;
                        int r = gpos[gi] + a;
                        RL(R_W)[r * LDW + c] += v; RL(R_W)[c * LDW + r] += v;
                        if (c == Z_RS || c == Z_PS) {
                            int cw = (c == Z_RS) ? Z_DRS : Z_DPS;
                            RL(R_W)[r * LDW + cw] += dc.dt * v; RL(R_W)[cw * LDW + r] += dc.dt * v;
                        }
                    }
                }
                for (int e = lane; e < 11 * 11; e += 64) {
                    int c1 = Z_PI + e / 11, c2 = Z_PI + e % 11;
                    double v = RL(R_P)[c1 * LDP + c2];
                    RL(R_W)[c1 * LDW + c2] += v;
                    bool s1 = (c1 == Z_RS || c1 == Z_PS), s2 = (c2 == Z_RS || c2 == Z_PS);
                    int w1 = (c1 == Z_RS) ? Z_DRS : Z_DPS, w2 = (c2 == Z_RS) ? Z_DRS : Z_DPS;
                    if (s2) RL(R_W)[c1 * LDW + w2] += dc.dt * v;
                    if (s1) RL(R_W)[w1 * LDW + c2] += dc.dt * v;
                    if (s1 && s2) RL(R_W)[w1 * LDW + w2] += dc.dt * dc.dt * v;
                }
            }
            BMPC_SYNC();
            for (int e = lane; e < NZ * 21; e += 64) {
                int i = e / 21, jj = e - 21 * i;
                int j = jj < 14 ? jj : Z_U + jj - 14;
                bool i_in = (i < Z_DDQ) || (i >= Z_U && i < Z_DRS);
                double ej0 = RL(R_Et)[j], ej1 = RL(R_Et)[NZ + j], ej2 = RL(R_Et)[2 * NZ + j];
                double v = RL(R_Y)[i * 3] * ej0 + RL(R_Y)[i * 3 + 1] * ej1 + RL(R_Y)[i * 3 + 2] * ej2;
                if (i_in) {
                    double ei0 = RL(R_Et)[i], ei1 = RL(R_Et)[NZ + i], ei2 = RL(R_Et)[2 * NZ + i];
                    const LDSD* Pp = RL(R_P) + Z_PI * LDP + Z_PI;
                    v += ei0 * RL(R_Y)[j * 3] + ei1 * RL(R_Y)[j * 3 + 1] + ei2 * RL(R_Y)[j * 3 + 2];
                    v += ei0 * (Pp[0] * ej0 + Pp[1] * ej1 + Pp[2] * ej2) + ei1 * (Pp[LDP] * ej0 + Pp[LDP + 1] * ej1 + Pp[LDP + 2] * ej2) +
                         ei2 * (Pp[2 * LDP] * ej0 + Pp[2 * LDP + 1] * ej1 + Pp[2 * LDP + 2] * ej2);
                    RL(R_W)[i * LDW + j] += v;
                } else {
                    RL(R_W)[i * LDW + j] += v;
                    RL(R_W)[j * LDW + i] += v;
                }
            }
            if (lane < NZ) {
                int c = lane;
                PhiCol pc = phi_col(c, dc);
                double gl = pc.c0 * RL(R_lam)[pc.i0] + pc.c1 * RL(R_lam)[pc.i1] + pc.c2 * RL(R_lam)[pc.i2];
                double a0 = pc.c0 * RL(R_vt0)[pc.i0] + pc.c1 * RL(R_vt0)[pc.i1] + pc.c2 * RL(R_vt0)[pc.i2];
                double a1 = pc.c0 * RL(R_vt1)[pc.i0] + pc.c1 * RL(R_vt1)[pc.i1] + pc.c2 * RL(R_vt1)[pc.i2];
                for (int a = 0; a < 3; a++) {
                    double ea = RL(R_Et)[a * NZ + c];
                    gl += ea * RL(R_lam)[Z_PI + a]; a0 += ea * RL(R_vt0)[Z_PI + a]; a1 += ea * RL(R_vt1)[Z_PI + a];
                }
                RL(R_gz)[c] += gl; RL(R_g0)[c] += a0; RL(R_g1)[c] += a1;
            }
        }
        BMPC_SYNC();
        // ---- adjoint multipliers + dual residual (gz now holds the Lagrangian gradient) ----
        if (lane < NZ) {
            double gl = RL(R_gz)[lane];
            if (lane >= NX || (k == 1 && lane >= 24)) dual = fmax(dual, fabs(gl));
            if (lane < NX) { RL(R_lam)[lane] = gl; lamsum += fabs(gl); }
        }
        // ---- control block factorisation, gains, Schur complement ----
        double Lc[45];
        if (!chol9r(RL(R_W), reg, Lc)) ok = false;
        double* krec = A.krec + pi * KREC;
        if (lane < NX + 2) {
            double rhs[NU];
            BMPC_UNROLL
            for (int l = 0; l < NU; l++)
                rhs[l] = (lane < NX) ? RL(R_W)[(NX + l) * LDW + lane] : (lane == NX ? RL(R_g0)[NX + l] : RL(R_g1)[NX + l]);
            chol9r_solve(Lc, rhs);
            BMPC_UNROLL
            for (int l = 0; l < NU; l++) {
                if (lane < NX) { RL(R_Kl)[l * NX + lane] = -rhs[l]; krec[l * NX + lane] = -rhs[l]; }
                else { RL(R_kf)[(lane - NX) * 16 + l] = -rhs[l]; krec[NU * NX + (lane - NX) * 16 + l] = -rhs[l]; }
            }
        }
        BMPC_SYNC();
        for (int e = lane; e < NX * NX; e += 64) {
            int i = e / NX, j = e % NX;
            if (j < i) continue;
            double v = RL(R_W)[i * LDW + j];
            for (int l = 0; l < NU; l++) v += RL(R_W)[(NX + l) * LDW + i] * RL(R_Kl)[l * NX + j];
            RL(R_P)[i * LDP + j] = v;
            RL(R_P)[j * LDP + i] = v;
        }
        {
            int i = lane & (NX - 1);
            const LDSD* g = (lane < NX) ? RL(R_g0) : RL(R_g1);
            const LDSD* kf = RL(R_kf) + ((lane < NX) ? 0 : 16);
            double v = g[i];
            for (int l = 0; l < NU; l++) v += RL(R_W)[(NX + l) * LDW + i] * kf[l];
            ((lane < NX) ? RL(R_pv0) : RL(R_pv1))[i] = v;
        }
        BMPC_SYNC();
    }
    lamsum_o = wg_sum(lamsum, RL(R_red), lane);
    dual_o = wg_max(dual, RL(R_red), lane);
    return ok;
}

// forward recursion: Newton direction dz for barrier parameter mu; false if the free part of the
// stage-1 value function is not positive definite
BMPC_DEV bool ric_forward(const PipeArgs& A, LDSD* lds, int b, int lane, double mu) {
    const int N = A.N;
    const DynC dc = make_dync(A.o.dt);
    bool ok = true;
    {
        double Pf[36], rhs[8];
        for (int i = 0; i < 8; i++) {
            double s = RL(R_pv0)[24 + i] + mu * RL(R_pv1)[24 + i];
            for (int j = 0; j < 24; j++) s += RL(R_P)[(24 + i) * LDP + j] * RL(R_r0)[j];
            rhs[i] = -s;
        }
#define PF(i, j) Pf[(i) * ((i) + 1) / 2 + (j)]
        for (int j = 0; j < 8; j++) {
            double d = RL(R_P)[(24 + j) * LDP + 24 + j];
            for (int l = 0; l < j; l++) d -= PF(j, l) * PF(j, l);
            if (!(d > 0)) { ok = false; d = 1.0; }
            d = sqrt(d);
            PF(j, j) = d;
            for (int i = j + 1; i < 8; i++) {
                double s = RL(R_P)[(24 + i) * LDP + 24 + j];
                for (int l = 0; l < j; l++) s -= PF(i, l) * PF(j, l);
                PF(i, j) = s / d;
            }
        }
        for (int i = 0; i < 8; i++) { double s = rhs[i]; for (int l = 0; l < i; l++) s -= PF(i, l) * rhs[l]; rhs[i] = s / PF(i, i); }
        for (int i = 7; i >= 0; i--) { double s = rhs[i]; for (int l = i + 1; l < 8; l++) s -= PF(l, i) * rhs[l]; rhs[i] = s / PF(i, i); }
#undef PF
        BMPC_SYNC();
        if (lane < 24) RL(R_dx)[lane] = RL(R_r0)[lane];
        if (lane == 0) for (int i = 0; i < 8; i++) RL(R_dx)[24 + i] = rhs[i];
        BMPC_SYNC();
    }
    if (!ok) return false;
    for (int k = 1; k < N; k++) {
        const size_t pi = pair_of(A, b, k);
        const double* krec = A.krec + pi * KREC;
        const double* rec = A.hrec + pi * HREC;
        for (int e = lane; e < NU * NX; e += 64) RL(R_Kl)[e] = krec[e];
        if (lane < 32) RL(R_kf)[lane] = krec[NU * NX + lane];
        if (lane < 42) RL(R_ew)[lane] = rec[F_EW + lane];
        if (lane < NX) RL(R_rdef)[lane] = rec[F_RDEF + lane];
        BMPC_SYNC();
        if (lane < NX) RL(R_dzeta)[lane] = RL(R_dx)[lane];
        else if (lane < NZ) {
            int l = lane - NX;
            double s = RL(R_kf)[l] + mu * RL(R_kf)[16 + l];
            for (int j = 0; j < NX; j++) s += RL(R_Kl)[l * NX + j] * RL(R_dx)[j];
            RL(R_dzeta)[lane] = s;
        }
        BMPC_SYNC();
        if (lane < NZ) {
            A.dz[(size_t)lane * A.NP + pi] = RL(R_dzeta)[lane];
            RL(R_dy)[lane] = nat_from_zeta(RL(R_dzeta), lane, dc);
        }
        BMPC_SYNC();
        if (k < N - 1) {
            if (lane < NX) {
                const LDSD* d = RL(R_dzeta);
                int i = lane;
                double v;
                if (i < Z_DQ) v = d[i] + dc.dt * d[i + 7] + 0.5 * dc.dt * dc.dt * d[i + 14] + dc.b3 * d[Z_U + i];
                else if (i < Z_DDQ) v = d[i] + dc.dt * d[i + 7] + dc.b2 * d[Z_U + i - 7];
                else if (i < Z_PI) v = d[i] + dc.b1 * d[Z_U + i - 14];
                else if (i < Z_RS) {
                    int a = i - Z_PI;
                    v = d[i];
                    for (int j = 0; j < 7; j++)
                        v += dc.dt * (RL(R_ew)[7 * a + j] * RL(R_dy)[Z_Q + j] + RL(R_ew)[21 + 7 * a + j] * RL(R_dy)[Z_DQ + j]);
                } else if (i == Z_RS) v = d[i] + dc.dt * d[Z_DRS];
                else if (i == Z_PS) v = d[i] + dc.dt * d[Z_DPS];
                else v = d[i];
                RL(R_dxn)[i] = v + RL(R_rdef)[i];
            }
            BMPC_SYNC();
            if (lane < NX) RL(R_dx)[lane] = RL(R_dxn)[lane];
        }
        BMPC_SYNC();
    }
    return true;
}

// lds: RIC_LDS_DOUBLES.  One workgroup (one wavefront) per entry of the eval list.
BMPC_DEV void k_ric_body(const PipeArgs& A, int blk, int lane, LDSD* lds) {
    const int count = A.L.cnt[0];
    if (blk >= count) return;
    const int b = A.L.eval[blk];
    const int N = A.N, n_w = 44 * N + 6;
    const SolverOpts& o = A.o;
    InstState* st = A.st + b;
    const double* lbx = A.lbx + (size_t)b * n_w;
    // pinned part of x_1 and its defect
    if (lane == 0) {
        double x1fix[24];
        x1fix_eval(lbx, N, o.dt, x1fix);
        for (int i = 0; i < 24; i++) RL(R_x1fix)[i] = x1fix[i];
    }
    BMPC_SYNC();
    if (lane < 24) RL(R_r0)[lane] = RL(R_x1fix)[lane] - A.zeta[(size_t)lane * A.NP + pair_of(A, b, 1)];
    // KKT partial sums of the pairs (fixed order)
    double cmax = 0, csum = 0, cmin = 1e300, zsum = 0, prim = 0, theta = 0, logs = 0, nrows = 0, fsum = 0;
    if (lane < N - 1) {
        const double* P = A.part + pair_of(A, b, 1) + lane;
        cmax = P[PT_CMAX * A.NP]; csum = P[PT_CSUM * A.NP]; cmin = P[PT_CMIN * A.NP]; zsum = P[PT_ZSUM * A.NP];
        prim = P[PT_PRIM * A.NP]; theta = P[PT_THETA * A.NP]; logs = P[PT_LOGS * A.NP]; nrows = P[PT_NROWS * A.NP];
        fsum = P[PT_FVAL * A.NP];
    }
    cmax = wg_max(cmax, RL(R_red), lane); csum = wg_sum(csum, RL(R_red), lane); cmin = wg_min(cmin, RL(R_red), lane);
    zsum = wg_sum(zsum, RL(R_red), lane); prim = wg_max(prim, RL(R_red), lane); theta = wg_sum(theta, RL(R_red), lane);
    logs = wg_sum(logs, RL(R_red), lane); nrows = wg_sum(nrows, RL(R_red), lane); fsum = wg_sum(fsum, RL(R_red), lane);

    int hess_mode = st->hess_mode, it = st->it, tries = 0;
    double hreg = st->hreg, mu = st->mu;
    const double reg = 1e-9;
    bool first = true;
    int status = -1;
    for (;;) {
        double lamsum, dual;
        bool ok = ric_backward(A, lds, b, lane, hess_mode, reg, hreg, lamsum, dual);
        if (first) {
            first = false;
            const int neq = NX * (N - 2) + 24;
            double sd = fmax(100.0, (lamsum + zsum) / ((double)neq + nrows)) / 100.0;
            double sc = fmax(100.0, zsum / nrows) / 100.0;
            double err = fmax(fmax(dual / sd, prim), cmax / sc);
            if (err <= o.tol && dual <= 1.0 && prim <= 1e-4 && cmax <= 1e-4) { status = 0; break; }
            if (it >= o.max_iter) { status = 1; break; }
            if (lane == 0) { st->err_prev = err; st->f0 = fsum; st->th0 = theta; st->ls0 = logs; }
            // monotone Fiacco-McCormick barrier update
            double emu = fmax(fmax(dual / sd, prim), fmax(fabs(cmax - mu), fabs(cmin - mu)) / sc);
            while (emu <= o.kappa_eps * mu && mu > o.tol / 10.0) {
                mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
                emu = fmax(fmax(dual / sd, prim), fmax(cmax - mu, 0.0) / sc);
            }
        }
        if (ok) ok = ric_forward(A, lds, b, lane, mu);
        if (ok) break;
        if (hess_mode) { hess_mode = 0; ++tries; }       // second-order terms not convex here: Gauss-Newton
        else {
            hreg = (hreg == 0.0) ? 1e-4 : hreg * 8;      // inertia correction (IPOPT delta_w)
            if (++tries > 12) { status = 3; break; }
        }
    }
    if (lane == 0) {
        if (status >= 0) {
            st->state = ST_DONE; st->status = status; st->fk = fsum;
            BMPC_ATOMIC_INC(A.L.cnt + 5);
        } else {
            if (tries == 0) hreg = (hreg < 1e-8) ? 0.0 : hreg / 3;
            st->hreg = hreg; st->mu = mu; st->tries = tries; st->state = ST_STEP;
            int pos = BMPC_ATOMIC_INC(A.L.cnt + 1);
            A.L.step[pos] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------
// per-instance control kernels (one thread per list entry)
// ------------------------------------------------------------------------------------------
BMPC_DEV void k_init_inst_body(const PipeArgs& A, int i) {
    if (i >= A.B) return;
    InstState* st = A.st + i;
    st->state = ST_EVAL; st->it = 0; st->status = 1; st->nfilt = 0; st->hess_mode = 0; st->bt = 0; st->armijo = 0; st->tries = 0;
    st->mu = A.o.mu_init; st->alpha = 0; st->ad = 0; st->ap = 0; st->hreg = 0; st->err_prev = 1e300; st->filt_mu = -1;
    st->theta_max = 1e300; st->theta_min = 0; st->fk = 0;
    A.L.eval[i] = i;
}

// after k_step: fraction-to-boundary step lengths, merit derivative, line-search start
BMPC_DEV void k_ls0_body(const PipeArgs& A, int i) {
    if (i >= A.L.cnt[1]) return;
    const int b = A.L.step[i], N = A.N;
    InstState* st = A.st + b;
    const double* P = A.part + pair_of(A, b, 1);
    double ap = 1.0, ad = 1.0, dbar = 0, dphif = 0;
    for (int k = 0; k < N - 1; k++) {
        ap = fmin(ap, P[PT_AP * A.NP + k]); ad = fmin(ad, P[PT_AD * A.NP + k]);
        dbar += P[PT_DBAR * A.NP + k]; dphif += P[PT_DPHIF * A.NP + k];
    }
    st->ap = ap; st->ad = ad;
    st->D = dphif + dbar;
    st->phi0 = st->f0 - st->mu * st->ls0;
    if (st->it == 0) { st->theta_max = 1e4 * fmax(1.0, st->th0); st->theta_min = 1e-4 * fmax(1.0, st->th0); }
    if (st->mu != st->filt_mu) { st->nfilt = 0; st->filt_mu = st->mu; }
    st->alpha = ap; st->bt = 0; st->armijo = 0;
    st->state = ST_TRIAL;
    int pos = BMPC_ATOMIC_INC(A.L.cnt + 2);
    A.L.trial[pos] = b;
}

// after k_trial: filter acceptance test
BMPC_DEV void k_ls_body(const PipeArgs& A, int i) {
    if (i >= A.L.cnt[2]) return;
    const int b = A.L.trial[i], N = A.N;
    InstState* st = A.st + b;
    const double* P = A.part + pair_of(A, b, 1);
    double f1 = 0, th1 = 0, ls1 = 0;
    for (int k = 0; k < N - 1; k++) { f1 += P[PT_F1 * A.NP + k]; th1 += P[PT_TH1 * A.NP + k]; ls1 += P[PT_LS1 * A.NP + k]; }
    const double mu = st->mu, th0 = st->th0, D = st->D, phi0 = st->phi0, alpha = st->alpha;
    double phi1 = f1 - mu * ls1;
    bool acc = (th1 <= st->theta_max);
    for (int j = 0; acc && j < st->nfilt; j++)
        if (th1 >= st->filt_th[j] && phi1 >= st->filt_phi[j]) acc = false;
    bool armijo_case = false;
    if (acc) {
        bool sw = (th0 <= st->theta_min) && (D < 0) && (alpha * pow(-D, 2.3) > pow(th0, 1.1));
        if (sw) { acc = (phi1 <= phi0 + 1e-4 * alpha * D + 1e-12 * fabs(phi0)); armijo_case = acc; }
        else acc = (th1 <= (1 - 1e-5) * th0) || (phi1 <= phi0 - 1e-5 * th0);
    }
    if (acc || st->bt >= 9) {
        if (!armijo_case) {
            const int MAXF = 8;
            int nf = st->nfilt;
            if (nf == MAXF) { for (int j = 0; j + 1 < MAXF; j++) { st->filt_th[j] = st->filt_th[j + 1]; st->filt_phi[j] = st->filt_phi[j + 1]; } nf--; }
            st->filt_th[nf] = (1 - 1e-5) * th0;
            st->filt_phi[nf] = phi0 - 1e-5 * th0;
            st->nfilt = nf + 1;
        }
        st->it += 1;
        st->hess_mode = (A.o.hess == 2 && st->err_prev < A.o.hess_switch) ? 1 : 0;
        st->state = ST_EVAL;
        int pos = BMPC_ATOMIC_INC(A.L.cnt + 3);
        A.L.eval_next[pos] = b;
    } else {
        st->alpha = 0.5 * alpha; st->bt += 1;
        int pos = BMPC_ATOMIC_INC(A.L.cnt + 4);
        A.L.trial_next[pos] = b;
    }
}

// rotate the list counters between super-steps (one thread)
BMPC_DEV void k_rotate_body(const PipeArgs& A) {
    int* c = A.L.cnt;
    c[0] = c[3]; c[3] = 0; c[1] = 0; c[2] = c[4]; c[4] = 0;
}

// per-instance outputs after k_out
BMPC_DEV void k_fin_body(const PipeArgs& A, int b) {
    if (b >= A.B) return;
    const InstState* st = A.st + b;
    const double* P = A.part + pair_of(A, b, 1);
    double v = 0;
    for (int k = 0; k < A.N - 1; k++) v += P[PT_F1 * A.NP + k];
    A.viol[b] = v;
    A.f[b] = st->fk;
    A.iters[b] = st->it;
    A.status[b] = st->status;
}

#undef RL
}  // namespace bmpc
