// bmpc_solver.hpp -- the per-instance interior-point loop (device code, one wavefront).
// Algorithm and notation: oracle/bmpc_solve.c (same steps, same constants); layout: DESIGN.md.
#pragma once
#include "bmpc_device.hpp"

// optional in-kernel phase timing (diagnostic builds only: -DBMPC_PROFILE; never in the product build)
#ifdef BMPC_PROFILE
// phase cycle sums are accumulated in LDS (misc[32..47]) and flushed once per instance
#define BMPC_PROF_START() long long prof_t0_ = clock64()
#ifndef BMPC_PROF_MASK
#define BMPC_PROF_MASK 0xffef   /* point 4 excluded: a lane-0 store right before the uniform hybrid-Hessian block changed results on gfx950 (open item, profiles/r01_pmc.md) */
#endif
#define BMPC_PROF(i) do { long long t1_ = clock64(); if (((BMPC_PROF_MASK >> (i)) & 1) && lane == 0) (lds + O_misc)[32 + (i)] += (double)(t1_ - prof_t0_); prof_t0_ = t1_; } while (0)
#else
#define BMPC_PROF_START() do {} while (0)
#define BMPC_PROF(i) do {} while (0)
#endif

namespace bmpc {

struct KktAcc {   // per-lane partial sums carried across a sweep
    double cmax, csum, cmin, zsum, prim, theta, logs, lamsum, dual;
    int nrows;
};
struct Kkt {
    double err, dual, prim, compl_, sd, sc, f, theta, avgc, minc, logs, cmu;
    int nrows;
};

struct WsPtr {
    double *zeta, *dz, *zsave, *t, *z, *dt, *dzr, *tsave, *K, *kf, *ev;
};
BMPC_INL void ws_carve(double* b, int N, WsPtr& w) {
    w.zeta = b; b += N * ZPAD; w.dz = b; b += N * ZPAD; w.zsave = b; b += N * ZPAD;
    w.t = b; b += N * NSLOT; w.z = b; b += N * NSLOT; w.dt = b; b += N * NSLOT;
    w.dzr = b; b += N * NSLOT; w.tsave = b; b += N * NSLOT;
    w.K = b; b += N * NU * NX; w.kf = b; b += N * 32; w.ev = b;
}

BMPC_INL double sp_b(const double* sp, int off, int row, int stride, int col) { return sp[off + row * stride + col]; }

// dynamics defect of stage k (needs zeta_k in (lds + O_zeta), zeta_{k+1} in (lds + O_znext), v in rc)
BMPC_INL double defect_row(LDSD* lds, const DynC d, int i) {
    const LDSD* z = (lds + O_zeta);
    double v;
    if (i < Z_DQ) v = z[i] + d.dt * z[i + 7] + 0.5 * d.dt * d.dt * z[i + 14] + d.b3 * z[Z_U + i];
    else if (i < Z_DDQ) v = z[i] + d.dt * z[i + 7] + d.b2 * z[Z_U + i - 7];
    else if (i < Z_PI) v = z[i] + d.b1 * z[Z_U + i - 14];
    else if (i < Z_RS) v = z[i] + d.dt * (lds + O_rc)[RC_V + 3 + (i - Z_PI)];
    else if (i == Z_RS) v = z[i] + d.dt * z[Z_DRS];
    else if (i == Z_PS) v = z[i] + d.dt * z[Z_DPS];
    else v = z[i];
    return v - (lds + O_znext)[i];
}

// 9x9 Cholesky in registers (every lane redundantly); returns false when not positive definite
BMPC_INL bool chol9(const LDSD* W, double reg, double* Lc /*45 packed lower*/) {
    bool ok = true;
#define LI(i, j) Lc[(i) * ((i) + 1) / 2 + (j)]
#pragma unroll
    for (int j = 0; j < NU; j++) {
        double d = W[(NX + j) * LDW + NX + j] + reg;
#pragma unroll
        for (int l = 0; l < j; l++) d -= LI(j, l) * LI(j, l);
        if (!(d > 0)) { ok = false; d = 1.0; }
        d = sqrt(d);
        LI(j, j) = d;
#pragma unroll
        for (int i = j + 1; i < NU; i++) {
            double s = W[(NX + i) * LDW + NX + j];
#pragma unroll
            for (int l = 0; l < j; l++) s -= LI(i, l) * LI(j, l);
            LI(i, j) = s / d;
        }
    }
    return ok;
}
BMPC_INL void chol9_solve(const double* Lc, double* b) {
#pragma unroll
    for (int i = 0; i < NU; i++) {
        double s = b[i];
#pragma unroll
        for (int l = 0; l < i; l++) s -= LI(i, l) * b[l];
        b[i] = s / LI(i, i);
    }
#pragma unroll
    for (int i = NU - 1; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int l = i + 1; l < NU; l++) s -= LI(l, i) * b[l];
        b[i] = s / LI(i, i);
    }
#undef LI
}

// ------------------------------------------------------------------------------------------
// Backward sweep: evaluation + KKT error + adjoint multipliers + Riccati factorisation with
// two right-hand sides (g = g0 + mu g1).  Returns false if a control block is not PD.
// ------------------------------------------------------------------------------------------
BMPC_NOINL bool backward_sweep(const Inst I, LDSD* lds, const WsPtr ws, const DynC dc, int lane,
                             double ad_pend, const double* iw0, double reg, double hreg, int hess_mode, Kkt& kk) {
    const int N = I.N;
    const LDSD* sp = (lds + O_sp);
    const LDSD* wts = sp + P_W;
    KktAcc ac;
    ac.cmax = 0; ac.csum = 0; ac.cmin = 1e300; ac.zsum = 0; ac.prim = 0; ac.theta = 0; ac.logs = 0;
    ac.lamsum = 0; ac.dual = 0; ac.nrows = 0;
    double fsum = 0;
    bool ok = true;
    if (lane < NX) { (lds + O_lam)[lane] = 0; (lds + O_pv0)[lane] = 0; (lds + O_pv1)[lane] = 0; }
    for (int k = N - 1; k >= 1; k--) {
        const bool term = (k == N - 1);
        BMPC_PROF_START();
        if (lane < NZ) (lds + O_zeta)[lane] = ws.zeta[k * ZPAD + lane];
        for (int e = lane; e < NZ * LDW; e += BMPC_NT) (lds + O_W)[e] = (e / LDW == e % LDW) ? hreg : 0.0;
        if (lane < ZPAD) { (lds + O_g0)[lane] = 0; (lds + O_g1)[lane] = 0; (lds + O_gz)[lane] = 0; }
        BMPC_SYNC();
        // the point was evaluated by the sweep that produced it (initialisation or accepted trial):
        // load that evaluation, add only the Hessian blocks
        for (int e = lane; e < EVAL_DOUBLES; e += BMPC_NT) (lds + O_yz)[e] = ws.ev[(size_t)k * WS_EVAL + e];
        BMPC_SYNC();
        fsum += stage_eval(I, lds, dc, k, lane, 2, iw0);
        BMPC_PROF(0);
        // ---- rows: slack/multiplier data, KKT partial sums ----
        for (int s = lane; s < NSLOT; s += BMPC_NT) {
            Row r;
            row_eval(I, lds, k, s, r);
            double sg = 0, r0 = 0, r1 = 0, zz = 0;
            if (r.kind) {
                double t = ws.t[k * NSLOT + s];
                zz = ws.z[k * NSLOT + s];
                if (ad_pend != 0.0) { zz += ad_pend * ws.dzr[k * NSLOT + s]; ws.z[k * NSLOT + s] = zz; }
                sg = zz / t; r0 = sg * (r.h + t); r1 = 1.0 / t;
                double c = t * zz;
                ac.cmax = fmax(ac.cmax, c); ac.csum += c; ac.cmin = fmin(ac.cmin, c); ac.zsum += zz;
                ac.prim = fmax(ac.prim, fabs(r.h + t)); ac.theta += fabs(r.h + t); ac.logs += log(t);
                ac.nrows++;
            }
            (lds + O_rowS)[s] = sg; (lds + O_rowS)[NSLOT + s] = r0; (lds + O_rowS)[2 * NSLOT + s] = r1; (lds + O_rowS)[3 * NSLOT + s] = zz;
            int pr = pose_row_index(s);
            if (pr >= 0) {
                for (int c = 0; c < 6; c++) (lds + O_rowA)[pr * 6 + c] = r.a[c];
                (lds + O_rowSl)[pr] = (r.kind == 3) ? (double)r.sel : -1.0;
            }
        }
        BMPC_SYNC();
        BMPC_PROF(1);
        // ---- group accumulations (one output per lane, serial over rows) ----
        {
            // every output is  sum_rows  sign * scalar[row] * fa[row] * fb[row] * mask[row]  (branch-free)
            for (int o = lane; o < 69; o += BMPC_NT) {
                // outputs: 0..20 M6 (sym), 21..38 mS[3][6], 39..41 sS, 42..59 b{0,1,z}[6], 60..68 bS{0,1,z}[3]
                int ia = 0, ib = 0, sel = 0, vecsel = 0, soff = 0;
                bool useA = false, useB = false;
                double sgn = 1.0;
                if (o < 21) {
                    int e = o, i = 0;
                    if (e >= 6) { e -= 6; i = 1; } if (i == 1 && e >= 5) { e -= 5; i = 2; } if (i == 2 && e >= 4) { e -= 4; i = 3; }
                    if (i == 3 && e >= 3) { e -= 3; i = 4; } if (i == 4 && e >= 2) { e -= 2; i = 5; }
                    ia = i; ib = i + e; useA = true; useB = true;
                } else if (o < 39) { sel = (o - 21) / 6 + 1; ia = (o - 21) % 6; useA = true; sgn = -1.0; }
                else if (o < 42) { sel = o - 39 + 1; }
                else if (o < 60) { vecsel = (o - 42) / 6; ia = (o - 42) % 6; useA = true; soff = (vecsel + 1) * NSLOT; }
                else { vecsel = (o - 60) / 3; sel = (o - 60) % 3 + 1; sgn = -1.0; soff = (vecsel + 1) * NSLOT; }
                double acc = 0;
                const double dsel = (double)sel;
#pragma unroll 4
                for (int pr = 0; pr < NPOSE; pr++) {
                    int s = pr + (pr < 21 ? S_EE : S_PHI - 21);
                    double v = (lds + O_rowS)[soff + s];
                    double fa = useA ? (lds + O_rowA)[pr * 6 + ia] : 1.0;
                    double fb = useB ? (lds + O_rowA)[pr * 6 + ib] : 1.0;
                    double mk = (sel == 0 || (lds + O_rowSl)[pr] == dsel) ? 1.0 : 0.0;
                    acc += v * fa * fb * mk;
                }
                acc *= sgn;
                if (o < 21) { (lds + O_Hp)[6 * ia + ib] += acc; if (ia != ib) (lds + O_Hp)[6 * ib + ia] += acc; }
                else if (o < 39) (lds + O_mS)[(sel - 1) * 6 + ia] = acc;
                else if (o < 42) (lds + O_sS)[sel - 1] = acc;
                else if (o < 60) { LDSD* dst = vecsel == 0 ? (lds + O_bp0) : vecsel == 1 ? (lds + O_bp1) : (lds + O_bpz); dst[ia] += acc; }
                else { LDSD* dst = vecsel == 0 ? (lds + O_bS0) : vecsel == 1 ? (lds + O_bS1) : (lds + O_bSz); dst[sel - 1] = acc; }
            }
            for (int o = lane; o < 132; o += BMPC_NT) {
                // per point c: 0..5 M3 sym, 6..8 mc, 9 sc, 10..18 b3{0,1,z}[3], 19..21 bc{0,1,z}
                int c = o / 22, q = o - 22 * c;
                const LDSD* a = sp + SP_ASETJ + 45 * c;
                int ia = 0, ib = 0, vs = 0, soff = 0;
                bool useA = false, useB = false;
                double sgn = 1.0;
                if (q < 6) { int e = q, i = 0; if (e >= 3) { e -= 3; i = 1; } if (i == 1 && e >= 2) { e -= 2; i = 2; } ia = i; ib = i + e; useA = true; useB = true; }
                else if (q < 9) { ia = q - 6; useA = true; sgn = -1.0; }
                else if (q == 9) { }
                else if (q < 19) { vs = (q - 10) / 3; ia = (q - 10) % 3; useA = true; soff = (vs + 1) * NSLOT; }
                else { vs = q - 19; sgn = -1.0; soff = (vs + 1) * NSLOT; }
                double acc = 0;
#pragma unroll 5
                for (int rr = 0; rr < 15; rr++) {
                    double v = (lds + O_rowS)[soff + S_COL + 15 * c + rr];
                    double fa = useA ? a[rr + 15 * ia] : 1.0;
                    double fb = useB ? a[rr + 15 * ib] : 1.0;
                    acc += v * fa * fb;
                }
                acc *= sgn;
                if (q < 6) { (lds + O_M3)[9 * c + 3 * ia + ib] = acc; (lds + O_M3)[9 * c + 3 * ib + ia] = acc; }
                else if (q < 9) (lds + O_mc)[3 * c + ia] = acc;
                else if (q == 9) (lds + O_sc)[c] = acc;
                else if (q < 19) { LDSD* dst = vs == 0 ? (lds + O_b30) : vs == 1 ? (lds + O_b31) : (lds + O_b3z); dst[3 * c + ia] = acc; }
                else { LDSD* dst = vs == 0 ? (lds + O_bc0) : vs == 1 ? (lds + O_bc1) : (lds + O_bcz); dst[c] = acc; }
            }
            // natural-diagonal rows straight into W / g
            if (lane < 38) {
                int s0, s1, pos;
                if (lane < 28) { s0 = 2 * lane; s1 = 2 * lane + 1; int blk = lane / 7, jj = lane - 7 * blk; pos = (blk == 0 ? Z_Q : blk == 1 ? Z_DQ : blk == 2 ? Z_DDQ : Z_U) + jj; }
                else if (lane < 32) { s0 = S_NONNEG + lane - 28; s1 = -1; int m = lane - 28; pos = (m == 0 ? Z_RS : m == 1 ? Z_DRS : m == 2 ? Z_PS : Z_DPS); }
                else { s0 = S_D1 + lane - 32; s1 = -1; pos = Z_D + lane - 32; }
                double c0 = (lane < 28) ? 1.0 : -1.0;   // upper rows +1, all others -1
                double h = (lds + O_rowS)[s0], a0 = c0 * (lds + O_rowS)[NSLOT + s0], a1 = c0 * (lds + O_rowS)[2 * NSLOT + s0], az = c0 * (lds + O_rowS)[3 * NSLOT + s0];
                if (s1 >= 0) { h += (lds + O_rowS)[s1]; a0 -= (lds + O_rowS)[NSLOT + s1]; a1 -= (lds + O_rowS)[2 * NSLOT + s1]; az -= (lds + O_rowS)[3 * NSLOT + s1]; }
                (lds + O_W)[pos * LDW + pos] += h; (lds + O_g0)[pos] += a0; (lds + O_g1)[pos] += a1; (lds + O_gz)[pos] += az;
            }
        }
        BMPC_SYNC();
        BMPC_PROF(2);
        // ---- chain pose/velocity space through O = d(pose, v)/d(q, dq, pi) ----
        for (int e = lane; e < 204; e += BMPC_NT) {
            int mat = e / 102, rr = (e % 102) / 17, cc = e % 17;
            double v = 0;
            if (mat == 0) {   // Op: rows p_pos (J lin) ; p_rot (dt/2 G_w | dt/2 J_w | I)
                if (rr < 3) v = (cc < 7) ? (lds + O_J)[7 * rr + cc] : 0.0;
                else if (cc < 7) v = 0.5 * dc.dt * (lds + O_G)[7 * rr + cc];
                else if (cc < 14) v = 0.5 * dc.dt * (lds + O_J)[7 * rr + cc - 7];
                else v = (cc - 14 == rr - 3) ? 1.0 : 0.0;
                (lds + O_Op)[e] = v;
            } else {          // Ov: [G | J | 0]
                v = (cc < 7) ? (lds + O_G)[7 * rr + cc] : (cc < 14 ? (lds + O_J)[7 * rr + cc - 7] : 0.0);
                (lds + O_Ov)[e - 102] = v;
            }
        }
        BMPC_SYNC();
        for (int e = lane; e < 204; e += BMPC_NT) {
            int mat = e / 102, rr = (e % 102) / 17, cc = e % 17;
            const LDSD* H = mat ? (lds + O_Hv) : (lds + O_Hp);
            const LDSD* O = mat ? (lds + O_Ov) : (lds + O_Op);
            double v = 0;
            for (int a = 0; a < 6; a++) v += H[6 * rr + a] * O[17 * a + cc];
            (mat ? (lds + O_T2) : (lds + O_T1))[17 * rr + cc] = v;
        }
        BMPC_SYNC();
        {
            auto pos17 = [](int i) { return i < 7 ? Z_Q + i : (i < 14 ? Z_DQ + i - 7 : Z_PI + i - 14); };
            const int spos[3] = {Z_PS, Z_RS, Z_D + 5};
            for (int e = lane; e < 289 + 51 + 3 + 20; e += BMPC_NT) {
                if (e < 289) {
                    int i = e / 17, j = e % 17;
                    double v = 0;
                    for (int a = 0; a < 6; a++) v += (lds + O_Op)[17 * a + i] * (lds + O_T1)[17 * a + j] + (lds + O_Ov)[17 * a + i] * (lds + O_T2)[17 * a + j];
                    (lds + O_W)[pos17(i) * LDW + pos17(j)] += v;
                } else if (e < 340) {
                    int i = (e - 289) / 3, sl = (e - 289) % 3;
                    double v = 0;
                    for (int a = 0; a < 6; a++) v += (lds + O_Op)[17 * a + i] * (lds + O_mS)[6 * sl + a];
                    (lds + O_W)[pos17(i) * LDW + spos[sl]] += v;
                    (lds + O_W)[spos[sl] * LDW + pos17(i)] += v;
                } else if (e < 343) {
                    int sl = e - 340;
                    (lds + O_W)[spos[sl] * LDW + spos[sl]] += (lds + O_sS)[sl];
                    (lds + O_g0)[spos[sl]] += (lds + O_bS0)[sl]; (lds + O_g1)[spos[sl]] += (lds + O_bS1)[sl]; (lds + O_gz)[spos[sl]] += (lds + O_bSz)[sl];
                } else if (e < 360) {
                    int i = e - 343;
                    double v0 = 0, v1 = 0, vz = 0;
                    for (int a = 0; a < 6; a++) {
                        v0 += (lds + O_Op)[17 * a + i] * (lds + O_bp0)[a] + (lds + O_Ov)[17 * a + i] * (lds + O_bv)[a];
                        v1 += (lds + O_Op)[17 * a + i] * (lds + O_bp1)[a];
                        vz += (lds + O_Op)[17 * a + i] * (lds + O_bpz)[a] + (lds + O_Ov)[17 * a + i] * (lds + O_bv)[a];
                    }
                    (lds + O_g0)[pos17(i)] += v0; (lds + O_g1)[pos17(i)] += v1; (lds + O_gz)[pos17(i)] += vz;
                }
            }
        }
        BMPC_SYNC();
        BMPC_PROF(3);
        // ---- collision points: q x q, q x d, d x d ----
        for (int e = lane; e < 49 + 42 + 6 + 7; e += BMPC_NT) {
            if (e < 49) {
                int i = e / 7, j = e % 7;
                double v = 0;
                for (int c = 0; c < 6; c++)
                    for (int a = 0; a < 3; a++) {
                        double t = 0;
                        for (int bb = 0; bb < 3; bb++) t += (lds + O_M3)[9 * c + 3 * a + bb] * (lds + O_Jp)[21 * c + 7 * bb + j];
                        v += (lds + O_Jp)[21 * c + 7 * a + i] * t;
                    }
                (lds + O_W)[(Z_Q + i) * LDW + Z_Q + j] += v;
            } else if (e < 91) {
                int i = (e - 49) / 6, c = (e - 49) % 6;
                double v = 0;
                for (int a = 0; a < 3; a++) v += (lds + O_Jp)[21 * c + 7 * a + i] * (lds + O_mc)[3 * c + a];
                (lds + O_W)[(Z_Q + i) * LDW + Z_D + c] += v;
                (lds + O_W)[(Z_D + c) * LDW + Z_Q + i] += v;
            } else if (e < 97) {
                int c = e - 91;
                (lds + O_W)[(Z_D + c) * LDW + Z_D + c] += (lds + O_sc)[c];
                (lds + O_g0)[Z_D + c] += (lds + O_bc0)[c]; (lds + O_g1)[Z_D + c] += (lds + O_bc1)[c]; (lds + O_gz)[Z_D + c] += (lds + O_bcz)[c];
            } else {
                int i = e - 97;
                double v0 = 0, v1 = 0, vz = 0;
                for (int c = 0; c < 6; c++)
                    for (int a = 0; a < 3; a++) {
                        double jp = (lds + O_Jp)[21 * c + 7 * a + i];
                        v0 += jp * (lds + O_b30)[3 * c + a]; v1 += jp * (lds + O_b31)[3 * c + a]; vz += jp * (lds + O_b3z)[3 * c + a];
                    }
                (lds + O_g0)[Z_Q + i] += v0; (lds + O_g1)[Z_Q + i] += v1; (lds + O_gz)[Z_Q + i] += vz;
            }
        }
        BMPC_SYNC();
        BMPC_PROF(4);
        // ---- second-order kinematic terms of the Lagrangian Hessian (hybrid mode) ----
        if (hess_mode) {
            if (lane < 27) {   // generalised forces: on p_ee (3), on v (6), on the 6 collision points (18)
                double v;
                if (lane < 3) v = (lds + O_bpz)[lane];
                else if (lane < 9) {
                    int a = lane - 3;
                    v = (lds + O_bv)[a];
                    if (a >= 3) { v += 0.5 * dc.dt * (lds + O_bpz)[a]; if (!term) v += dc.dt * (lds + O_lam)[Z_PI + a - 3]; }
                } else v = (lds + O_b3z)[lane - 9];
                (lds + O_misc)[lane] = v;
            }
            BMPC_SYNC();
            const int njc[6] = {2, 3, 4, 5, 6, 4};
            for (int e = lane; e < 98; e += BMPC_NT) {
                const LDSD* Fp = (lds + O_misc); const LDSD* Fv = (lds + O_misc) + 3; const LDSD* Fc = (lds + O_misc) + 9;
                if (e < 49) {          // q_a x q_b
                    int a = e / 7, bq = e % 7, m = a < bq ? a : bq, M = a < bq ? bq : a;
                    double zm[3] = {(lds + O_zax)[3 * m], (lds + O_zax)[3 * m + 1], (lds + O_zax)[3 * m + 2]};
                    double cM[3] = {(lds + O_J)[M], (lds + O_J)[7 + M], (lds + O_J)[14 + M]}, zc[3];
                    cross3(zm, cM, zc);
                    double acc = dot3(Fp, zc);
                    for (int c = 0; c < 6; c++)
                        if (M < njc[c]) {
                            double cc[3] = {(lds + O_Jp)[21 * c + M], (lds + O_Jp)[21 * c + 7 + M], (lds + O_Jp)[21 * c + 14 + M]};
                            cross3(zm, cc, zc);
                            acc += dot3(Fc + 3 * c, zc);
                        }
                    // q-q block of the v = J(q) dq curvature (third-order kinematics times dq)
                    for (int j = 0; j < 7; j++) {
                        double dqj = (lds + O_yz)[Z_DQ + j];
                        if (dqj == 0.0) continue;
                        int m1 = a < j ? a : j, M1 = a < j ? j : a;
                        double z1[3] = {(lds + O_zax)[3 * m1], (lds + O_zax)[3 * m1 + 1], (lds + O_zax)[3 * m1 + 2]};
                        double c1[3] = {(lds + O_J)[M1], (lds + O_J)[7 + M1], (lds + O_J)[14 + M1]};
                        double t1[3] = {0, 0, 0}, t2[3], dzm[3], dcM[3];
                        if (bq < m1) { double zb[3] = {(lds + O_zax)[3 * bq], (lds + O_zax)[3 * bq + 1], (lds + O_zax)[3 * bq + 2]}; cross3(zb, z1, dzm); cross3(dzm, c1, t1); }
                        int m2 = bq < M1 ? bq : M1, M2 = bq < M1 ? M1 : bq;
                        double z2[3] = {(lds + O_zax)[3 * m2], (lds + O_zax)[3 * m2 + 1], (lds + O_zax)[3 * m2 + 2]};
                        double c2[3] = {(lds + O_J)[M2], (lds + O_J)[7 + M2], (lds + O_J)[14 + M2]};
                        cross3(z2, c2, dcM);
                        cross3(z1, dcM, t2);
                        double lin = Fv[0] * (t1[0] + t2[0]) + Fv[1] * (t1[1] + t2[1]) + Fv[2] * (t1[2] + t2[2]);
                        double ang = 0;
                        if (a < j) {
                            double za[3] = {(lds + O_zax)[3 * a], (lds + O_zax)[3 * a + 1], (lds + O_zax)[3 * a + 2]};
                            double zj[3] = {(lds + O_zax)[3 * j], (lds + O_zax)[3 * j + 1], (lds + O_zax)[3 * j + 2]};
                            double zb[3] = {(lds + O_zax)[3 * bq], (lds + O_zax)[3 * bq + 1], (lds + O_zax)[3 * bq + 2]};
                            double u1[3] = {0, 0, 0}, u2[3] = {0, 0, 0}, tmp[3];
                            if (bq < a) { cross3(zb, za, tmp); cross3(tmp, zj, u1); }
                            if (bq < j) { cross3(zb, zj, tmp); cross3(za, tmp, u2); }
                            ang = Fv[3] * (u1[0] + u2[0]) + Fv[4] * (u1[1] + u2[1]) + Fv[5] * (u1[2] + u2[2]);
                        }
                        acc += dqj * (lin + ang);
                    }
                    (lds + O_W)[(Z_Q + a) * LDW + Z_Q + bq] += acc;
                } else {               // q_i x dq_j  (d2 v / dq_i d dq_j = dJ[:, j]/dq_i)
                    int i = (e - 49) / 7, j = (e - 49) % 7, m = i < j ? i : j, M = i < j ? j : i;
                    double zm[3] = {(lds + O_zax)[3 * m], (lds + O_zax)[3 * m + 1], (lds + O_zax)[3 * m + 2]};
                    double cM[3] = {(lds + O_J)[M], (lds + O_J)[7 + M], (lds + O_J)[14 + M]}, zc[3];
                    cross3(zm, cM, zc);
                    double acc = dot3(Fv, zc);
                    if (i < j) {
                        double zi[3] = {(lds + O_zax)[3 * i], (lds + O_zax)[3 * i + 1], (lds + O_zax)[3 * i + 2]};
                        double zj[3] = {(lds + O_zax)[3 * j], (lds + O_zax)[3 * j + 1], (lds + O_zax)[3 * j + 2]}, zz[3];
                        cross3(zi, zj, zz);
                        acc += dot3(Fv + 3, zz);
                    }
                    (lds + O_W)[(Z_Q + i) * LDW + Z_DQ + j] += acc;
                    (lds + O_W)[(Z_DQ + j) * LDW + Z_Q + i] += acc;
                }
            }
            BMPC_SYNC();
        }
        BMPC_PROF(5);
        // ---- direct quadratic cost terms (natural coordinates) ----
        if (lane < 20) {
            int pos; double w2; double val; double extra = 0;
            if (lane < 3) { pos = Z_DQ + 2 + lane; w2 = 2 * wts[6]; }
            else if (lane < 10) { pos = Z_U + lane - 3; w2 = 2 * wts[7]; }
            else if (lane < 14) { int m = lane - 10; pos = (m == 0 ? Z_RS : m == 1 ? Z_DRS : m == 2 ? Z_PS : Z_DPS); w2 = 2 * ((m & 1) ? wts[10] : wts[9]); }
            else { int i = lane - 14; pos = Z_D + i; w2 = term ? (2 * wts[10] + (i != 4 ? 2 * wts[8] : 0.0)) : 0.0; extra = (term && i != 4) ? 2 * wts[8] * sp[P_SLACKS0 + i] : 0.0; }
            val = w2 * (lds + O_yz)[pos] + extra;
            (lds + O_W)[pos * LDW + pos] += w2; (lds + O_g0)[pos] += val; (lds + O_gz)[pos] += val;
        }
        BMPC_SYNC();
        BMPC_PROF(6);
        // ---- natural -> zeta coordinates: H = T^T Hy T (column pass, then row pass + vectors) ----
        for (int e = lane; e < NZ * 9; e += BMPC_NT) {
            int i = e / 9, t = e % 9;
            LDSD* row = (lds + O_W) + i * LDW;
            if (t < 7) row[Z_U + t] += dc.c3 * row[Z_Q + t] + dc.c2 * row[Z_DQ + t] + dc.c1 * row[Z_DDQ + t];
            else if (t == 7) row[Z_DRS] += 0.5 * dc.dt * row[Z_RS];
            else row[Z_DPS] += 0.5 * dc.dt * row[Z_PS];
        }
        BMPC_SYNC();
        for (int e = lane; e < NZ * 9 + 27; e += BMPC_NT) {
            if (e < NZ * 9) {
                int j = e / 9, t = e % 9;
                LDSD* W = (lds + O_W);
                if (t < 7) W[(Z_U + t) * LDW + j] += dc.c3 * W[(Z_Q + t) * LDW + j] + dc.c2 * W[(Z_DQ + t) * LDW + j] + dc.c1 * W[(Z_DDQ + t) * LDW + j];
                else if (t == 7) W[Z_DRS * LDW + j] += 0.5 * dc.dt * W[Z_RS * LDW + j];
                else W[Z_DPS * LDW + j] += 0.5 * dc.dt * W[Z_PS * LDW + j];
            } else {
                int vsel = (e - NZ * 9) / 9, t = (e - NZ * 9) % 9;
                LDSD* g = vsel == 0 ? (lds + O_g0) : vsel == 1 ? (lds + O_g1) : (lds + O_gz);
                if (t < 7) g[Z_U + t] += dc.c3 * g[Z_Q + t] + dc.c2 * g[Z_DQ + t] + dc.c1 * g[Z_DDQ + t];
                else if (t == 7) g[Z_DRS] += 0.5 * dc.dt * g[Z_RS];
                else g[Z_DPS] += 0.5 * dc.dt * g[Z_PS];
            }
        }
        BMPC_SYNC();
        if (k == 1 && lane < 2) {   // zeta-diagonal rows rs~_1, ps~_1 >= 0
            int s = S_RS1 + lane, pos = lane ? Z_PS : Z_RS;
            (lds + O_W)[pos * LDW + pos] += (lds + O_rowS)[s];
            (lds + O_g0)[pos] -= (lds + O_rowS)[NSLOT + s]; (lds + O_g1)[pos] -= (lds + O_rowS)[2 * NSLOT + s]; (lds + O_gz)[pos] -= (lds + O_rowS)[3 * NSLOT + s];
        }
        BMPC_PROF(7);
        // ---- coupling with stage k+1 ----
        if (!term) {
            if (lane < NZ) {
                int c = lane;
                for (int a = 0; a < 3; a++) {
                    double v = 0;
                    if (c < Z_DQ) v = dc.dt * (lds + O_G)[7 * (3 + a) + c];
                    else if (c < Z_DDQ) v = dc.dt * (lds + O_J)[7 * (3 + a) + c - 7];
                    else if (c >= Z_U && c < Z_DRS) v = dc.dt * (dc.c3 * (lds + O_G)[7 * (3 + a) + c - Z_U] + dc.c2 * (lds + O_J)[7 * (3 + a) + c - Z_U]);
                    (lds + O_Et)[a * NZ + c] = v;
                }
                PhiCol pc = phi_col(c, dc);
                for (int a = 0; a < 3; a++)
                    (lds + O_Y)[c * 3 + a] = pc.c0 * (lds + O_P)[pc.i0 * LDP + Z_PI + a] + pc.c1 * (lds + O_P)[pc.i1 * LDP + Z_PI + a] +
                                     pc.c2 * (lds + O_P)[pc.i2 * LDP + Z_PI + a];
            }
            if (lane < NX) {
                double r = defect_row(lds, dc, lane);
                (lds + O_rdef)[lane] = r;
                ac.prim = fmax(ac.prim, fabs(r)); ac.theta += fabs(r);
            }
            BMPC_SYNC();
            if (lane < NX) {
                double v = (lds + O_pv0)[lane];
                for (int j = 0; j < NX; j++) v += (lds + O_P)[lane * LDP + j] * (lds + O_rdef)[j];
                (lds + O_vt0)[lane] = v;
            } else if (lane < 2 * NX) {
                (lds + O_vt1)[lane - NX] = (lds + O_pv1)[lane - NX];
            }
            BMPC_SYNC();
            // ---- W += [As Bs]^T P+ [As Bs]: the joint columns form 4 groups (q~, dq~, ddq~, u) whose
            // [As Bs] columns are alpha[g][0..2] times the (q~, dq~, ddq~) rows, so each of the 49 joint
            // pairs (a,b) loads its 3x3 block of P once and updates all 16 group blocks of W ----
            {
                const double al[4][3] = {{1.0, 0.0, 0.0}, {dc.dt, 1.0, 0.0}, {0.5 * dc.dt * dc.dt, dc.dt, 1.0}, {dc.b3, dc.b2, dc.b1}};
                const int gpos[4] = {Z_Q, Z_DQ, Z_DDQ, Z_U};
                for (int e = lane; e < 49; e += BMPC_NT) {
                    int a = e / 7, bq = e - 7 * a;
                    double Pb[3][3];
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int s = 0; s < 3; s++) Pb[r][s] = (lds + O_P)[(7 * r + a) * LDP + 7 * s + bq];
#pragma unroll
                    for (int gi = 0; gi < 4; gi++) {
                        double t0 = al[gi][0] * Pb[0][0] + al[gi][1] * Pb[1][0] + al[gi][2] * Pb[2][0];
                        double t1 = al[gi][0] * Pb[0][1] + al[gi][1] * Pb[1][1] + al[gi][2] * Pb[2][1];
                        double t2 = al[gi][0] * Pb[0][2] + al[gi][1] * Pb[1][2] + al[gi][2] * Pb[2][2];
#pragma unroll
                        for (int gj = 0; gj < 4; gj++)
                            (lds + O_W)[(gpos[gi] + a) * LDW + gpos[gj] + bq] += t0 * al[gj][0] + t1 * al[gj][1] + t2 * al[gj][2];
                    }
                }
                // joint columns x single columns c in [Z_PI, Z_U) (identity in As), plus drs/dps = dt * (rs~/ps~)
                for (int e = lane; e < 7 * 11; e += BMPC_NT) {
                    int a = e / 11, c = Z_PI + (e - 11 * a);
                    double p0 = (lds + O_P)[a * LDP + c], p1 = (lds + O_P)[(7 + a) * LDP + c], p2 = (lds + O_P)[(14 + a) * LDP + c];
#pragma unroll
                    for (int gi = 0; gi < 4; gi++) {
                        double v = al[gi][0] * p0 + al[gi][1] * p1 + al[gi][2] * p2;
                        int r = gpos[gi] + a;
                        (lds + O_W)[r * LDW + c] += v; (lds + O_W)[c * LDW + r] += v;
                        if (c == Z_RS || c == Z_PS) {
                            int cw = (c == Z_RS) ? Z_DRS : Z_DPS;
                            (lds + O_W)[r * LDW + cw] += dc.dt * v; (lds + O_W)[cw * LDW + r] += dc.dt * v;
                        }
                    }
                }
                // single x single
                for (int e = lane; e < 11 * 11; e += BMPC_NT) {
                    int c1 = Z_PI + e / 11, c2 = Z_PI + e % 11;
                    double v = (lds + O_P)[c1 * LDP + c2];
                    (lds + O_W)[c1 * LDW + c2] += v;
                    bool s1 = (c1 == Z_RS || c1 == Z_PS), s2 = (c2 == Z_RS || c2 == Z_PS);
                    int w1 = (c1 == Z_RS) ? Z_DRS : Z_DPS, w2 = (c2 == Z_RS) ? Z_DRS : Z_DPS;
                    if (s2) (lds + O_W)[c1 * LDW + w2] += dc.dt * v;
                    if (s1) (lds + O_W)[w1 * LDW + c2] += dc.dt * v;
                    if (s1 && s2) (lds + O_W)[w1 * LDW + w2] += dc.dt * dc.dt * v;
                }
            }
            BMPC_SYNC();
            // ---- dense rank-3 part: E~ is nonzero only on the q~, dq~ and u columns (21) ----
            for (int e = lane; e < NZ * 21; e += BMPC_NT) {
                int i = e / 21, jj = e - 21 * i;
                int j = jj < 14 ? jj : Z_U + jj - 14;
                bool i_in = (i < Z_DDQ) || (i >= Z_U && i < Z_DRS);
                double ej0 = (lds + O_Et)[j], ej1 = (lds + O_Et)[NZ + j], ej2 = (lds + O_Et)[2 * NZ + j];
                double v = (lds + O_Y)[i * 3] * ej0 + (lds + O_Y)[i * 3 + 1] * ej1 + (lds + O_Y)[i * 3 + 2] * ej2;
                if (i_in) {
                    double ei0 = (lds + O_Et)[i], ei1 = (lds + O_Et)[NZ + i], ei2 = (lds + O_Et)[2 * NZ + i];
                    const LDSD* Pp = (lds + O_P) + Z_PI * LDP + Z_PI;
                    v += ei0 * (lds + O_Y)[j * 3] + ei1 * (lds + O_Y)[j * 3 + 1] + ei2 * (lds + O_Y)[j * 3 + 2];
                    v += ei0 * (Pp[0] * ej0 + Pp[1] * ej1 + Pp[2] * ej2) + ei1 * (Pp[LDP] * ej0 + Pp[LDP + 1] * ej1 + Pp[LDP + 2] * ej2) +
                         ei2 * (Pp[2 * LDP] * ej0 + Pp[2 * LDP + 1] * ej1 + Pp[2 * LDP + 2] * ej2);
                    (lds + O_W)[i * LDW + j] += v;
                } else {
                    (lds + O_W)[i * LDW + j] += v;
                    (lds + O_W)[j * LDW + i] += v;
                }
            }
            if (lane < NZ) {
                int c = lane;
                PhiCol pc = phi_col(c, dc);
                double gl = pc.c0 * (lds + O_lam)[pc.i0] + pc.c1 * (lds + O_lam)[pc.i1] + pc.c2 * (lds + O_lam)[pc.i2];
                double a0 = pc.c0 * (lds + O_vt0)[pc.i0] + pc.c1 * (lds + O_vt0)[pc.i1] + pc.c2 * (lds + O_vt0)[pc.i2];
                double a1 = pc.c0 * (lds + O_vt1)[pc.i0] + pc.c1 * (lds + O_vt1)[pc.i1] + pc.c2 * (lds + O_vt1)[pc.i2];
                for (int a = 0; a < 3; a++) {
                    double ea = (lds + O_Et)[a * NZ + c];
                    gl += ea * (lds + O_lam)[Z_PI + a]; a0 += ea * (lds + O_vt0)[Z_PI + a]; a1 += ea * (lds + O_vt1)[Z_PI + a];
                }
                (lds + O_gz)[c] += gl; (lds + O_g0)[c] += a0; (lds + O_g1)[c] += a1;
            }
        }
        BMPC_SYNC();
        BMPC_PROF(8);
        // ---- adjoint multipliers + dual residual (gz now holds the Lagrangian gradient) ----
        if (lane < NZ) {
            double gl = (lds + O_gz)[lane];
            if (lane >= NX || (k == 1 && lane >= 24)) ac.dual = fmax(ac.dual, fabs(gl));
            if (lane < NX) { (lds + O_lam)[lane] = gl; ac.lamsum += fabs(gl); }
        }
        BMPC_PROF(9);
        // ---- control block factorisation, gains, Schur complement ----
        double Lc[45];
        if (!chol9((lds + O_W), reg, Lc)) ok = false;
        if (lane < NX + 2) {
            double rhs[NU];
#pragma unroll
            for (int l = 0; l < NU; l++)
                rhs[l] = (lane < NX) ? (lds + O_W)[(NX + l) * LDW + lane] : (lane == NX ? (lds + O_g0)[NX + l] : (lds + O_g1)[NX + l]);
            chol9_solve(Lc, rhs);
#pragma unroll
            for (int l = 0; l < NU; l++) {
                if (lane < NX) { (lds + O_Kl)[l * NX + lane] = -rhs[l]; ws.K[(size_t)k * NU * NX + l * NX + lane] = -rhs[l]; }
                else { (lds + O_kf)[(lane - NX) * 16 + l] = -rhs[l]; ws.kf[k * 32 + (lane - NX) * 16 + l] = -rhs[l]; }
            }
        }
        BMPC_SYNC();
        for (int e = lane; e < NX * NX; e += BMPC_NT) {
            int i = e / NX, j = e % NX;
            if (j < i) continue;
            double v = (lds + O_W)[i * LDW + j];
            for (int l = 0; l < NU; l++) v += (lds + O_W)[(NX + l) * LDW + i] * (lds + O_Kl)[l * NX + j];
            (lds + O_P)[i * LDP + j] = v;
            (lds + O_P)[j * LDP + i] = v;
        }
        if (lane < 2 * NX) {
            int i = lane & (NX - 1);
            const LDSD* g = (lane < NX) ? (lds + O_g0) : (lds + O_g1);
            const LDSD* kf = (lds + O_kf) + ((lane < NX) ? 0 : 16);
            double v = g[i];
            for (int l = 0; l < NU; l++) v += (lds + O_W)[(NX + l) * LDW + i] * kf[l];
            ((lane < NX) ? (lds + O_pv0) : (lds + O_pv1))[i] = v;
        }
        if (lane < NZ) (lds + O_znext)[lane] = (lds + O_zeta)[lane];
        BMPC_SYNC();
        BMPC_PROF(10);
    }
    // initial defect of the pinned part of x_1 (zeta_1 is still in (lds + O_zeta))
    if (lane < 24) {
        double r = (lds + O_x1fix)[lane] - (lds + O_zeta)[lane];
        (lds + O_r0)[lane] = r;
        ac.prim = fmax(ac.prim, fabs(r)); ac.theta += fabs(r);
    }
    double cmax = wg_max(ac.cmax, (lds + O_red), lane), csum = wg_sum(ac.csum, (lds + O_red), lane), cmin = wg_min(ac.cmin, (lds + O_red), lane);
    double zsum = wg_sum(ac.zsum, (lds + O_red), lane), prim = wg_max(ac.prim, (lds + O_red), lane), theta = wg_sum(ac.theta, (lds + O_red), lane);
    double logs = wg_sum(ac.logs, (lds + O_red), lane), lamsum = wg_sum(ac.lamsum, (lds + O_red), lane), dual = wg_max(ac.dual, (lds + O_red), lane);
    int nrows = (int)(wg_sum((double)ac.nrows, (lds + O_red), lane) + 0.5);
    int neq = NX * (N - 2) + 24;
    kk.sd = fmax(100.0, (lamsum + zsum) / (double)(neq + nrows)) / 100.0;
    kk.sc = fmax(100.0, zsum / (double)nrows) / 100.0;
    kk.dual = dual; kk.prim = prim; kk.compl_ = cmax;
    kk.err = fmax(fmax(dual / kk.sd, prim), cmax / kk.sc);
    kk.f = fsum; kk.theta = theta; kk.avgc = csum / nrows; kk.minc = cmin; kk.logs = logs; kk.nrows = nrows;
    return ok;
}

// ------------------------------------------------------------------------------------------
// Forward sweep: Newton step, row steps, fraction-to-boundary, merit derivative
// ------------------------------------------------------------------------------------------
struct StepInfo { double ap, ad, dphi_f, dphi_bar; bool ok; };

BMPC_NOINL void forward_sweep(const Inst I, LDSD* lds, const WsPtr ws, const DynC dc, int lane,
                            double mu, const double* iw0, StepInfo& si) {
    const int N = I.N;
    const LDSD* sp = (lds + O_sp);
    const LDSD* wts = sp + P_W;
    double tau = fmax(0.99, 1.0 - mu);
    double ap_l = 1.0, ad_l = 1.0, dbar_l = 0.0, dphi_f = 0.0;
    si.ok = true;
    // stage-1 step: pinned part = initial defect, free part (rs~, ps~, d) minimises the cost-to-go
    {
        double Pf[36], rhs[8];
        for (int i = 0; i < 8; i++) {
            double s = (lds + O_pv0)[24 + i] + mu * (lds + O_pv1)[24 + i];
            for (int j = 0; j < 24; j++) s += (lds + O_P)[(24 + i) * LDP + j] * (lds + O_r0)[j];
            rhs[i] = -s;
        }
#define PF(i, j) Pf[(i) * ((i) + 1) / 2 + (j)]
        for (int j = 0; j < 8; j++) {
            double d = (lds + O_P)[(24 + j) * LDP + 24 + j];
            for (int l = 0; l < j; l++) d -= PF(j, l) * PF(j, l);
            if (!(d > 0)) { si.ok = false; d = 1.0; }
            d = sqrt(d);
            PF(j, j) = d;
            for (int i = j + 1; i < 8; i++) {
                double s = (lds + O_P)[(24 + i) * LDP + 24 + j];
                for (int l = 0; l < j; l++) s -= PF(i, l) * PF(j, l);
                PF(i, j) = s / d;
            }
        }
        for (int i = 0; i < 8; i++) { double s = rhs[i]; for (int l = 0; l < i; l++) s -= PF(i, l) * rhs[l]; rhs[i] = s / PF(i, i); }
        for (int i = 7; i >= 0; i--) { double s = rhs[i]; for (int l = i + 1; l < 8; l++) s -= PF(l, i) * rhs[l]; rhs[i] = s / PF(i, i); }
#undef PF
        BMPC_SYNC();
        if (lane < 24) (lds + O_dx)[lane] = (lds + O_r0)[lane];
        if (lane == 0) for (int i = 0; i < 8; i++) (lds + O_dx)[24 + i] = rhs[i];
        BMPC_SYNC();
    }
    for (int k = 1; k < N; k++) {
        if (lane < NZ) {
            double z = ws.zeta[k * ZPAD + lane];
            (lds + O_zeta)[lane] = z;
            ws.zsave[k * ZPAD + lane] = z;
            (lds + O_znext)[lane] = (k < N - 1) ? ws.zeta[(k + 1) * ZPAD + lane] : 0.0;
        }
        for (int e = lane; e < NU * NX; e += BMPC_NT) (lds + O_Kl)[e] = ws.K[(size_t)k * NU * NX + e];
        if (lane < 32) (lds + O_kf)[lane] = ws.kf[k * 32 + lane];
        for (int e = lane; e < EVAL_DOUBLES; e += BMPC_NT) (lds + O_yz)[e] = ws.ev[(size_t)k * WS_EVAL + e];   // evaluated by the backward sweep
        BMPC_SYNC();
        // g0 := dzeta, g1 := dy (natural)
        if (lane < NX) (lds + O_g0)[lane] = (lds + O_dx)[lane];
        else if (lane < NZ) {
            int l = lane - NX;
            double s = (lds + O_kf)[l] + mu * (lds + O_kf)[16 + l];
            for (int j = 0; j < NX; j++) s += (lds + O_Kl)[l * NX + j] * (lds + O_dx)[j];
            (lds + O_g0)[lane] = s;
        }
        BMPC_SYNC();
        if (lane < NZ) { ws.dz[k * ZPAD + lane] = (lds + O_g0)[lane]; (lds + O_g1)[lane] = nat_from_zeta((lds + O_g0), lane, dc); }
        BMPC_SYNC();
        if (lane < 24) {
            double s = 0;
            if (lane < 3) { for (int j = 0; j < 7; j++) s += (lds + O_J)[7 * lane + j] * (lds + O_g1)[Z_Q + j]; (lds + O_dloc)[lane] = s; }
            else if (lane < 6) {
                for (int j = 0; j < 7; j++) s += (lds + O_G)[7 * lane + j] * (lds + O_g1)[Z_Q + j] + (lds + O_J)[7 * lane + j] * (lds + O_g1)[Z_DQ + j];
                (lds + O_dloc)[lane] = (lds + O_g1)[Z_PI + lane - 3] + 0.5 * dc.dt * s;
            } else {
                int c = (lane - 6) / 3, a = (lane - 6) % 3;
                for (int j = 0; j < 7; j++) s += (lds + O_Jp)[21 * c + 7 * a + j] * (lds + O_g1)[Z_Q + j];
                (lds + O_dpt)[lane - 6] = s;
            }
        }
        BMPC_SYNC();
        // directional derivative of f (every lane, registers)
        {
            double s = 0;
            for (int a = 0; a < 6; a++) s += (lds + O_kin)[KN_G12 + a] * (lds + O_dloc)[a];
            for (int a = 0; a < 6; a++) {
                double dv = 0;
                for (int j = 0; j < 7; j++) dv += (lds + O_G)[7 * a + j] * (lds + O_g1)[Z_Q + j] + (lds + O_J)[7 * a + j] * (lds + O_g1)[Z_DQ + j];
                s += (lds + O_kin)[KN_G12 + 6 + a] * dv;
            }
            for (int j = 2; j <= 4; j++) s += 2 * wts[6] * (lds + O_yz)[Z_DQ + j] * (lds + O_g1)[Z_DQ + j];
            for (int j = 0; j < 7; j++) s += 2 * wts[7] * (lds + O_yz)[Z_U + j] * (lds + O_g1)[Z_U + j];
            s += 2 * wts[9] * (lds + O_yz)[Z_RS] * (lds + O_g1)[Z_RS] + 2 * wts[10] * (lds + O_yz)[Z_DRS] * (lds + O_g1)[Z_DRS] +
                 2 * wts[9] * (lds + O_yz)[Z_PS] * (lds + O_g1)[Z_PS] + 2 * wts[10] * (lds + O_yz)[Z_DPS] * (lds + O_g1)[Z_DPS];
            if (k == N - 1)
                for (int i = 0; i < 6; i++) {
                    double gg = 2 * wts[10] * (lds + O_yz)[Z_D + i] + (i != 4 ? 2 * wts[8] * (sp[P_SLACKS0 + i] + (lds + O_yz)[Z_D + i]) : 0.0);
                    s += gg * (lds + O_g1)[Z_D + i];
                }
            dphi_f += s;
        }
        for (int s = lane; s < NSLOT; s += BMPC_NT) {
            Row r;
            row_eval(I, lds, k, s, r);
            if (!r.kind) continue;
            double adot;
            if (r.kind == 1) adot = r.coef * (lds + O_g1)[r.pos];
            else if (r.kind == 2) adot = r.coef * (lds + O_g0)[r.pos];
            else if (r.kind == 3) {
                adot = 0;
                for (int c = 0; c < 6; c++) adot += r.a[c] * (lds + O_dloc)[c];
                if (r.sel == 1) adot -= (lds + O_g1)[Z_PS];
                else if (r.sel == 2) adot -= (lds + O_g1)[Z_RS];
                else if (r.sel == 3) adot -= (lds + O_g1)[Z_D + 5];
            } else {
                int c = r.pos;
                adot = r.a[0] * (lds + O_dpt)[3 * c] + r.a[1] * (lds + O_dpt)[3 * c + 1] + r.a[2] * (lds + O_dpt)[3 * c + 2] - (lds + O_g1)[Z_D + c];
            }
            double t = ws.t[k * NSLOT + s], z = ws.z[k * NSLOT + s];
            double dti = -(r.h + t) - adot;
            double dzi = (mu - t * z - z * dti) / t;
            ws.dt[k * NSLOT + s] = dti; ws.dzr[k * NSLOT + s] = dzi; ws.tsave[k * NSLOT + s] = t;
            if (dti < 0) ap_l = fmin(ap_l, -tau * t / dti);
            if (dzi < 0) ad_l = fmin(ad_l, -tau * z / dzi);
            dbar_l -= mu * dti / t;
        }
        // next dx = A dx + B dw + defect
        if (k < N - 1) {
            if (lane < NX) {
                const LDSD* d = (lds + O_g0);
                int i = lane;
                double v;
                if (i < Z_DQ) v = d[i] + dc.dt * d[i + 7] + 0.5 * dc.dt * dc.dt * d[i + 14] + dc.b3 * d[Z_U + i];
                else if (i < Z_DDQ) v = d[i] + dc.dt * d[i + 7] + dc.b2 * d[Z_U + i - 7];
                else if (i < Z_PI) v = d[i] + dc.b1 * d[Z_U + i - 14];
                else if (i < Z_RS) {
                    int a = i - Z_PI;
                    v = d[i];
                    for (int j = 0; j < 7; j++)
                        v += dc.dt * ((lds + O_G)[7 * (3 + a) + j] * (lds + O_g1)[Z_Q + j] + (lds + O_J)[7 * (3 + a) + j] * (lds + O_g1)[Z_DQ + j]);
                } else if (i == Z_RS) v = d[i] + dc.dt * d[Z_DRS];
                else if (i == Z_PS) v = d[i] + dc.dt * d[Z_DPS];
                else v = d[i];
                (lds + O_dxn)[i] = v + defect_row(lds, dc, i);
            }
            BMPC_SYNC();
            if (lane < NX) (lds + O_dx)[lane] = (lds + O_dxn)[lane];
        }
        BMPC_SYNC();
    }
    si.ap = fmin(1.0, wg_min(ap_l, (lds + O_red), lane));
    si.ad = fmin(1.0, wg_min(ad_l, (lds + O_red), lane));
    si.dphi_bar = wg_sum(dbar_l, (lds + O_red), lane);
    si.dphi_f = dphi_f;
}

// trial point zeta = zsave + alpha dz, t = tsave + alpha dt: barrier objective pieces
BMPC_NOINL void trial_sweep(const Inst I, LDSD* lds, const WsPtr ws, const DynC dc, int lane,
                          double alpha, const double* iw0, double& f1, double& th1, double& ls1) {
    const int N = I.N;
    double th_l = 0, ls_l = 0, fs = 0;
    for (int k = N - 1; k >= 1; k--) {
        if (lane < NZ) {
            double z = ws.zsave[k * ZPAD + lane] + alpha * ws.dz[k * ZPAD + lane];
            (lds + O_zeta)[lane] = z;
            ws.zeta[k * ZPAD + lane] = z;
        }
        BMPC_SYNC();
        fs += stage_eval(I, lds, dc, k, lane, 1, iw0);
        for (int e = lane; e < EVAL_DOUBLES; e += BMPC_NT) ws.ev[(size_t)k * WS_EVAL + e] = (lds + O_yz)[e];   // reused by the next sweeps
        for (int s = lane; s < NSLOT; s += BMPC_NT) {
            Row r;
            row_eval(I, lds, k, s, r);
            if (!r.kind) continue;
            double t = ws.tsave[k * NSLOT + s] + alpha * ws.dt[k * NSLOT + s];
            ws.t[k * NSLOT + s] = t;
            th_l += fabs(r.h + t);
            ls_l += log(t);
        }
        if (k < N - 1 && lane < NX) th_l += fabs(defect_row(lds, dc, lane));
        BMPC_SYNC();
        if (lane < NZ) (lds + O_znext)[lane] = (lds + O_zeta)[lane];
        BMPC_SYNC();
    }
    if (lane < 24) th_l += fabs((lds + O_x1fix)[lane] - (lds + O_zeta)[lane]);
    th1 = wg_sum(th_l, (lds + O_red), lane);
    ls1 = wg_sum(ls_l, (lds + O_red), lane);
    f1 = fs;
}

// ------------------------------------------------------------------------------------------
// One instance
// ------------------------------------------------------------------------------------------
BMPC_DEV void solve_instance(const KernelArgs& A, LDSD* lds, double* wsbase, int b, int lane) {
    const SolverOpts& o = A.o;
    Inst I;
    I.N = A.o.N; I.b = b;
    I.lbx = A.lbx + (size_t)b * (44 * A.o.N + 6); I.ubx = A.ubx + (size_t)b * (44 * A.o.N + 6);
    I.pg = A.p + (size_t)b * NPAR;
    I.prof = A.prof;
    const int N = o.N, n_w = 44 * N + 6;
    const double dt = o.dt;
    DynC dc;
    dc.dt = dt; dc.c1 = dt / 2; dc.c2 = dt * dt / 6; dc.c3 = dt * dt * dt / 24;
    dc.b1 = dt; dc.b2 = dt * dt; dc.b3 = 7 * dt * dt * dt / 12;
    WsPtr ws;
    ws_carve(wsbase, N, ws);
    const double* x0 = A.x0 + (size_t)b * n_w;
    const double* lbx = A.lbx + (size_t)b * n_w;
#ifdef BMPC_PROFILE
    if (lane < 16) (lds + O_misc)[32 + lane] = 0.0;
#endif
    for (int e = lane; e < 90; e += BMPC_NT) {   // robot constants: jxyz[21] jrot[63] ee_xyz[3] l4c_xyz[3]
        const double* rcp = (const double*)A.rc;
        (lds + O_rob)[e] = rcp[e < 87 ? e : e + 9];
    }
    for (int e = lane; e < NSP; e += BMPC_NT) (lds + O_sp)[e] = A.p[(size_t)b * NPAR + (e < SP_ASETJ ? e : e + (P_ASETJ - SP_ASETJ))];
    // stage-0 pins (BoundMPC.py:551-556): lbx == ubx there
    double iw0[3];
    for (int c = 0; c < 3; c++) iw0[c] = lbx[28 * N + (3 + c) * N];
    if (lane < 7) {
        int j = lane;
        double q0 = lbx[j * N], dq0 = lbx[7 * N + j * N], ddq0 = lbx[14 * N + j * N], u0 = lbx[21 * N + j * N];
        (lds + O_x1fix)[Z_Q + j] = q0 + dt * dq0 + dt * dt / 2 * ddq0 + dt * dt * dt / 8 * u0;
        (lds + O_x1fix)[Z_DQ + j] = dq0 + dt * ddq0 + dt * dt / 3 * u0;
        (lds + O_x1fix)[Z_DDQ + j] = ddq0 + dt / 2 * u0;
    } else if (lane < 10) {
        int c = lane - 7;
        (lds + O_x1fix)[Z_PI + c] = lbx[28 * N + (3 + c) * N] + dt / 2 * lbx[34 * N + (3 + c) * N];
    }
    BMPC_SYNC();
    // ---- initial iterate from x0 (natural -> zeta); pi_k = p_rot_k - dt/2 w(q_k, dq_k) ----
    for (int k = 1; k < N; k++) {
        if (lane < 7) {
            int j = lane;
            double uu = x0[21 * N + j * N + k];
            (lds + O_zeta)[Z_Q + j] = x0[j * N + k] - dc.c3 * uu;
            (lds + O_zeta)[Z_DQ + j] = x0[7 * N + j * N + k] - dc.c2 * uu;
            (lds + O_zeta)[Z_DDQ + j] = x0[14 * N + j * N + k] - dc.c1 * uu;
            (lds + O_zeta)[Z_U + j] = uu;
        } else if (lane < 10) {
            int c = lane - 7;
            (lds + O_zeta)[Z_PI + c] = x0[28 * N + (3 + c) * N + k];      // p_rot for now
        } else if (lane == 10) {
            double rs = x0[40 * N + 6 + k], drs = x0[41 * N + 6 + k], ps = x0[42 * N + 6 + k], dps = x0[43 * N + 6 + k];
            (lds + O_zeta)[Z_RS] = rs - dt / 2 * drs; (lds + O_zeta)[Z_PS] = ps - dt / 2 * dps;
            (lds + O_zeta)[Z_DRS] = drs; (lds + O_zeta)[Z_DPS] = dps;
        } else if (lane < 17) {
            int i = lane - 11;
            (lds + O_zeta)[Z_D + i] = x0[40 * N + i];
        }
        BMPC_SYNC();
        stage_eval(I, lds, dc, k, lane, 1, iw0);
        if (lane < NZ) {
            double z = (lds + O_zeta)[lane];
            if (lane >= Z_PI && lane < Z_RS) z -= dt / 2 * (lds + O_rc)[RC_V + 3 + lane - Z_PI];
            ws.zeta[k * ZPAD + lane] = z;
        }
        BMPC_SYNC();
    }
    // ---- row slacks / multipliers: t = max(-h, 1e-2), z = 1 ----
    {
            for (int k = N - 1; k >= 1; k--) {
            if (lane < NZ) (lds + O_zeta)[lane] = ws.zeta[k * ZPAD + lane];
            BMPC_SYNC();
            stage_eval(I, lds, dc, k, lane, 1, iw0);
            for (int e = lane; e < EVAL_DOUBLES; e += BMPC_NT) ws.ev[(size_t)k * WS_EVAL + e] = (lds + O_yz)[e];
            for (int s = lane; s < NSLOT; s += BMPC_NT) {
                Row r;
                row_eval(I, lds, k, s, r);
                ws.t[k * NSLOT + s] = r.kind ? fmax(-r.h, 1e-2) : 1.0;
                ws.z[k * NSLOT + s] = r.kind ? 1.0 : 0.0;
                ws.dzr[k * NSLOT + s] = 0.0;
            }
            BMPC_SYNC();
        }
    }
    int st = 1, it = 0;
    double mu = o.mu_init;
    const int MAXF = 8;
    double filt_th[MAXF], filt_phi[MAXF], filt_mu = -1, theta_max = 1e300, theta_min = 0;
    int nfilt = 0;
    double ad_pend = 0.0, reg = 1e-9, hreg = 0.0;
    Kkt kk;
    double err_prev = 1e300;
    for (it = 0;; it++) {
        int hess_mode = (o.hess == 2 && err_prev < o.hess_switch) ? 1 : 0;
        bool ok = backward_sweep(I, lds, ws, dc, lane, ad_pend, iw0, reg, hreg, hess_mode, kk);
        ad_pend = 0.0;
        if (kk.err <= o.tol && kk.dual <= 1.0 && kk.prim <= 1e-4 && kk.compl_ <= 1e-4) { st = 0; break; }
        if (it >= o.max_iter) { st = 1; break; }
        err_prev = kk.err;
        // monotone Fiacco-McCormick barrier update (oracle: mu_strategy 1)
        {
            double emu = fmax(fmax(kk.dual / kk.sd, kk.prim), fmax(fabs(kk.compl_ - mu), fabs(kk.minc - mu)) / kk.sc);
            while (emu <= o.kappa_eps * mu && mu > o.tol / 10.0) {
                mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
                emu = fmax(fmax(kk.dual / kk.sd, kk.prim), fmax(kk.compl_ - mu, 0.0) / kk.sc);
            }
        }
        int tries = 0;
        StepInfo si;
        for (;;) {
            if (ok) { BMPC_PROF_START(); forward_sweep(I, lds, ws, dc, lane, mu, iw0, si); ok = si.ok; BMPC_PROF(12); }
            if (ok) break;
            if (hess_mode) { hess_mode = 0; ++tries; }        // second-order terms not convex here: Gauss-Newton
            else {
                hreg = (hreg == 0.0) ? 1e-4 : hreg * 8;       // inertia correction (IPOPT delta_w)
                if (++tries > 12) { st = 3; break; }
            }
            ok = backward_sweep(I, lds, ws, dc, lane, 0.0, iw0, reg, hreg, hess_mode, kk);
        }
        if (st == 3) break;
        if (tries == 0) hreg = (hreg < 1e-8) ? 0.0 : hreg / 3;
        double f0 = kk.f, th0 = kk.theta, ls0 = kk.logs;
        double D = si.dphi_f + si.dphi_bar;
        double phi0 = f0 - mu * ls0;
        if (it == 0) { theta_max = 1e4 * fmax(1.0, th0); theta_min = 1e-4 * fmax(1.0, th0); }
        if (mu != filt_mu) { nfilt = 0; filt_mu = mu; }
        double alpha = si.ap;
        bool armijo_case = false;
        for (int bt = 0; bt < 10; bt++) {
            double f1, th1, ls1;
            BMPC_PROF_START(); trial_sweep(I, lds, ws, dc, lane, alpha, iw0, f1, th1, ls1); BMPC_PROF(13);
            double phi1 = f1 - mu * ls1;
            bool acc = (th1 <= theta_max);
            for (int j = 0; acc && j < nfilt; j++)
                if (th1 >= filt_th[j] && phi1 >= filt_phi[j]) acc = false;
            if (acc) {
                bool sw = (th0 <= theta_min) && (D < 0) && (alpha * pow(-D, 2.3) > pow(th0, 1.1));
                if (sw) { acc = (phi1 <= phi0 + 1e-4 * alpha * D + 1e-12 * fabs(phi0)); armijo_case = acc; }
                else acc = (th1 <= (1 - 1e-5) * th0) || (phi1 <= phi0 - 1e-5 * th0);
            }
            if (acc) break;
            alpha *= 0.5;
        }
        if (!armijo_case) {
            if (nfilt == MAXF) { for (int j = 0; j + 1 < MAXF; j++) { filt_th[j] = filt_th[j + 1]; filt_phi[j] = filt_phi[j + 1]; } nfilt--; }
            filt_th[nfilt] = (1 - 1e-5) * th0;
            filt_phi[nfilt] = phi0 - 1e-5 * th0;
            nfilt++;
        }
        ad_pend = si.ad;
    }
    // ---- outputs in the reference layout ----
    double* x = A.x + (size_t)b * n_w;
    double viol_l = 0;
    {
        for (int e = lane; e < n_w; e += BMPC_NT) x[e] = 0.0;
        BMPC_SYNC();
        if (lane < 7) {
            int j = lane;
            for (int blk = 0; blk < 4; blk++) x[blk * 7 * N + j * N] = lbx[blk * 7 * N + j * N];
        } else if (lane < 13) {
            int c = lane - 7;
            x[28 * N + c * N] = lbx[28 * N + c * N];
            x[34 * N + c * N] = lbx[34 * N + c * N];
        }
            for (int k = N - 1; k >= 1; k--) {
            if (lane < NZ) (lds + O_zeta)[lane] = ws.zeta[k * ZPAD + lane];
            BMPC_SYNC();
            stage_eval(I, lds, dc, k, lane, 1, iw0);
            if (lane < 28) {
                int blk = lane / 7, j = lane - 7 * blk;
                int pos = (blk == 0 ? Z_Q : blk == 1 ? Z_DQ : blk == 2 ? Z_DDQ : Z_U) + j;
                x[blk * 7 * N + j * N + k] = (lds + O_yz)[pos];
            } else if (lane < 34) {
                int c = lane - 28;
                x[28 * N + c * N + k] = (lds + O_rc)[RC_POSE + c];
                x[34 * N + c * N + k] = (lds + O_rc)[RC_V + c];
            } else if (lane == 34) {
                x[40 * N + 6 + k] = (lds + O_yz)[Z_RS]; x[41 * N + 6 + k] = (lds + O_yz)[Z_DRS];
                x[42 * N + 6 + k] = (lds + O_yz)[Z_PS]; x[43 * N + 6 + k] = (lds + O_yz)[Z_DPS];
                if (k == 1) { x[40 * N + 6] = (lds + O_zeta)[Z_RS]; x[42 * N + 6] = (lds + O_zeta)[Z_PS]; }
            } else if (lane < 41 && k == N - 1) {
                int i = lane - 35;
                x[40 * N + i] = (lds + O_yz)[Z_D + i];
            }
            // constraint violation as BoundMPC.py:613-615 (g rows only, 1e-6 dead band)
            for (int s = lane; s < NSLOT; s += BMPC_NT) {
                if (s < S_EE) continue;
                Row r;
                row_eval(I, lds, k, s, r);
                if (r.kind && r.h > 1e-6) viol_l += r.h;
            }
            if (k < N - 1 && lane < NX) { double r = fabs(defect_row(lds, dc, lane)); if (r > 1e-6 && lane < Z_D) viol_l += r; }
            if (A.g) {
                // constraint vector in the reference order (casadi_ocp_formulation.py:144-164, 304-380)
                double* g = A.g + (size_t)b * (147 * (N - 1) + 21);
                if (k < N - 1 && lane < 35) {
                    double v = 0.0;
                    if (lane < 21) v = defect_row(lds, dc, lane);
                    else if (lane >= 24 && lane < 27) v = defect_row(lds, dc, Z_PI + lane - 24);
                    else if (lane == 33) v = defect_row(lds, dc, Z_RS);
                    else if (lane == 34) v = defect_row(lds, dc, Z_PS);
                    g[35 * k + lane] = v;
                }
                double* gi = g + 35 * (N - 1) + 112 * (k - 1);
                for (int s = lane; s < S_END; s += BMPC_NT) {
                    if (s < S_EE) continue;
                    if (s >= S_TSET && k != N - 1) continue;
                    Row r;
                    row_eval(I, lds, k, s, r);
                    double v;
                    bool lower = (s >= S_ROTL && s < S_COL) || (s >= S_TROTL);
                    if (r.kind) v = lower ? -r.h : r.h;
                    else if (s < S_ROTU) v = -sp_b(A.p + (size_t)b * NPAR, P_BSET, s - S_EE, 4, (int)(lds + O_rc)[RC_SEG]) - (lds + O_yz)[Z_PS];
                    else if (s < S_PHI) { int c = (s - S_COL) / 15, rr = (s - S_COL) - 15 * c; v = -(lds + O_sp)[SP_BSETJ + rr * 6 + c] - (lds + O_rc)[RC_SL + c]; }
                    else v = -sp_b(A.p + (size_t)b * NPAR, P_BSET, s - S_TSET, 4, (int)(lds + O_rc)[RC_SEG + 1]) - (lds + O_rc)[RC_SL + 5];
                    gi[s - S_EE] = v;
                }
            }
            BMPC_SYNC();
            if (lane < NZ) (lds + O_znext)[lane] = (lds + O_zeta)[lane];
            BMPC_SYNC();
        }
        if (lane < 24) { double r = fabs((lds + O_x1fix)[lane] - (lds + O_zeta)[lane]); if (r > 1e-6) viol_l += r; }
        if (A.g && lane < 35) {
            double* g = A.g + (size_t)b * (147 * (N - 1) + 21);
            double v = 0.0;
            if (lane < 21) v = (lds + O_x1fix)[lane] - (lds + O_zeta)[lane];
            else if (lane >= 24 && lane < 27) v = (lds + O_x1fix)[Z_PI + lane - 24] - (lds + O_zeta)[Z_PI + lane - 24];
            g[lane] = v;
        }
    }
    double viol = wg_sum(viol_l, (lds + O_red), lane);
#ifdef BMPC_PROFILE
    if (lane < 16 && A.prof) A.prof[(size_t)BMPC_BLOCK() * 16 + lane] += (lds + O_misc)[32 + lane];
#endif
    if (lane == 0) {
        A.f[b] = kk.f;
        A.iters[b] = it;
        A.status[b] = st;
        A.viol[b] = viol;
    }
    BMPC_SYNC();
}

}  // namespace bmpc
