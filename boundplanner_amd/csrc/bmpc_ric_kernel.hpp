// bmpc_ric_kernel.hpp -- the sequential part of one interior-point iteration: ONE WAVEFRONT PER
// INSTANCE runs the Riccati recursion over the horizon (the stage-banded KKT factorisation, what
// IPOPT hands to MUMPS) with the value-function Hessian P (32x32) and the stage matrix W (41x41)
// in LDS, then the convergence test / barrier update (monotone Fiacco-McCormick) and the forward
// recursion for the Newton direction dz.  Stage records come from k_eval (bmpc_pair_kernels.hpp),
// gains go to a per-pair scratch record that the forward recursion reads back.
// Also here: the per-instance control kernels of the filter line search.
#pragma once
#include <type_traits>
#include "bmpc_pipeline.hpp"

namespace bmpc {

// ---- LDS layout of k_ric (doubles) ----
constexpr int R_P = 0;
constexpr int R_W = R_P + NPSYM;
constexpr int R_g0 = R_W + NZ * LDW;
constexpr int R_g1 = R_g0 + ZPAD;
constexpr int R_gz = R_g1 + ZPAD;
constexpr int R_lam = R_gz + ZPAD;
constexpr int R_pv0 = R_lam + NX;
constexpr int R_pv1 = R_pv0 + NX;
constexpr int R_vt0 = R_pv1 + NX;
constexpr int R_vt1 = R_vt0 + NX;
constexpr int R_rdef = R_vt1 + NX;
constexpr int R_kf = R_rdef + NX;           // 2 x 16
constexpr int R_Et = R_kf + 32;             // 3 x 41
constexpr int R_Y = R_Et + 3 * NZ;          // 41 x 3
constexpr int R_ew = R_Y + 3 * NZ;          // G_ang[3][7], J_ang[3][7]
constexpr int R_sufz = R_ew + 42;           // [7][3]
constexpr int R_dz2 = R_sufz + 24;          // sigma[2], r0[2], r1[2], zz[2]
constexpr int R_r0 = R_dz2 + 8;
#ifdef BMPC_PROFILE
constexpr int R_MISC_DOUBLES = 48;          // junk targets of the scatter and of the record touches [0,32), phase timers [32,48)
#else
constexpr int R_MISC_DOUBLES = 32;
#endif
constexpr int R_misc = R_r0 + NX;
constexpr int R_acc = R_misc + R_MISC_DOUBLES;   // 2 x 48: per-lane |lambda| sums and dual-residual maxima (lanes < 41)
constexpr int R_park = R_acc + 96;          // 16: uniform scalars parked across the sweep calls
#ifndef BMPC_RIC_LDS_PAD
#define BMPC_RIC_LDS_PAD 0      // (experiments: padding that lowers the number of resident workgroups)
#endif
constexpr int RIC_LDS_DOUBLES = R_park + 16 + BMPC_RIC_LDS_PAD;
// the gains K (9 x 32) of a stage live from its factorisation to its Schur complement (both in the factor phase): they share the
// coupling phase's R_Et, R_Y and the record's R_ew (read by the load and coupling phases, rewritten by the next stage's scatter)
constexpr int R_Kl = R_Et;
static_assert(R_Kl + NU * NX <= R_sufz, "gains fit the coupling scratch");
// used before the backward sweep / by the forward start only: they share the coupling phase's R_Et, R_Y (246 doubles)
constexpr int R_dx = R_Et;
constexpr int R_x1fix = R_dx + NX;
constexpr int R_dzeta = R_x1fix + NX;       // 48 (forward start: packed 8 x 8 factor)
constexpr int R_dy = R_dzeta + ZPAD;        // 48 (forward start: right-hand side)
static_assert(R_dy + ZPAD <= R_Y + 3 * NZ, "forward-start scratch fits the coupling scratch");
// 24.8 KB (31.6 KB with P as 32 x 33 and gains of their own, rounds 1-4: five workgroups per CU): SIX workgroups per CU (160 KB of
// LDS), i.e. 12 wavefronts = 3 per SIMD, which needs <= 168 VGPRs
#if !defined(BMPC_PROFILE) && BMPC_RIC_LDS_PAD == 0
static_assert(RIC_LDS_DOUBLES * 8 <= 26624, "k_ric LDS: 6 workgroups per CU");
#endif
// P[r][c] in the packed lower triangle
BMPC_INL int ptri(int i) { return BMPC_MUL24(i, i + 1) >> 1; }            // offset of row i
BMPC_INL int psym_hl(int hi, int lo) { return ptri(hi) + lo; }             // hi >= lo known
BMPC_INL int psym(int r, int c) { return r >= c ? ptri(r) + c : ptri(c) + r; }

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
    for (int i = 0; i < 17; i++) {
        const int f = F_P17 + 7 * i, pos = pos17(i);
        for (int sl = 0; sl < 3; sl++) set(f + sl, 2, Wo(pos, spos[sl]), Wo(spos[sl], pos));
        set(f + 3, 2, Wo(pos, pos), -1);
        set(f + 4, 1, R_g0 + pos, -1); set(f + 5, 1, R_g1 + pos, -1); set(f + 6, 1, R_gz + pos, -1);
    }
    for (int I = 14; I < 38; I++) {
        const int f = F_DGR + 4 * (I - 14), pos = dg_pos(I);
        set(f, 2, Wo(pos, pos), -1);
        set(f + 1, 1, R_g0 + pos, -1); set(f + 2, 1, R_g1 + pos, -1); set(f + 3, 1, R_gz + pos, -1);
    }
    for (int a = 0; a < 7; a++)
        for (int b = 0; b < 7; b++) {
            set(F_CQQ + a * 7 + b, 3, Wo(Z_Q + a, Z_Q + b), -1);
            set(F_CQD + a * 7 + b, 3, Wo(Z_Q + a, Z_DQ + b), Wo(Z_DQ + b, Z_Q + a));
        }
    for (int i = 0; i < 7; i++)
        for (int a = 0; a < 3; a++) set(F_CQP + 3 * i + a, 3, Wo(Z_Q + i, Z_PI + a), Wo(Z_PI + a, Z_Q + i));
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

// 9x9 Cholesky of the control block, reciprocal form: Lc packed lower, invd[j] = 1 / L[j][j]
BMPC_INL bool chol9i(const LDSD* W, double reg, double* Lc, double* invd) {
    bool ok = true;
#define LI(i, j) Lc[(i) * ((i) + 1) / 2 + (j)]
    BMPC_UNROLL
    for (int j = 0; j < NU; j++) {
        double d = W[(NX + j) * LDW + NX + j] + reg;
        BMPC_UNROLL
        for (int l = 0; l < j; l++) d -= LI(j, l) * LI(j, l);
        if (!(d > 0)) { ok = false; d = 1.0; }
        double r = BMPC_RSQRT(d);
        invd[j] = r;
        LI(j, j) = d * r;
        BMPC_UNROLL
        for (int i = j + 1; i < NU; i++) {
            double s = W[(NX + i) * LDW + NX + j];
            BMPC_UNROLL
            for (int l = 0; l < j; l++) s -= LI(i, l) * LI(j, l);
            LI(i, j) = s * r;
        }
    }
    return ok;
}
BMPC_INL void chol9i_solve(const double* Lc, const double* invd, double* b) {
    BMPC_UNROLL
    for (int i = 0; i < NU; i++) {
        double s = b[i];
        BMPC_UNROLL
        for (int l = 0; l < i; l++) s -= LI(i, l) * b[l];
        b[i] = s * invd[i];
    }
    BMPC_UNROLL
    for (int i = NU - 1; i >= 0; i--) {
        double s = b[i];
        BMPC_UNROLL
        for (int l = i + 1; l < NU; l++) s -= LI(l, i) * b[l];
        b[i] = s * invd[i];
    }
#undef LI
}

#define RL(x) (lds + (x))

// How the Riccati kernel reads its argument block.  On the GPU: straight from the kernel-argument segment (constant address
// space, scalar loads into SGPRs) -- `RicArgs` is an empty tag.  Handing `const PipeArgsH&` to the non-inlined sweeps forced a
// copy of the 480-byte block into every lane's scratch memory at kernel entry (30 x 16-byte stores per lane = 61 KB per
// workgroup and launch, more than the 49 KB of gains an instance writes: the kernel's 2.7 x WRITE_SIZE of round 3) and every
// use of a pointer inside the stage loop was a flat load from that copy on the critical path.  Emulation: a plain reference.
#ifdef BMPC_KERNARG_ARGS
typedef const __attribute__((address_space(4))) PipeArgsH* RicArgsPtr;
struct RicArgs { RicArgsPtr p; };
typedef const __attribute__((address_space(4))) PipeArgsH& RicArgsRef;
BMPC_INL RicArgs ric_kernel_args() { return RicArgs{(RicArgsPtr)__builtin_amdgcn_kernarg_segment_ptr()}; }
BMPC_INL RicArgsRef ric_args(RicArgs h) {      // the handle is uniform: back into scalar registers (function arguments arrive in VGPRs)
    const unsigned long long v = (unsigned long long)h.p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return *(RicArgsPtr)(((unsigned long long)hi << 32) | lo);
}
#else
typedef const PipeArgsH& RicArgs;
typedef const PipeArgsH& RicArgsRef;
BMPC_INL RicArgsRef ric_args(RicArgs a) { return a; }
#endif

// optional phase timing (diagnostic builds only: -DBMPC_PROFILE; never in the product build)
#ifdef BMPC_PROFILE
#define RPROF_START() long long rp_t0_ = clock64()
#define RPROF(i) do { long long t1_ = clock64(); if (lane == 0) RL(R_misc)[32 + (i)] += (double)(t1_ - rp_t0_); rp_t0_ = t1_; } while (0)
#else
#define RPROF_START() do {} while (0)
#define RPROF(i) do {} while (0)
#endif

// workgroup-wide reductions over NT lanes through LDS (fixed order -> reproducible)
template <int NT> BMPC_DEV double rsum(double v, LDSD* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = 0;
    for (int i = 0; i < NT; i++) s += red[i];
    return s;
}
template <int NT> BMPC_DEV double rmax(double v, LDSD* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = red[0];
    for (int i = 1; i < NT; i++) s = fmax(s, red[i]);
    return s;
}
template <int NT> BMPC_DEV double rmin(double v, LDSD* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = red[0];
    for (int i = 1; i < NT; i++) s = fmin(s, red[i]);
    return s;
}

// gains of (slot b, stage k): copy 0 in the slot's block, copies 1 .. RIC_NATT-1 (speculative attempts, k_ric_att) in A.kspec
template <class AT> BMPC_INL auto ric_krec(const AT& A, int b, int k, int copy) -> decltype(A.krec) {
    if (copy == 0) return A.krec + krec_of(A, b, k);
    return A.kspec + ((size_t)b * (RIC_NATT - 1) + (size_t)(copy - 1)) * ((size_t)(A.N - 1) * KREC) + (size_t)(k - 1) * KREC;
}

// The sequence of factorisation attempts of one iteration (oracle/bmpc_solve.c "inertia"; Waechter & Biegler 2006, Algorithm IC):
// attempt 0 = the iterate's Hessian (exact or Gauss-Newton, delta_w = 0); after a failed attempt either the Gauss-Newton fallback or
// the next delta_w.  The sequence is known BEFORE any attempt has run, which is what lets k_ric_att run several at once.
struct RicAttempt { int hess_mode, tries, gn_fell, dead; double dw; };
BMPC_INL void ric_attempt_next(RicAttempt& a, int gn_ok, double dwl, double dw0) {
    if (a.hess_mode && gn_ok) { a.hess_mode = 0; a.tries += 1; a.gn_fell = 1; }      // second-order terms not convex here: Gauss-Newton
    else {
        // inertia correction: delta_w I on the Hessian, natural coordinates
        double dw = a.dw;
        if (dw == 0.0) dw = (dwl == 0.0) ? dw0 : fmax(1e-20, dwl / 3.0);
        else dw *= (dwl == 0.0) ? 100.0 : 8.0;
        a.dw = dw;
        if (++a.tries > 14 || dw > 1e20) a.dead = 1;
    }
}

// backward recursion over the horizon; returns false if a control block is not positive definite
// The Riccati kernel reads the argument block through the plain-pointer view: with global-qualified
// pointers this compiler (ROCm 7.2) mis-assigns an odd register pair when it reloads a spilled
// 64-bit memory operand ("Subtarget requires even aligned vector registers")
// The stage loop of the backward recursion is cut into three non-inlined phases so that nothing stays live in registers
// across them (the loop body as one function needed the whole register file and spilled): record -> stage matrix in
// zeta coordinates; coupling with stage k+1; adjoint, control-block factorisation, gains, Schur complement.  Integer and
// pointer arguments only (see the note on floating-point arguments below).
template <int NT>
BMPC_INL void ric_phase_load_impl(RicArgs AH, LDSD* lds, int b, int lane, int k, int hess_mode, const int* tpk) {
    RicArgsRef A = ric_args(AH);
    const double hreg = lds[R_park + 12];
    const int N = A.N;
    const DynC dc = make_dync(A.o.dt);
    constexpr int NF = HREC / NT;
    constexpr int NE2 = (NZ * 9 + 27 + NT - 1) / NT;
    const int junk = R_misc + (lane & 31);
    const bool term = (k == N - 1);
    const size_t pi = pair_of(A, b, k);
    RPROF_START();
    // ---- stage matrix from the record (natural coordinates): the record's loads are in flight while W is cleared ----
    GCD hrec_k = (GCD)(A.hrec + hrec_of(A, b, k));
    double rv[NF];
    // (fields F_CQP .. HREC, the curvature block k_curv writes, are read in hess_mode only: the last 128 doubles of the record)
    BMPC_UNROLL
    for (int i = 0; i < NF; i++) rv[i] = (hess_mode || NT * (i + 1) <= F_CQP) ? hrec_k[lane + NT * i] : 0.0;
    {
        const bmpc_v2d z2 = {0.0, 0.0};
        for (int e = lane; e < NZ * LDW / 2; e += NT) *(LDSV2*)(RL(R_W) + 2 * e) = z2;
        if ((NZ * LDW) % 2 != 0 && lane == 0) RL(R_W)[NZ * LDW - 1] = 0.0;
    }
    BMPC_SYNC();
    BMPC_UNROLL
    for (int i = 0; i < NF; i++) {
        const int ps = tpk[i] >> 26, o1 = tpk[i] & 8191, o2 = (tpk[i] >> 13) & 8191;
        lds[ps == 1 ? o1 : junk] = rv[i]; lds[ps == 1 ? o2 : junk] = rv[i];
    }
    BMPC_SYNC();
    // adds: every address receives at most one add per pass (order-independent result)
    BMPC_UNROLL
    for (int i = 0; i < NF; i++) {
        const int ps = tpk[i] >> 26, o1 = tpk[i] & 8191, o2 = (tpk[i] >> 13) & 8191;
        if (ps == 2) { BMPC_LDS_ADD(lds + o1, rv[i]); if (o2 != 8191) BMPC_LDS_ADD(lds + o2, rv[i]); }
    }
    if (hreg != 0.0) {
        BMPC_SYNC();
        if (lane < NZ) RL(R_W)[lane * LDW + lane] += hreg;
    }
    BMPC_SYNC();
    if (hess_mode) {
        BMPC_UNROLL
        for (int i = 0; i < NF; i++) {
            const int ps = tpk[i] >> 26, o1 = tpk[i] & 8191, o2 = (tpk[i] >> 13) & 8191;
            if (ps == 3) { BMPC_LDS_ADD(lds + o1, rv[i]); if (o2 != 8191) BMPC_LDS_ADD(lds + o2, rv[i]); }
        }
        BMPC_SYNC();
    }
    // every lane has consumed the staged record: fetch the next stage's behind the rest of this stage
    // the next stage's record (5 KB): one dword per 128-byte line into the junk area by LDS-DMA (no register, nobody waits
    // for it) so that its loads hit the L2 when that stage starts
    if (k > 1 && lane < (hess_mode ? HREC : F_CQP) / 16) BMPC_TOUCH_LINE((GCD)(A.hrec + hrec_of(A, b, k - 1)) + 16 * lane, RL(R_misc));
    RPROF(0);
    // ---- second-order term of the pi dynamics: multiplier lam_pi(k+1) times d2(dt w)/d(q,dq)2 ----
    if (hess_mode && !term) {
        const double l0 = dc.dt * RL(R_lam)[Z_PI], l1 = dc.dt * RL(R_lam)[Z_PI + 1], l2 = dc.dt * RL(R_lam)[Z_PI + 2];
        const double lamv[3] = {l0, l1, l2};
        auto zax = [&](int i, double* z) { z[0] = RL(R_ew)[21 + i]; z[1] = RL(R_ew)[28 + i]; z[2] = RL(R_ew)[35 + i]; };
        auto suf = [&](int m, double* s) {   // sufz[m], m = 1..7
            if (m >= 7) { s[0] = 0; s[1] = 0; s[2] = 0; }
            else { s[0] = RL(R_sufz)[3 * (m - 1)]; s[1] = RL(R_sufz)[3 * (m - 1) + 1]; s[2] = RL(R_sufz)[3 * (m - 1) + 2]; }
        };
        for (int e = lane; e < 98; e += NT) {
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
    RPROF(1);
    // ---- natural -> zeta coordinates: H = T^T Hy T.  Column pass (W[i][u_t] += c3 W[i][q_t] + c2 W[i][dq_t] + c1 W[i][ddq_t], the
    // two slack columns likewise), then row pass (+ the three gradient vectors).  Lane l of each wavefront owns row (column) i = l;
    // the nine transformed positions t are split between the wavefronts, so that every offset is an immediate: the kernel is bound by
    // instruction issue, not by lanes (round 4: entries dealt out evenly over all lanes cost ~25 index instructions each) ----
    {
        const int wl = lane & 63, wv = lane >> 6;
        BMPC_UNROLL
        for (int pass = 0; pass < 2; pass++) {
            const int str = pass ? LDW : 1;
            if (wl < NZ) {
                LDSD* bp = RL(R_W) + (pass ? wl : wl * LDW);
                auto run = [&](auto T0c, auto T1c) {
                    constexpr int T0 = decltype(T0c)::value, T1 = decltype(T1c)::value;
                    double d0[T1 - T0], s0[T1 - T0], s1[T1 - T0], s2[T1 - T0];
                    BMPC_UNROLL
                    for (int t = T0; t < T1; t++) {
                        const bool isj = t < 7;
                        const int sp = isj ? Z_Q + t : (t == 7 ? Z_RS : Z_PS), dp = isj ? Z_U + t : (t == 7 ? Z_DRS : Z_DPS);
                        d0[t - T0] = bp[dp * str]; s0[t - T0] = bp[sp * str];
                        if (isj) { s1[t - T0] = bp[(sp + 7) * str]; s2[t - T0] = bp[(sp + 14) * str]; }
                    }
                    BMPC_UNROLL
                    for (int t = T0; t < T1; t++) {
                        const bool isj = t < 7;
                        const int dp = isj ? Z_U + t : (t == 7 ? Z_DRS : Z_DPS);
                        if (isj) bp[dp * str] = d0[t - T0] + dc.c3 * s0[t - T0] + dc.c2 * s1[t - T0] + dc.c1 * s2[t - T0];
                        else bp[dp * str] = d0[t - T0] + (0.5 * dc.dt) * s0[t - T0];
                    }
                };
                if constexpr (NT == 64) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
                else if (wv == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
                else run(std::integral_constant<int, 5>{}, std::integral_constant<int, 9>{});
            } else if (pass == 1) {
                // the 27 entries of the three gradient vectors: one per lane among the lanes without a column
                constexpr int FREE = 64 - NZ;
                static_assert(FREE * (NT / 64) >= 27 || NT == 64, "vector entries: one per free lane");
                for (int ev = (wl - NZ) + FREE * wv; ev < 27; ev += FREE * (NT / 64)) {
                    const int i = ev / 9, t = ev - 9 * i;
                    const bool isj = t < 7;
                    const int sp = isj ? Z_Q + t : (t == 7 ? Z_RS : Z_PS), dp = isj ? Z_U + t : (t == 7 ? Z_DRS : Z_DPS);
                    LDSD* vb = RL(i == 0 ? R_g0 : i == 1 ? R_g1 : R_gz);
                    const double d0 = vb[dp], s0 = vb[sp], s1 = vb[isj ? sp + 7 : sp], s2 = vb[isj ? sp + 14 : sp];
                    vb[dp] = isj ? d0 + dc.c3 * s0 + dc.c2 * s1 + dc.c1 * s2 : d0 + (0.5 * dc.dt) * s0;
                }
            }
            BMPC_SYNC();
        }
    }
    if (k == 1 && lane < 2) {   // zeta-diagonal rows rs~_1, ps~_1 >= 0
        int pos = lane ? Z_PS : Z_RS;
        RL(R_W)[pos * LDW + pos] += RL(R_dz2)[lane];
        RL(R_g0)[pos] -= RL(R_dz2)[2 + lane]; RL(R_g1)[pos] -= RL(R_dz2)[4 + lane]; RL(R_gz)[pos] -= RL(R_dz2)[6 + lane];
    }
    RPROF(2);
}

template <int NT>
BMPC_INL void ric_phase_couple_impl(RicArgs AH, LDSD* lds, int lane) {
    RicArgsRef A = ric_args(AH);
    const DynC dc = make_dync(A.o.dt);
    const bool term = false;
    RPROF_START();
    // ---- coupling with stage k+1:  W += Phi^T P Phi,  g += Phi^T (pv + P rdef),  gz += Phi^T lam,
    // Phi = [A B] + the three pi rows E (dt * d w / d(q~, dq~, u)) ----
    if (!term) {
        // C1: E^T (R_Et), Y~ = Phi0^T P[:, pi] + 1/2 E^T P[pi, pi] (R_Y), vt0 = pv0 + P rdef
        if (lane < NZ) {
            const int c = lane;
            double et[3];
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) {
                double v = 0;
                if (c < Z_DQ) v = dc.dt * RL(R_ew)[7 * a + c];
                else if (c < Z_DDQ) v = dc.dt * RL(R_ew)[21 + 7 * a + c - 7];
                else if (c >= Z_U && c < Z_DRS) v = dc.dt * (dc.c3 * RL(R_ew)[7 * a + c - Z_U] + dc.c2 * RL(R_ew)[21 + 7 * a + c - Z_U]);
                et[a] = v;
            }
            PhiCol pc = phi_col(c, dc);
            double pr[3][3], pp[3][3];
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) {
                pr[0][a] = RL(R_P)[psym(pc.i0, Z_PI + a)]; pr[1][a] = RL(R_P)[psym(pc.i1, Z_PI + a)]; pr[2][a] = RL(R_P)[psym(pc.i2, Z_PI + a)];
                pp[0][a] = RL(R_P)[psym(Z_PI, Z_PI + a)]; pp[1][a] = RL(R_P)[psym(Z_PI + 1, Z_PI + a)]; pp[2][a] = RL(R_P)[psym(Z_PI + 2, Z_PI + a)];
            }
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) {
                RL(R_Et)[a * NZ + c] = et[a];
                RL(R_Y)[c * 3 + a] = pc.c0 * pr[0][a] + pc.c1 * pr[1][a] + pc.c2 * pr[2][a] +
                                     0.5 * (et[0] * pp[0][a] + et[1] * pp[1][a] + et[2] * pp[2][a]);
            }
        }
        if (lane >= NT - 32) {
            const int r = lane - (NT - 32);
            double v = RL(R_pv0)[r];
            const int tr = ptri(r);
            BMPC_UNROLL
            for (int j = 0; j < NX; j++) v += RL(R_P)[j <= r ? tr + j : j * (j + 1) / 2 + r] * RL(R_rdef)[j];
            RL(R_vt0)[r] = v;
        }
        BMPC_SYNC();
        RPROF(7);
        // C2: structured part Phi0^T P Phi0.  Every lane first gathers the operands of ALL its
        // entries (joint x joint block, two joint x single entries, two single x single entries),
        // then computes and scatters: one LDS latency instead of five
        {
            const double al[4][3] = {{1.0, 0.0, 0.0}, {dc.dt, 1.0, 0.0}, {0.5 * dc.dt * dc.dt, dc.dt, 1.0}, {dc.b3, dc.b2, dc.b1}};
            const int gpos[4] = {Z_Q, Z_DQ, Z_DDQ, Z_U};
            LDSD* W = RL(R_W);
            const LDSD* P = RL(R_P);
            // (a) joint x joint: lane < 49 -> pair (a, bq)
            const bool hasA = lane < 49;
            const int aA = hasA ? lane / 7 : 0, bA = hasA ? lane - 7 * aA : 0;
            double Pb[3][3], w[4][4];
            // (b) joint rows x single columns c in [Z_PI, Z_U): 77 (a, c) pairs, lanes 49..63 take 0..14,
            // lanes 0..61 take 15..76
            constexpr int UB = (77 + 49 + NT - 1) / NT, UC = (121 + NT - 1) / NT;
            int eB[UB], aB[UB], cB[UB], cwB[UB];
            bool hasB[UB], slB[UB];
            double pB[UB][3], wr[UB][4], wc[UB][4], xr[UB][4], xc[UB][4];
            // (c) single x single: 121 pairs
            int c1C[UC], c2C[UC], w1C[UC], w2C[UC];
            bool hasC[UC], s1C[UC], s2C[UC];
            double vC[UC], aC[UC][4];
            if (hasA) {
                BMPC_UNROLL
                for (int r = 0; r < 3; r++)
                    BMPC_UNROLL
                    for (int s2 = 0; s2 < 3; s2++) Pb[r][s2] = P[r > s2 ? psym_hl(7 * r + aA, 7 * s2 + bA) : r < s2 ? psym_hl(7 * s2 + bA, 7 * r + aA) : psym(7 * r + aA, 7 * r + bA)];
                BMPC_UNROLL
                for (int gi = 0; gi < 4; gi++)
                    BMPC_UNROLL
                    for (int gj = 0; gj < 4; gj++) w[gi][gj] = W[(gpos[gi] + aA) * LDW + gpos[gj] + bA];
            }
            BMPC_UNROLL
            for (int u = 0; u < UB; u++) {
                eB[u] = lane - 49 + NT * u;
                hasB[u] = eB[u] >= 0 && eB[u] < 77;
                const int e = hasB[u] ? eB[u] : 0;
                aB[u] = e / 11; cB[u] = Z_PI + (e - 11 * aB[u]);
                slB[u] = (cB[u] == Z_RS || cB[u] == Z_PS);
                cwB[u] = (cB[u] == Z_RS) ? Z_DRS : Z_DPS;
                if (hasB[u]) {
                    pB[u][0] = P[psym_hl(cB[u], aB[u])]; pB[u][1] = P[psym_hl(cB[u], 7 + aB[u])]; pB[u][2] = P[psym_hl(cB[u], 14 + aB[u])];      // (cB >= 21 > 14 + aB)
                    BMPC_UNROLL
                    for (int gi = 0; gi < 4; gi++) {
                        const int r = gpos[gi] + aB[u];
                        wr[u][gi] = W[r * LDW + cB[u]]; wc[u][gi] = W[cB[u] * LDW + r];
                        xr[u][gi] = slB[u] ? W[r * LDW + cwB[u]] : 0.0; xc[u][gi] = slB[u] ? W[cwB[u] * LDW + r] : 0.0;
                    }
                }
            }
            BMPC_UNROLL
            for (int u = 0; u < UC; u++) {
                const int ec = lane + NT * u;
                hasC[u] = ec < 121;
                const int q1 = hasC[u] ? ec / 11 : 0;
                c1C[u] = Z_PI + q1; c2C[u] = Z_PI + (hasC[u] ? ec - 11 * q1 : 0);
                s1C[u] = (c1C[u] == Z_RS || c1C[u] == Z_PS); s2C[u] = (c2C[u] == Z_RS || c2C[u] == Z_PS);
                w1C[u] = (c1C[u] == Z_RS) ? Z_DRS : Z_DPS; w2C[u] = (c2C[u] == Z_RS) ? Z_DRS : Z_DPS;
                if (hasC[u]) {
                    vC[u] = P[psym(c1C[u], c2C[u])];
                    aC[u][0] = W[c1C[u] * LDW + c2C[u]];
                    aC[u][1] = s2C[u] ? W[c1C[u] * LDW + w2C[u]] : 0.0;
                    aC[u][2] = s1C[u] ? W[w1C[u] * LDW + c2C[u]] : 0.0;
                    aC[u][3] = (s1C[u] && s2C[u]) ? W[w1C[u] * LDW + w2C[u]] : 0.0;
                }
            }
            // compute + scatter (all targets of one lane and of different lanes are distinct)
            if (hasA) {
                BMPC_UNROLL
                for (int gi = 0; gi < 4; gi++) {
                    double t0 = al[gi][0] * Pb[0][0] + al[gi][1] * Pb[1][0] + al[gi][2] * Pb[2][0];
                    double t1 = al[gi][0] * Pb[0][1] + al[gi][1] * Pb[1][1] + al[gi][2] * Pb[2][1];
                    double t2 = al[gi][0] * Pb[0][2] + al[gi][1] * Pb[1][2] + al[gi][2] * Pb[2][2];
                    BMPC_UNROLL
                    for (int gj = 0; gj < 4; gj++)
                        W[(gpos[gi] + aA) * LDW + gpos[gj] + bA] = w[gi][gj] + t0 * al[gj][0] + t1 * al[gj][1] + t2 * al[gj][2];
                }
            }
            BMPC_UNROLL
            for (int u = 0; u < UB; u++) {
                if (hasB[u]) {
                    BMPC_UNROLL
                    for (int gi = 0; gi < 4; gi++) {
                        const int r = gpos[gi] + aB[u];
                        double v = al[gi][0] * pB[u][0] + al[gi][1] * pB[u][1] + al[gi][2] * pB[u][2];
                        W[r * LDW + cB[u]] = wr[u][gi] + v; W[cB[u] * LDW + r] = wc[u][gi] + v;
                        if (slB[u]) { W[r * LDW + cwB[u]] = xr[u][gi] + dc.dt * v; W[cwB[u] * LDW + r] = xc[u][gi] + dc.dt * v; }
                    }
                }
            }
            BMPC_UNROLL
            for (int u = 0; u < UC; u++) {
                if (hasC[u]) {
                    W[c1C[u] * LDW + c2C[u]] = aC[u][0] + vC[u];
                    if (s2C[u]) W[c1C[u] * LDW + w2C[u]] = aC[u][1] + dc.dt * vC[u];
                    if (s1C[u]) W[w1C[u] * LDW + c2C[u]] = aC[u][2] + dc.dt * vC[u];
                    if (s1C[u] && s2C[u]) W[w1C[u] * LDW + w2C[u]] = aC[u][3] + dc.dt * dc.dt * vC[u];
                }
            }
        }
        BMPC_SYNC();
        RPROF(8);
        // C3: rank-3 part  D[i][j] = Y~[i] . E[:, j] + Y~[j] . E[:, i]  on the 21 columns j where E is
        // nonzero: lane = column (three lane groups split the rows), two rows per batch
        {
            constexpr int G3 = NT / 21, NR = (NZ + G3 - 1) / G3, RB = 4, NB = (NR + RB - 1) / RB;   // row groups, rows per lane, rows per batch, batches
            const bool act = lane < 21 * G3;
            const int g3 = act ? lane / 21 : 0, jj = act ? lane - 21 * g3 : 0;
            const int j = jj < 14 ? jj : Z_U + jj - 14;
            LDSD* W = RL(R_W);
            const double ej0 = RL(R_Et)[j], ej1 = RL(R_Et)[NZ + j], ej2 = RL(R_Et)[2 * NZ + j];
            const double yj0 = RL(R_Y)[3 * j], yj1 = RL(R_Y)[3 * j + 1], yj2 = RL(R_Y)[3 * j + 2];
            BMPC_UNROLL
            for (int mb = 0; mb < NB; mb++) {
                double yi[RB][3], ei[RB][3], w0[RB], w1[RB];
                bool in_[RB], val[RB];
                int ii[RB];
                BMPC_UNROLL
                for (int u = 0; u < RB; u++) {
                    const int i = g3 + G3 * (RB * mb + u);
                    ii[u] = i; val[u] = act && (i < NZ);
                    const int ic = val[u] ? i : 0;
                    in_[u] = (ic < Z_DDQ) || (ic >= Z_U && ic < Z_DRS);
                    BMPC_UNROLL
                    for (int a = 0; a < 3; a++) { yi[u][a] = RL(R_Y)[3 * ic + a]; ei[u][a] = RL(R_Et)[a * NZ + ic]; }
                    w0[u] = W[ic * LDW + j]; w1[u] = W[j * LDW + ic];
                }
                BMPC_UNROLL
                for (int u = 0; u < RB; u++) {
                    if (val[u]) {
                        double v = yi[u][0] * ej0 + yi[u][1] * ej1 + yi[u][2] * ej2 + yj0 * ei[u][0] + yj1 * ei[u][1] + yj2 * ei[u][2];
                        W[ii[u] * LDW + j] = w0[u] + v;
                        if (!in_[u]) W[j * LDW + ii[u]] = w1[u] + v;
                    }
                }
            }
        }
        if (lane < NZ) {
            const int c = lane;
            PhiCol pc = phi_col(c, dc);
            double l3[3], v03[3], v13[3], lp[3], v0p[3], v1p[3], ea[3];
            l3[0] = RL(R_lam)[pc.i0]; l3[1] = RL(R_lam)[pc.i1]; l3[2] = RL(R_lam)[pc.i2];
            v03[0] = RL(R_vt0)[pc.i0]; v03[1] = RL(R_vt0)[pc.i1]; v03[2] = RL(R_vt0)[pc.i2];
            v13[0] = RL(R_pv1)[pc.i0]; v13[1] = RL(R_pv1)[pc.i1]; v13[2] = RL(R_pv1)[pc.i2];
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) { ea[a] = RL(R_Et)[a * NZ + c]; lp[a] = RL(R_lam)[Z_PI + a]; v0p[a] = RL(R_vt0)[Z_PI + a]; v1p[a] = RL(R_pv1)[Z_PI + a]; }
            double gl = pc.c0 * l3[0] + pc.c1 * l3[1] + pc.c2 * l3[2];
            double a0 = pc.c0 * v03[0] + pc.c1 * v03[1] + pc.c2 * v03[2];
            double a1 = pc.c0 * v13[0] + pc.c1 * v13[1] + pc.c2 * v13[2];
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) { gl += ea[a] * lp[a]; a0 += ea[a] * v0p[a]; a1 += ea[a] * v1p[a]; }
            RL(R_gz)[c] += gl; RL(R_g0)[c] += a0; RL(R_g1)[c] += a1;
        }
    }
}

template <int NT>
BMPC_INL bool ric_phase_factor_impl(RicArgs AH, LDSD* lds, int b, int lane, int k) {
    RicArgsRef A = ric_args(AH);
    const double reg = 1e-9;      // fixed regularisation of the control block
    const size_t pi = pair_of(A, b, k);
    bool ok = true;
    RPROF_START();
    BMPC_SYNC();
    RPROF(3);
    // ---- adjoint multipliers + dual residual (gz now holds the Lagrangian gradient) ----
    if (lane < NZ) {
        double gl = RL(R_gz)[lane];
        if (lane >= NX || (k == 1 && lane >= 24)) RL(R_acc)[48 + lane] = fmax(RL(R_acc)[48 + lane], fabs(gl));
        if (lane < NX) { RL(R_lam)[lane] = gl; RL(R_acc)[lane] += fabs(gl); }
    }
    // ---- control block factorisation, gains, Schur complement ----
    // the factor is needed by the 34 lanes that solve for a gain column: the first wavefront computes it (every lane for
    // itself: no LDS round trips inside the factorisation), the second skips the whole block; the verdict travels through LDS
    double* krec = ric_krec(A, b, k, (int)RL(R_park)[15]);      // (the gains of a speculative attempt go to its own copy)
    if (lane < 64) {
      double Lc[45], invd[NU];
      const bool pd = chol9i(RL(R_W), reg, Lc, invd);
      if (lane == 0) RL(R_park)[13] = pd ? 1.0 : 0.0;
      if (lane < NX + 2) {
        double rhs[NU];
        BMPC_UNROLL
        for (int l = 0; l < NU; l++)
            rhs[l] = (lane < NX) ? RL(R_W)[(NX + l) * LDW + lane] : (lane == NX ? RL(R_g0)[NX + l] : RL(R_g1)[NX + l]);
        chol9i_solve(Lc, invd, rhs);
        BMPC_UNROLL
        for (int l = 0; l < NU; l++) {
            if (lane < NX) { RL(R_Kl)[l * NX + lane] = -rhs[l]; krec[l * NX + lane] = -rhs[l]; }
            else { RL(R_kf)[(lane - NX) * 16 + l] = -rhs[l]; krec[NU * NX + (lane - NX) * 16 + l] = -rhs[l]; }
        }
      }
    }
    BMPC_SYNC();
    if (RL(R_park)[13] == 0.0) ok = false;
    RPROF(4);
    // P = W_xx + W_xu K: lane = column j (two half-waves split the rows), K[:, j] in registers,
    // W_ux rows fetched two at a time as broadcast 16-byte reads; loads of a batch precede its stores
    {
        constexpr int NH = NT / 32, RPH = NX / NH;      // row blocks, rows per block
        const int j = lane & 31, h = lane >> 5;
        double kj[NU];
        BMPC_UNROLL
        for (int l = 0; l < NU; l++) kj[l] = RL(R_Kl)[l * NX + j];
        const LDSD* W = RL(R_W);
        {
            BMPC_UNROLL
            for (int ib = 0; ib < RPH / 4; ib++) {
                const int i0 = RPH * h + 4 * ib;
                bmpc_v2d wa[NU], wb[NU];
                BMPC_UNROLL
                for (int l = 0; l < NU; l++) {
                    if constexpr (LDW % 2 == 0) {
                        wa[l] = *(const LDSV2*)(W + (NX + l) * LDW + i0);
                        wb[l] = *(const LDSV2*)(W + (NX + l) * LDW + i0 + 2);
                    } else {          // (odd row stride: rows are not 16-byte aligned)
                        wa[l][0] = W[(NX + l) * LDW + i0]; wa[l][1] = W[(NX + l) * LDW + i0 + 1];
                        wb[l][0] = W[(NX + l) * LDW + i0 + 2]; wb[l][1] = W[(NX + l) * LDW + i0 + 3];
                    }
                }
                double p0 = W[i0 * LDW + j], p1 = W[(i0 + 1) * LDW + j], p2 = W[(i0 + 2) * LDW + j], p3 = W[(i0 + 3) * LDW + j];
                BMPC_UNROLL
                for (int l = 0; l < NU; l++) { p0 += wa[l][0] * kj[l]; p1 += wa[l][1] * kj[l]; p2 += wb[l][0] * kj[l]; p3 += wb[l][1] * kj[l]; }
                // (P is symmetric up to rounding: the lower triangle is kept)
                const int pb = ptri(i0) + j;
                if (i0 >= j) RL(R_P)[pb] = p0;
                if (i0 + 1 >= j) RL(R_P)[pb + i0 + 1] = p1;
                if (i0 + 2 >= j) RL(R_P)[pb + 2 * i0 + 3] = p2;
                if (i0 + 3 >= j) RL(R_P)[pb + 3 * i0 + 6] = p3;
            }
        }
        // pv = g_x + W_xu kf (two right-hand sides)
        if (lane < 2 * NX) {
            const int i = lane & (NX - 1);
            const LDSD* g = (lane < NX) ? RL(R_g0) : RL(R_g1);
            const LDSD* kf = RL(R_kf) + ((lane < NX) ? 0 : 16);
            double wv[NU], kv[NU];
            BMPC_UNROLL
            for (int l = 0; l < NU; l++) { wv[l] = W[(NX + l) * LDW + i]; kv[l] = kf[l]; }
            double v = g[i];
            BMPC_UNROLL
            for (int l = 0; l < NU; l++) v += wv[l] * kv[l];
            ((lane < NX) ? RL(R_pv0) : RL(R_pv1))[i] = v;
        }
    }
    BMPC_SYNC();
    RPROF(5);
    return ok;
}

// One stage of the backward sweep reduced to the ADJOINT recursion: Lagrangian gradient gz_k in zeta coordinates, + Phi^T lam_{k+1},
// dual-residual / |lambda| accumulators, lam_k.  Runs for the stages below the first control block that was not positive definite in
// the FIRST sweep of an iteration: that sweep has to go on (the KKT error of the iterate needs the whole adjoint recursion), but
// nothing of its factorisation is used any more -- the retry sweep (Gauss-Newton fallback or delta_w) redoes it.  Same expressions
// on gz as ric_phase_load / _couple / _factor, so the KKT error does not depend on where the sweep failed.
template <int NT>
BMPC_INL void ric_stage_adjoint_impl(RicArgs AH, LDSD* lds, int b, int lane, int k, const int* tpk) {
    RicArgsRef A = ric_args(AH);
    const int N = A.N;
    const DynC dc = make_dync(A.o.dt);
    constexpr int NF = HREC / NT;
    const int junk = R_misc + (lane & 31);
    const bool term = (k == N - 1);
    GCD hrec_k = (GCD)(A.hrec + hrec_of(A, b, k));
    double rv[NF];
    BMPC_UNROLL
    for (int i = 0; i < NF; i++) rv[i] = (NT * (i + 1) <= F_CQP) ? hrec_k[lane + NT * i] : 0.0;
    BMPC_SYNC();                                       // the previous stage has read gz / ew / dz2
    BMPC_UNROLL
    for (int i = 0; i < NF; i++) {                     // the store pass of the scatter (gz, ew, dz2 among it; W entries are not used)
        const int ps = tpk[i] >> 26, o1 = tpk[i] & 8191, o2 = (tpk[i] >> 13) & 8191;
        lds[ps == 1 ? o1 : junk] = rv[i]; lds[ps == 1 ? o2 : junk] = rv[i];
    }
    if (k > 1 && lane < F_CQP / 16) BMPC_TOUCH_LINE((GCD)(A.hrec + hrec_of(A, b, k - 1)) + 16 * lane, RL(R_misc));
    BMPC_SYNC();
    // natural -> zeta coordinates of gz: the nine (u_j, drs, dps) entries of the row pass of ric_phase_load
    {
        const int t = lane < 9 ? lane : 0;
        const bool isj = t < 7;
        const int sp = isj ? Z_Q + t : (t == 7 ? Z_RS : Z_PS), dp = isj ? Z_U + t : (t == 7 ? Z_DRS : Z_DPS);
        const int o_d = lane < 9 ? R_gz + dp : junk, o_s = lane < 9 ? R_gz + sp : junk;
        const double d0 = lds[o_d], s0 = lds[o_s], s1 = lds[isj && lane < 9 ? o_s + 7 : o_s], s2 = lds[isj && lane < 9 ? o_s + 14 : o_s];
        lds[o_d] = d0 + (isj ? dc.c3 : 0.5 * dc.dt) * s0 + (isj ? dc.c2 : 0.0) * s1 + (isj ? dc.c1 : 0.0) * s2;
    }
    BMPC_SYNC();
    if (k == 1 && lane < 2) { const int pos = lane ? Z_PS : Z_RS; RL(R_gz)[pos] -= RL(R_dz2)[6 + lane]; }
    BMPC_SYNC();                                       // (the line above touches the RS / PS entries that lanes 24 / 25 update next)
    if (!term && lane < NZ) {                          // gz += Phi^T lam_{k+1} (the vector part of the coupling phase)
        const int c = lane;
        double ea[3];
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) {
            double v = 0;
            if (c < Z_DQ) v = dc.dt * RL(R_ew)[7 * a + c];
            else if (c < Z_DDQ) v = dc.dt * RL(R_ew)[21 + 7 * a + c - 7];
            else if (c >= Z_U && c < Z_DRS) v = dc.dt * (dc.c3 * RL(R_ew)[7 * a + c - Z_U] + dc.c2 * RL(R_ew)[21 + 7 * a + c - Z_U]);
            ea[a] = v;
        }
        PhiCol pc = phi_col(c, dc);
        double l3[3], lp[3];
        l3[0] = RL(R_lam)[pc.i0]; l3[1] = RL(R_lam)[pc.i1]; l3[2] = RL(R_lam)[pc.i2];
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) lp[a] = RL(R_lam)[Z_PI + a];
        double gl = pc.c0 * l3[0] + pc.c1 * l3[1] + pc.c2 * l3[2];
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) gl += ea[a] * lp[a];
        RL(R_gz)[c] += gl;
    }
    BMPC_SYNC();
    if (lane < NZ) {                                   // adjoint multipliers + dual residual, as ric_phase_factor
        double gl = RL(R_gz)[lane];
        if (lane >= NX || (k == 1 && lane >= 24)) RL(R_acc)[48 + lane] = fmax(RL(R_acc)[48 + lane], fabs(gl));
        if (lane < NX) { RL(R_lam)[lane] = gl; RL(R_acc)[lane] += fabs(gl); }
    }
}

template <int NT> BMPC_NOINL void ric_phase_load(RicArgs AH, LDSD* lds, int b, int lane, int k, int hess_mode, const int* tpk) {
    ric_phase_load_impl<NT>(AH, lds, b, lane, k, hess_mode, tpk);
}
template <int NT> BMPC_NOINL void ric_phase_couple(RicArgs AH, LDSD* lds, int lane) { ric_phase_couple_impl<NT>(AH, lds, lane); }
template <int NT> BMPC_NOINL bool ric_phase_factor(RicArgs AH, LDSD* lds, int b, int lane, int k) {
    return ric_phase_factor_impl<NT>(AH, lds, b, lane, k);
}

// SPLIT = true: the three phases are separate functions (230 VGPRs, two wavefronts per SIMD: the throughput variant, used while
// many instances are alive); SPLIT = false: one body (the whole register file, one wavefront per SIMD, but 19 % less latency
// per stage: nothing is recomputed at the phase boundaries) for the straggler tail, where latency is all that counts.
// Same arithmetic either way.
template <int NT, bool SPLIT>
BMPC_NOINL bool ric_backward(RicArgs AH, LDSD* lds, int b, int lane, int hess_mode, int may_abort) {
    RicArgsRef A = ric_args(AH);
    // no floating-point arguments: an odd-aligned 64-bit argument pair that gets spilled trips a
    // register-alignment bug of this compiler; scalars travel through LDS (R_park)
    const double reg = 1e-9;      // fixed regularisation of the control block
    const double hreg = lds[R_park + 12];
    const int N = A.N;
    const DynC dc = make_dync(A.o.dt);
    bool ok = true;
    // per-lane accumulators of |lambda| and of the dual residual live in LDS (long live ranges in
    // registers only get spilled): R_acc[lane], R_acc[48 + lane] (lanes < 41 accumulate)
    if (lane < 48) { RL(R_acc)[lane] = 0.0; RL(R_acc)[48 + lane] = 0.0; }
    if (lane < NX) { RL(R_lam)[lane] = 0; RL(R_pv0)[lane] = 0; RL(R_pv1)[lane] = 0; }
    constexpr int NF = HREC / NT;
    static_assert(HREC % NT == 0, "record loads are unconditional");
    // this lane's scatter-table entries (the same for every stage), packed: o1 | o2 << 13 | pass << 26
    // (offset 8191 = none); stores of fields that are not in the store pass go to a junk slot
    int tpk[NF];
    const int junk = R_misc + (lane & 31);
    BMPC_UNROLL
    for (int i = 0; i < NF; i++) {
        int f = lane + NT * i;
        int ps = (f < HREC) ? A.tbl[3 * f] : 0, o1 = (f < HREC) ? A.tbl[3 * f + 1] : -1, o2 = (f < HREC) ? A.tbl[3 * f + 2] : -1;
        if (ps == 0) { o1 = -1; o2 = -1; }
        if (ps == 1 && o2 < 0) o2 = junk;
        tpk[i] = (o1 < 0 ? 8191 : o1) | ((o2 < 0 ? 8191 : o2) << 13) | (ps << 26);
    }
    constexpr int NE2 = (NZ * 9 + 27 + NT - 1) / NT;
    static_assert((NT == 64 || NT == 128) && HREC % 128 == 0, "record DMA: 1 KiB per wavefront instruction");
    for (int k = N - 1; k >= 1; k--) {
        const bool term = (k == N - 1);
        const size_t pi = pair_of(A, b, k);
        if constexpr (SPLIT) {
#ifdef BMPC_RIC_CALLS
            ric_phase_load<NT>(AH, lds, b, lane, k, hess_mode, tpk);
            if (!term) ric_phase_couple<NT>(AH, lds, lane);
            if (!ric_phase_factor<NT>(AH, lds, b, lane, k)) ok = false;
#else
            // the phases inline, each on a lane index the compiler cannot relate to the others': the per-lane offsets are
            // recomputed per phase and stage (as with real calls) instead of being hoisted out of the stage loop, and no
            // call sequence / callee-saved registers are involved
            int l0 = lane, l1 = lane, l2 = lane;
            BMPC_OPAQUE_I(l0);
            ric_phase_load_impl<NT>(AH, lds, b, l0, k, hess_mode, tpk);
            BMPC_OPAQUE_I(l1);
            if (!term) ric_phase_couple_impl<NT>(AH, lds, l1);
            BMPC_OPAQUE_I(l2);
            if (!ric_phase_factor_impl<NT>(AH, lds, b, l2, k)) ok = false;
#endif
        } else {
            ric_phase_load_impl<NT>(AH, lds, b, lane, k, hess_mode, tpk);
            if (!term) ric_phase_couple_impl<NT>(AH, lds, lane);
            if (!ric_phase_factor_impl<NT>(AH, lds, b, lane, k)) ok = false;
        }
        // a retry pass (the KKT quantities of the iterate are known from the first one) stops at the first control block that
        // is not positive definite; the verdict is uniform (read from LDS behind a barrier)
        if (may_abort && !ok) break;
        // the first pass goes on for the KKT error of the iterate, but only with the adjoint recursion: the factorisation below a
        // failed block is not used (round 4: a failed first sweep cost 19 full stages; the stragglers fail at every iteration)
        if (!ok) {
            for (int k2 = k - 1; k2 >= 1; k2--) {
                int l3 = lane;
                BMPC_OPAQUE_I(l3);
                ric_stage_adjoint_impl<NT>(AH, lds, b, l3, k2, tpk);
            }
            break;
        }
    }
    // |lambda| sum (lanes < 32 contribute) and dual-residual maximum (lanes < 41), in lane order
    BMPC_SYNC();
    if (lane < 2) {
        double v = RL(R_acc)[48 * lane];
        const int n = lane ? NZ : NX;
#pragma unroll 4
        for (int i = 1; i < n; i++) { double x = RL(R_acc)[48 * lane + i]; v = lane ? fmax(v, x) : v + x; }
        RL(R_park)[9 + lane] = v;                         // results through LDS (see k_ric_body)
    }
    BMPC_SYNC();
    return ok;
}

// start of the forward recursion: step of x_1 for barrier parameter mu (pinned part = initial defect, free
// part minimises the cost-to-go); false if the free part of the stage-1 value function is not positive definite
// SPLIT: one instantiation per kernel variant, so that each is compiled under its kernel's register budget
template <int NT, bool SPLIT>
BMPC_NOINL bool ric_forward(RicArgs AH, LDSD* lds, int b, int lane) {
    RicArgsRef A = ric_args(AH);
    const double mu = lds[R_park + 11];
    bool ok = true;
    // 8 x 8 system of the free part of x_1, once per iteration: rolled loops on LDS operands (unrolled, this function alone
    // claimed the whole register file); every lane computes it redundantly, lane 0 publishes the result
    {
        double Pf[36], rhs[8];
#pragma unroll 1
        for (int i = 0; i < 8; i++) {
            double s = RL(R_pv0)[24 + i] + mu * RL(R_pv1)[24 + i];
#pragma unroll 4
            for (int j = 0; j < 24; j++) s += RL(R_P)[psym(24 + i, j)] * RL(R_r0)[j];
            rhs[i] = -s;
        }
#define PF(i, j) Pf[(i) * ((i) + 1) / 2 + (j)]
        BMPC_UNROLL
        for (int j = 0; j < 8; j++) {
            double d = RL(R_P)[psym(24 + j, 24 + j)];
            BMPC_UNROLL
            for (int l = 0; l < j; l++) d -= PF(j, l) * PF(j, l);
            if (!(d > 0)) { ok = false; d = 1.0; }
            d = sqrt(d);
            PF(j, j) = d;
            BMPC_UNROLL
            for (int i = j + 1; i < 8; i++) {
                double s = RL(R_P)[psym(24 + i, 24 + j)];
                BMPC_UNROLL
                for (int l = 0; l < j; l++) s -= PF(i, l) * PF(j, l);
                PF(i, j) = s / d;
            }
        }
        BMPC_UNROLL
        for (int i = 0; i < 8; i++) {
            double s = rhs[i];
            BMPC_UNROLL
            for (int l = 0; l < i; l++) s -= PF(i, l) * rhs[l];
            rhs[i] = s / PF(i, i);
        }
        BMPC_UNROLL
        for (int i = 7; i >= 0; i--) {
            double s = rhs[i];
            BMPC_UNROLL
            for (int l = i + 1; l < 8; l++) s -= PF(l, i) * rhs[l];
            rhs[i] = s / PF(i, i);
        }
#undef PF
        BMPC_SYNC();
        if (lane < 24) RL(R_dx)[lane] = RL(R_r0)[lane];
        if (lane == 0) {
            BMPC_UNROLL
            for (int i = 0; i < 8; i++) RL(R_dx)[24 + i] = rhs[i];
        }
        BMPC_SYNC();
    }
    if (!ok) return false;
    // the recursion over the stages runs in k_fwd (tiny LDS footprint: every instance resident at once)
    if (lane < NX) A.dx1[(size_t)b * NX + lane] = RL(R_dx)[lane];
    return true;
}

// ------------------------------------------------------------------------------------------
// k_fwd: forward Riccati recursion dz_k = (dx_k, kf_k + K_k dx_k), dx_{k+1} = A dx + B dw + E dy + defect.
// One wavefront per instance of the step list, 4.5 KB of LDS: all instances are resident together and
// hide each other's latency (inside k_ric this part ran at 4 instances per CU).
// ------------------------------------------------------------------------------------------
#ifndef BMPC_FW_DEPTH
#define BMPC_FW_DEPTH 1
#endif
constexpr int FW_DEPTH = BMPC_FW_DEPTH;
static_assert(FW_DEPTH == 1, "k_fwd keeps one stage in flight (depth 3 / 5 measured in round 4: no gain)");
constexpr int FW_kf = 0, FW_ew = FW_kf + 32, FW_rdef = FW_ew + 42, FW_dx = FW_rdef + NX + 6,
              FW_dzeta = FW_dx + NX, FW_part = FW_dzeta + ZPAD, FW_LDS_DOUBLES = FW_part + 40;
static_assert(NX == 32 && NU == 9, "k_fwd: nine gain rows in four parts of eight columns, one (row, part) per lane");

// The product K dx runs on 36 lanes -- lane 4 l + p holds columns 8 p .. 8 p + 7 of row l in registers, straight from the gain
// record (64 contiguous bytes per lane), and the four partial sums of a row are added in the order p = 0 .. 3.  (Until the end of
// round 4 nine lanes walked a row of 32 each out of LDS, all nine on the same bank: the kernel was bound by instruction issue
// at 9 .. 32 busy lanes of 64.)
BMPC_DEV void k_fwd_body(const PipeArgs& A, int blk, int lane, LDSD* lds) {
    const int count = A.L.cnt[1];
    if (blk >= count) return;
    const int b = A.L.step[blk], N = A.N;
    const DynC dc = make_dync(A.o.dt);
    const double mu = A.st[b].mu;
    const int ksel = A.st[b].ksel;                  // which copy of the gains this iteration's factorisation left (ric_finish)
    const int kl = (lane < 4 * NU) ? lane : 0;      // (lanes beyond the 36 load row 0 again and drop the product)
    const int krow = kl >> 2, kpart = kl & 3;
    // stage data in flight: one stage ahead
    double nK[8], n_kf = 0, n_ew = 0, n_rd = 0;
    auto fetch = [&](int k) {
        GCD krec = ric_krec(A, b, k, ksel);
        GCD rec = A.hrec + hrec_of(A, b, k);
        BMPC_UNROLL
        for (int i = 0; i < 8; i++) nK[i] = krec[krow * NX + 8 * kpart + i];
        if (lane < 32) n_kf = krec[NU * NX + lane];
        if (lane < 42) n_ew = rec[F_EW + lane];
        if (lane < NX) n_rd = rec[F_RDEF + lane];
    };
    fetch(1);
    if (lane < NX) lds[FW_dx + lane] = A.dx1[(size_t)b * NX + lane];
#pragma unroll 1
    for (int k = 1; k < N; k++) {
        const size_t pi = pair_of(A, b, k);
        double cK[8];
        BMPC_UNROLL
        for (int i = 0; i < 8; i++) cK[i] = nK[i];
        if (lane < 32) lds[FW_kf + lane] = n_kf;
        if (lane < 42) lds[FW_ew + lane] = n_ew;
        if (lane < NX) lds[FW_rdef + lane] = n_rd;
        if (k + 1 < N) fetch(k + 1);
        BMPC_SYNC();
        {   // partial sums of K dx
            double s = 0;
            BMPC_UNROLL
            for (int i = 0; i < 8; i++) s += cK[i] * lds[FW_dx + 8 * kpart + i];
            if (lane < 4 * NU) lds[FW_part + lane] = s;
        }
        BMPC_SYNC();
        // dzeta = (dx, kf0 + mu kf1 + K dx)
        double dzv = 0;
        if (lane < NX) dzv = lds[FW_dx + lane];
        else if (lane < NZ) {
            const int l = lane - NX;
            double sum = lds[FW_kf + l] + mu * lds[FW_kf + 16 + l];
            BMPC_UNROLL
            for (int pp = 0; pp < 4; pp++) sum += lds[FW_part + 4 * l + pp];
            dzv = sum;
        }
        if (lane < NZ) { lds[FW_dzeta + lane] = dzv; A.dz[(size_t)lane * A.NP + pi] = dzv; }
        BMPC_SYNC();
        if (k < N - 1) {
            double v = 0;
            if (lane < NX) {
                const LDSD* d = lds + FW_dzeta;
                const int i = lane;
                if (i < Z_DQ) v = d[i] + dc.dt * d[i + 7] + 0.5 * dc.dt * dc.dt * d[i + 14] + dc.b3 * d[Z_U + i];
                else if (i < Z_DDQ) v = d[i] + dc.dt * d[i + 7] + dc.b2 * d[Z_U + i - 7];
                else if (i < Z_PI) v = d[i] + dc.b1 * d[Z_U + i - 14];
                else if (i < Z_RS) {
                    const int a = i - Z_PI;
                    v = d[i];
                    BMPC_UNROLL
                    for (int j = 0; j < 7; j++) {      // natural steps dq = dq~ + c3 du, d(dq) = d(dq~) + c2 du
                        const double dyq = d[Z_Q + j] + dc.c3 * d[Z_U + j], dydq = d[Z_DQ + j] + dc.c2 * d[Z_U + j];
                        v += dc.dt * (lds[FW_ew + 7 * a + j] * dyq + lds[FW_ew + 21 + 7 * a + j] * dydq);
                    }
                } else if (i == Z_RS) v = d[i] + dc.dt * d[Z_DRS];
                else if (i == Z_PS) v = d[i] + dc.dt * d[Z_DPS];
                else v = d[i];
                v += lds[FW_rdef + i];
            }
            BMPC_SYNC();      // every lane has read dzeta / dx of this stage
            if (lane < NX) lds[FW_dx + lane] = v;
        }
        BMPC_SYNC();
    }
}

// after the first backward sweep of an iteration: scaled KKT error (IPOPT's termination test), stall bookkeeping, monotone
// Fiacco-McCormick barrier update with the error-tied floor.  Returns 0 converged / 1 iteration limit / -1 go on (then the new
// barrier parameter is in R_park[11]).  A function of its own so that nothing of it (the polynomial constants of pow(), 17
// register pairs) is alive in the kernel body across the sweep calls, where it was saved and restored around every call.
template <bool SPLIT>      // (one instantiation per kernel variant, compiled under its kernel's register budget)
BMPC_NOINL int ric_kkt_and_barrier(RicArgs AH, LDSD* lds, int b, int lane, int it, int hess_mode, int gn_ok) {
    RicArgsRef A = ric_args(AH);
    const auto& o = A.o;
    InstState* st = A.st + b;
    const int N = A.N;
    const double lamsum = RL(R_park)[9], dual = RL(R_park)[10];
    double mu = RL(R_park)[11];
    const LDSD* pk = RL(R_park);
    const double cmax = pk[0], cmin = pk[2], zsum = pk[3], prim = pk[4], nrows = pk[7];
    const int neq = NX * (N - 2) + 24;
    double sd = fmax(100.0, (lamsum + zsum) / ((double)neq + nrows)) / 100.0;
    double sc = fmax(100.0, zsum / nrows) / 100.0;
    double err = fmax(fmax(dual / sd, prim), cmax / sc);
#ifdef BMPC_EMU_TRACE
    if (lane == 0 && getenv("BMPC_EMU_TRACE")) printf("[b %d] it %3d err %.6e (d %.6e p %.6e c %.6e) mu %.2e hess_mode %d gn_ok %d stall %d\n", A.src[b], it, err, dual, prim, cmax, mu, hess_mode, gn_ok, st->stall);
#endif
    if (err <= o.tol && dual <= 1.0 && prim <= 1e-4 && cmax <= 1e-4) return 0;
    if (it >= o.max_iter) return 1;
    // (the error goes into the instance state in ric_note_err: k_ric_att must not touch what its sibling attempts read)
    if (lane == 0) RL(R_park)[5] = err;
    // monotone Fiacco-McCormick barrier update; a decrease stops at (scaled error) / mu_floor_k
    double emu = fmax(fmax(dual / sd, prim), fmax(fabs(cmax - mu), fabs(cmin - mu)) / sc);
    while (emu <= o.kappa_eps * mu && mu > o.tol / 10.0) {
        const double mu_before = mu;
        mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
        if (o.mu_floor_k > 0) {
            const double m2 = fmax(mu, fmin(emu / o.mu_floor_k, 0.1));
            if (m2 >= mu_before) { mu = mu_before; break; }
            mu = m2;
        }
        emu = fmax(fmax(dual / sd, prim), fmax(cmax - mu, 0.0) / sc);
    }
    BMPC_SYNC();                                     // every lane has read mu before lane 0 replaces it
    if (lane == 0) RL(R_park)[11] = mu;
    BMPC_SYNC();
    return -1;
}

BMPC_INL void ric_note_err(InstState* st, double err) {      // lane 0, after ric_kkt_and_barrier returned -1
    st->err_prev = err;
    if (err < 0.9 * st->err_best) { st->err_best = err; st->stall = 0; } else st->stall += 1;
}

// ---- pieces of the Riccati kernels ----
// pinned part of x_1 and its defect -> R_r0
template <class AT> BMPC_INL void ric_load_r0(const AT& A, LDSD* lds, int b, int lane) {
    const int N = A.N, n_w = 44 * N + 6;
    const double* lbx = A.lbx + (size_t)A.src[b] * n_w;
    if (lane == 0) {
        double x1fix[24];
        x1fix_eval((GCD)lbx, N, A.o.dt, x1fix);
        for (int i = 0; i < 24; i++) RL(R_x1fix)[i] = x1fix[i];
    }
    BMPC_SYNC();
    if (lane < 24) RL(R_r0)[lane] = RL(R_x1fix)[lane] - cur_zeta(A, A.st[b].flip)[(size_t)lane * A.NP + pair_of(A, b, 1)];
}
// KKT partial sums of the pairs: staged in LDS [quantity][pair], then one lane per quantity adds them
// in pair order (fixed order -> reproducible; N - 1 <= 63 pairs) -> R_park[0..4], [7]; f0 -> [8]
template <class AT> BMPC_INL void ric_load_kkt_sums(const AT& A, LDSD* lds, int b, int lane) {
    const int N = A.N;
    LDSD* stg = RL(R_W);          // free until the backward sweep
    if (lane < N - 1) {
        const double* P = A.part + pair_of(A, b, 1) + lane;
        stg[0 * 64 + lane] = P[PT_CMAX * A.NP]; stg[1 * 64 + lane] = P[PT_CSUM * A.NP]; stg[2 * 64 + lane] = P[PT_CMIN * A.NP];
        stg[3 * 64 + lane] = P[PT_ZSUM * A.NP]; stg[4 * 64 + lane] = P[PT_PRIM * A.NP]; stg[5 * 64 + lane] = P[PT_NROWS * A.NP];
    }
    BMPC_SYNC();
    if (lane < 6) {
        const int kind = (lane == 0 || lane == 4) ? 1 : (lane == 2 ? 2 : 0);      // 0 sum, 1 max, 2 min
        double v = stg[lane * 64];
#pragma unroll 4
        for (int k = 1; k < N - 1; k++) {
            double x = stg[lane * 64 + k];
            v = kind == 0 ? v + x : (kind == 1 ? fmax(v, x) : fmin(v, x));
        }
        RL(R_park)[lane == 5 ? 7 : lane] = v;        // cmax, csum, cmin, zsum, prim -> [0..4], nrows -> [7]
    }
    if (lane == 6) RL(R_park)[8] = A.st[b].f0;
    BMPC_SYNC();
}
// what an iteration's factorisation leaves in the instance state, and the list the instance goes to (lane 0)
template <class AT> BMPC_INL void ric_finish(const AT& A, LDSD* lds, int b, int status, const RicAttempt& at, int hess_mode_in, double mu, int ksel) {
    InstState* st = A.st + b;
    const auto& o = A.o;
    if (status >= 0) {
        st->state = ST_DONE; st->status = status; st->fk = RL(R_park)[8];
        BMPC_ATOMIC_INC(A.L.cnt + 5);
        int pos = BMPC_ATOMIC_INC(A.L.cnt + 8);      // to be retired (outputs written, slot refilled)
        A.L.done[pos] = b;
    } else {
        if (at.dw > 0.0) st->dw_last = at.dw;
        // after a Gauss-Newton fallback the exact Hessian is tried again after 1, 2, ... gn_backoff iterations (ls_instance, in k_trial)
        if (at.gn_fell && o.gn_backoff > 0) {
            const int gb = st->gn_back ? (2 * st->gn_back < o.gn_backoff ? 2 * st->gn_back : o.gn_backoff) : 1;
            st->gn_back = gb; st->gn_skip = gb;
        } else if (at.tries == 0 && hess_mode_in) st->gn_back = 0;
        st->hreg = at.dw; st->mu = mu; st->tries = at.tries; st->ksel = ksel; st->state = ST_STEP;
        int pos = BMPC_ATOMIC_INC(A.L.cnt + 1);
        A.L.step[pos] = b;
    }
}

// lds: RIC_LDS_DOUBLES.  One workgroup (one wavefront) per entry of the eval list.
// RESUME: the second kernel of the speculative pair (k_ric_att -> k_ric_sel): attempts 0 .. RIC_NATT-1 have been run side by side by
// k_ric_att, one workgroup each; this kernel walks the same sequence, takes the first attempt whose sweep succeeded (forward start
// from the state that attempt saved) and only runs sweeps of its own for attempts beyond the speculated ones.
template <int NT, bool SPLIT = true, bool RESUME = false>
BMPC_DEV void k_ric_body(RicArgs AH, int blk, int lane, LDSD* lds) {
    RicArgsRef A = ric_args(AH);
    const int count = A.L.cnt[0];
    if (blk >= count) return;
    const int b = BMPC_UNIFORM(A.L.eval[blk]);
    if (!RESUME && lane == 0) BMPC_ATOMIC_INC(A.L.cnt + (SPLIT ? 11 : 12));      // instance-iterations of this variant (bmpc_debug_ric_stats)
    const auto& o = A.o;
    InstState* st = A.st + b;
    ric_load_r0(A, lds, b, lane);
    if (!RESUME) ric_load_kkt_sums(A, lds, b, lane);
    else { if (lane == 6) RL(R_park)[8] = st->f0; }
#ifdef BMPC_PROFILE
    if (lane < 16) RL(R_misc)[32 + lane] = 0.0;
#endif
    // uniform scalars that must survive the (non-inlined) sweeps: integers in scalar registers, mu, delta_w and the last
    // successful delta_w in LDS (R_park[11], [12], [14]) -- every vector register kept across the calls adds to the kernel's count
    RicAttempt at;
    at.hess_mode = BMPC_UNIFORM(st->hess_mode); at.tries = 0; at.gn_fell = 0; at.dead = 0; at.dw = 0.0;
    const int hess_mode_in = at.hess_mode;
    const int it = BMPC_UNIFORM(st->it);
    // not positive definite with the exact Hessian: Gauss-Newton fallback while far from a solution and still improving,
    // inertia correction otherwise (oracle/bmpc_solve.c, "inertia"); err_prev is still the previous iterate's error here
    const int gn_ok = BMPC_UNIFORM((o.inertia == 0 || (o.inertia == 2 && st->err_prev > o.inertia_err && st->stall < o.stall_n)) ? 1 : 0);
    if (lane == 0) { RL(R_park)[12] = 0.0; RL(R_park)[11] = st->mu; RL(R_park)[14] = st->dw_last; RL(R_park)[15] = 0.0; }
    bool first = true;
    int status = -1, ksel = 0;
    for (int att = 0;; att++) {
        BMPC_SYNC();
        bool ok;
        if (RESUME && att < A.natt) {
            // the attempt has been run by k_ric_att: its verdict, (attempt 0) the KKT test and the new barrier parameter, and the
            // state the forward start needs
            const double* fs = A.fspec + ((size_t)b * RIC_NATT + att) * RIC_FS;
            ok = fs[0] != 0.0;
            if (att == 0) {
                status = (int)fs[1];
                if (lane == 0) { RL(R_park)[11] = fs[2]; RL(R_park)[5] = fs[3]; }
            }
            if (ok && status < 0) {
                for (int e = lane; e < RIC_FS_P; e += NT) RL(R_P)[RIC_FS_P0 + e] = fs[8 + e];
                if (lane < 8) { RL(R_pv0)[24 + lane] = fs[8 + RIC_FS_P + lane]; RL(R_pv1)[24 + lane] = fs[8 + RIC_FS_P + 8 + lane]; }
            }
            BMPC_SYNC();
        } else ok = ric_backward<NT, SPLIT>(AH, lds, b, lane, at.hess_mode, first ? 0 : 1);
        if (first) {
            first = false;
            if (!RESUME) status = ric_kkt_and_barrier<SPLIT>(AH, lds, b, lane, it, at.hess_mode, gn_ok);
            if (status >= 0) break;
            if (lane == 0) ric_note_err(st, RL(R_park)[5]);
        }
        { RPROF_START(); if (ok) ok = ric_forward<NT, SPLIT>(AH, lds, b, lane); RPROF(6); }
        if (ok) { ksel = (RESUME && att < A.natt) ? att : 0; break; }
        at.dw = RL(R_park)[12];                          // (delta_w lives in LDS across the sweeps, not in a register)
        ric_attempt_next(at, gn_ok, RL(R_park)[14], o.dw0);
#ifdef BMPC_EMU_TRACE
        if (lane == 0 && getenv("BMPC_EMU_TRACE")) printf("[b %d]      attempt %d: hess_mode %d dw %.3e\n", A.src[b], at.tries, at.hess_mode, at.dw);
#endif
        BMPC_SYNC();
        if (lane == 0) { RL(R_park)[12] = at.dw; RL(R_park)[15] = 0.0; }
        if (at.dead) { status = 3; break; }
    }
    BMPC_SYNC();
    const double mu = RL(R_park)[11];
    at.dw = RL(R_park)[12];
#ifdef BMPC_PROFILE
    BMPC_SYNC();
    if (lane < 16) atomicAdd(A.prof + lane, RL(R_misc)[32 + lane]);
#endif
    if (lane == 0) ric_finish(A, lds, b, status, at, hess_mode_in, mu, ksel);
}

// The first kernel of the speculative pair: workgroup (instance e, attempt a) runs the backward sweep of attempt a of the sequence
// above -- attempt 0 with the KKT test and the barrier update, as k_ric_body; attempts >= 1 stop at the first control block that is
// not positive definite -- and leaves in A.fspec what k_ric_sel needs: verdict, (a = 0) status / barrier parameter / KKT error, and
// the rows of P and the entries of the value-function gradients that the forward start reads.  Gains go to the attempt's own copy.
// Nothing any sibling attempt reads is written.  In the straggler tail the slowest instance of a launch needs two or three sweeps;
// side by side they cost the time of one.
template <int NT, bool SPLIT>
BMPC_DEV void k_ric_att_body(RicArgs AH, int blk, int lane, LDSD* lds) {
    RicArgsRef A = ric_args(AH);
    const int count = A.L.cnt[0];
    if (blk >= count * A.natt) return;
    const int att = blk / count, e = blk - att * count;
    const int b = BMPC_UNIFORM(A.L.eval[e]);
    const auto& o = A.o;
    InstState* st = A.st + b;
    if (att == 0 && lane == 0) BMPC_ATOMIC_INC(A.L.cnt + 12);      // counted with the latency regime whichever compilation of the sweeps runs (bench.py: flops per launch time)
    if (att == 0) ric_load_kkt_sums(A, lds, b, lane);
    RicAttempt at;
    at.hess_mode = BMPC_UNIFORM(st->hess_mode); at.tries = 0; at.gn_fell = 0; at.dead = 0; at.dw = 0.0;
    const int it = BMPC_UNIFORM(st->it);
    const int gn_ok = BMPC_UNIFORM((o.inertia == 0 || (o.inertia == 2 && st->err_prev > o.inertia_err && st->stall < o.stall_n)) ? 1 : 0);
    const double dwl = st->dw_last;
    for (int i = 0; i < att; i++) ric_attempt_next(at, gn_ok, dwl, o.dw0);
    double* fs = A.fspec + ((size_t)b * RIC_NATT + att) * RIC_FS;
    if (at.dead) { if (lane == 0) fs[0] = 0.0; return; }
    BMPC_SYNC();
    if (lane == 0) { RL(R_park)[12] = at.dw; RL(R_park)[11] = st->mu; RL(R_park)[14] = dwl; RL(R_park)[15] = (double)att; RL(R_park)[5] = 0.0; }
    BMPC_SYNC();
    const bool ok = ric_backward<NT, SPLIT>(AH, lds, b, lane, at.hess_mode, att == 0 ? 0 : 1);
    int status = -1;
    if (att == 0) status = ric_kkt_and_barrier<SPLIT>(AH, lds, b, lane, it, at.hess_mode, gn_ok);
    BMPC_SYNC();
    if (lane == 0) { fs[0] = ok ? 1.0 : 0.0; fs[1] = (double)status; fs[2] = RL(R_park)[11]; fs[3] = RL(R_park)[5]; }
    if (ok) {
        for (int i = lane; i < RIC_FS_P; i += NT) fs[8 + i] = RL(R_P)[RIC_FS_P0 + i];
        if (lane < 8) { fs[8 + RIC_FS_P + lane] = RL(R_pv0)[24 + lane]; fs[8 + RIC_FS_P + 8 + lane] = RL(R_pv1)[24 + lane]; }
    }
}

// ------------------------------------------------------------------------------------------
// per-instance control kernels (one thread per list entry)
// ------------------------------------------------------------------------------------------
BMPC_INL void inst_reset(const PipeArgs& A, int slot) {
    GST st = A.st + slot;
    st->state = ST_EVAL; st->it = 0; st->status = 1; st->nfilt = 0; st->hess_mode = 0; st->bt = 0; st->armijo = 0; st->tries = 0;
    st->flip = 0; st->stall = 0; st->dw_last = 0; st->err_best = 1e300; st->gn_skip = 0; st->gn_back = 0;
    st->mu = A.o.mu_init; st->alpha = 0; st->ad = 0; st->ap = 0; st->hreg = 0; st->err_prev = 1e300; st->filt_mu = -1;
    st->theta_max = 1e300; st->theta_min = 0; st->fk = 0; st->ksel = 0;
}

// first fill of the pool: slot i takes input row i (host sets cnt[0] = cnt[9] = cnt[6] = number of slots filled)
BMPC_DEV void k_init_inst_body(const PipeArgs& A, int i) {
    if (i >= A.L.cnt[9]) return;
    A.src[i] = i;
    inst_reset(A, i);
    A.L.eval[i] = i; A.L.admit[i] = i;
}

// retirement: every slot of the done list (its outputs have just been written by k_out / k_fin) takes the next input
// row, if there is one, and joins the eval and admit lists (k_init then builds its first iterate)
BMPC_DEV void k_admit_body(const PipeArgs& A, int e) {
    if (e >= A.L.cnt[8]) return;
    const int slot = A.L.done[e];
    int row;
    if (A.cont) {
        // closed loop: the caller's retire hook has post-processed the solution of this row and, if the rollout goes on,
        // prepared its next problem in place: the slot keeps its row
        row = A.src[slot];
        if (!A.cont[row]) { BMPC_ATOMIC_INC(A.L.cnt + 7); return; }
    } else {
        BMPC_ATOMIC_INC(A.L.cnt + 7);
        row = BMPC_ATOMIC_INC(A.L.cnt + 6);
        if (row >= A.B) return;                    // no input left: the slot stays empty
    }
    A.src[slot] = row;
    inst_reset(A, slot);
    int pos = BMPC_ATOMIC_INC(A.L.cnt + 9);
    A.L.admit[pos] = slot;
    pos = BMPC_ATOMIC_INC(A.L.cnt + 0);
    A.L.eval[pos] = slot;
}

// after k_init: merit pieces (f, theta, sum log t) of the initial point (admit list)
BMPC_DEV void k_init_fin_body(const PipeArgs& A, int e) {
    if (e >= A.L.cnt[9]) return;
    const int b = A.L.admit[e];
    GST st = A.st + b;
    GCD P = A.part + pair_of(A, b, 1);
    double f1 = 0, th1 = 0, ls1 = 0;
    for (int k = 0; k < A.N - 1; k++) { f1 += P[PT_F1 * A.NP + k]; th1 += P[PT_TH1 * A.NP + k]; ls1 += P[PT_LS1 * A.NP + k]; }
    st->f0 = f1; st->th0 = th1; st->ls0 = ls1;
}

// the done and admit lists have been consumed
BMPC_DEV void k_pool_reset_body(const PipeArgs& A, bool done_too) {
    if (done_too) A.L.cnt[8] = 0;
    A.L.cnt[9] = 0;
}

// rotate the list counters between super-steps (one thread)
BMPC_DEV void k_rotate_body(const PipeArgs& A) {
    GI c = A.L.cnt;
    c[0] = c[3]; c[3] = 0; c[1] = 0; c[2] = c[4]; c[4] = 0; c[10] = 0;
}

// per-instance outputs after k_out (done list)
BMPC_DEV void k_fin_body(const PipeArgs& A, int e) {
    if (e >= A.L.cnt[8]) return;
    const int b = A.L.done[e];
    const GST st = A.st + b;
    GCD P = A.part + pair_of(A, b, 1);
    double v = 0;
    for (int k = 0; k < A.N - 1; k++) v += P[PT_F1 * A.NP + k];
    const int row = A.src[b];
    A.viol[row] = v;
    A.f[row] = st->fk;
    A.iters[row] = st->it;
    A.status[row] = st->status;
}

// adjoint sweep of the multiplier recovery (see k_mult): one thread per instance, backwards over the stages
BMPC_DEV void k_mult_sweep_body(const PipeArgs& A, int b) {
    if (b >= A.B) return;
    const int N = A.N, n_w = 44 * N + 6, n_g = 147 * (N - 1) + 21;
    const double dt = A.o.dt;
    GD lg = A.lam_g + (size_t)A.src[b] * n_g;
    GD lx = A.lam_x + (size_t)A.src[b] * n_w;
    double lq[7], ldq[7], lddq[7], lprot[3] = {0, 0, 0}, lrs = 0, lps = 0;
    for (int j = 0; j < 7; j++) { lq[j] = 0; ldq[j] = 0; lddq[j] = 0; }
    for (int k = N - 1; k >= 1; k--) {
        GCD rec = A.hrec + hrec_of(A, b, k);
        GD blk = lg + 35 * (k - 1);
        const double n0 = lprot[0], n1 = lprot[1], n2 = lprot[2];          // lam_p_rot of the next block
        double nq[7], ndq[7], nddq[7];
        for (int j = 0; j < 7; j++) {
            const double gl = dt * (rec[M_GANG + j] * n0 + rec[M_GANG + 7 + j] * n1 + rec[M_GANG + 14 + j] * n2);
            const double zl = dt * (rec[M_ZX + j] * n0 + rec[M_ZX + 7 + j] * n1 + rec[M_ZX + 14 + j] * n2);
            nq[j] = rec[M_CQ + j] + lq[j] + gl;
            ndq[j] = rec[M_CDQ + j] + ldq[j] + dt * lq[j] + zl;
            nddq[j] = rec[M_CDDQ + j] + lddq[j] + dt * ldq[j] + 0.5 * dt * dt * lq[j];
        }
        for (int j = 0; j < 7; j++) { lq[j] = nq[j]; ldq[j] = ndq[j]; lddq[j] = nddq[j]; blk[j] = nq[j]; blk[7 + j] = ndq[j]; blk[14 + j] = nddq[j]; }
        for (int a = 0; a < 3; a++) {
            const double nx = (a == 0 ? n0 : a == 1 ? n1 : n2);
            lprot[a] = rec[M_BZ + 3 + a] + nx;
            blk[21 + a] = rec[M_BZ + a];
            blk[24 + a] = lprot[a];
            blk[27 + a] = rec[M_BZV + a];
            blk[30 + a] = rec[M_BZV + 3 + a] + 0.5 * dt * (lprot[a] + nx);
        }
        lrs = rec[M_GRS] + lrs; lps = rec[M_GPS] + lps;
        blk[33] = lrs; blk[34] = lps;
    }
    // stage-0 variables (pinned by lbx == ubx / the eliminated stage-0 slacks): lam_x = -(grad f + J_g^T lam_g), what
    // IPOPT reports for fixed variables under fixed_variable_treatment=make_parameter
    for (int j = 0; j < 7; j++) {
        lx[(size_t)j * N] = -lq[j];
        lx[(size_t)(7 + j) * N] = -(ldq[j] + dt * lq[j]);
        lx[(size_t)(14 + j) * N] = -(lddq[j] + dt * ldq[j] + 0.5 * dt * dt * lq[j]);
        lx[(size_t)(21 + j) * N] = -(dt * dt * dt / 8 * lq[j] + dt * dt / 3 * ldq[j] + dt / 2 * lddq[j]);
    }
    for (int a = 0; a < 3; a++) lx[(size_t)(37 + a) * N] = -0.5 * dt * lprot[a];      // v_0 (angular); p_0: zero (the NLP
    lx[(size_t)40 * N + 6] = -lrs; lx[(size_t)41 * N + 6] = -0.5 * dt * lrs;           // is invariant to a common shift of
    lx[(size_t)42 * N + 6] = -lps; lx[(size_t)43 * N + 6] = -0.5 * dt * lps;           // all p_rot_k)
}

#undef RL
}  // namespace bmpc
