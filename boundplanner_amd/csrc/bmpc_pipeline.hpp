// bmpc_pipeline.hpp -- batch-synchronous interior-point pipeline (device code, gfx950).
//
// One interior-point iteration (oracle/bmpc_solve.c documents the algebra) is split by its parallelism:
//   * everything that is independent per (instance, stage) pair -- kinematics, rows, Hessian
//     assembly, row steps, line-search trial evaluation -- runs ONE THREAD PER PAIR
//     (bmpc_stage.hpp), iterate and row data in SoA arrays [field][pair] so that consecutive
//     lanes touch consecutive doubles (coalesced HBM access across the batch);
//   * the only sequential part, the Riccati recursion over the horizon (banded KKT solve), runs
//     ONE WAVEFRONT PER INSTANCE with the value-function Hessian P and the stage matrix W in LDS;
//   * tiny per-instance control kernels (filter line search, barrier schedule) advance a state
//     machine; work lists are compacted by atomic append, so finished instances cost nothing.
// The algorithm (and every constant) is that of oracle/bmpc_solve.c; DESIGN.md documents the data
// layout and the launch sequence.
#pragma once
#include "bmpc_stage.hpp"

namespace bmpc {

// ------------------------------------------------------------------------------------------
// stage record written by k_eval (thread per pair), read by k_ric (wave per instance):
// natural-coordinate Hessian pieces + gradients + dynamics linearisation, AoS [pair][HREC]
// ------------------------------------------------------------------------------------------
constexpr int F_CD = 0;                 // [c<5][i<7]    W[q_i][d_c]
constexpr int F_P17 = F_CD + 35;        // [i<17]{C3[PS], C3[RS], C3[D5], D, g0, g1, gz} of position pos17(i)
constexpr int F_H17 = F_P17 + 119;      // columns j=0..16, rows i<=j   (q, dq, pi) block
constexpr int F_EW = F_H17 + 153;       // G_ang[3][7], J_ang[3][7]   (emitted while k_eval still holds the kinematic columns)
constexpr int F_SUFZ = F_EW + 42;       // sufz[1..7][3] = sum_{j>=m} z_j dq_j
constexpr int F_DGR = F_SUFZ + 21;      // [dg positions 14..37]{D, g0, g1, gz}: ddq, u, rs, drs, ps, dps, d
constexpr int F_DZ2 = F_DGR + 96;       // zeta-diagonal rows (k == 1): sigma of rs~_1, ps~_1
constexpr int F_GZ2 = F_DZ2 + 2;        // [r0, r1, zz][2]
constexpr int F_RDEF = F_GZ2 + 6;       // dynamics defect (32)
constexpr int F_MAIN_END = F_RDEF + 32; // 506: written by k_eval (padded to 512)
constexpr int F_CQP = 512;              // written by k_curv: [i][a] q_i x pi_a block of the sigmoid-weighted error terms' curvature
constexpr int F_CQQ = F_CQP + 21;       // [a][b] second-order terms, q x q
constexpr int F_CQD = F_CQQ + 49;       // [i][j] q_i x dq_j
constexpr int F_END = F_CQD + 49;
constexpr int HREC = 640;               // multiple of 64: k_ric reads it with 10 unconditional loads per lane
static_assert(F_MAIN_END <= F_CQP && F_CQP % 16 == 0 && F_END <= HREC && HREC % 64 == 0, "record layout");

// DG order: q, dq, ddq, u, rs, drs, ps, dps, d, pi
BMPC_HD int dg_pos(int i) {
    if (i < 7) return Z_Q + i;
    if (i < 14) return Z_DQ + i - 7;
    if (i < 21) return Z_DDQ + i - 14;
    if (i < 28) return Z_U + i - 21;
    if (i == 28) return Z_RS;
    if (i == 29) return Z_DRS;
    if (i == 30) return Z_PS;
    if (i == 31) return Z_DPS;
    if (i < 38) return Z_D + i - 32;
    return Z_PI + i - 38;
}
constexpr int dg_pos_c(int i) {
    return i < 7 ? Z_Q + i : i < 14 ? Z_DQ + i - 7 : i < 21 ? Z_DDQ + i - 14 : i < 28 ? Z_U + i - 21 : i == 28 ? Z_RS
           : i == 29 ? Z_DRS : i == 30 ? Z_PS : i == 31 ? Z_DPS : i < 38 ? Z_D + i - 32 : Z_PI + i - 38;
}
BMPC_HD int pos17(int i) { return i < 7 ? Z_Q + i : (i < 14 ? Z_DQ + i - 7 : Z_PI + i - 14); }

constexpr int KREC = 320;               // gains per pair: K (9x32) + kf (2x16)
// speculative factorisation attempts (k_ric_att / k_ric_sel, bmpc_ric_kernel.hpp): attempts run side by side per instance, and per
// (instance, attempt) the doubles k_ric_sel needs: [0] sweep succeeded [1] status [2] barrier parameter [3] KKT error, [8 ..) rows
// 24 .. 31 of the packed P, then pv0[24..31], pv1[24..31]
constexpr int RIC_NATT = 5 /* at most; PipeArgs.natt of them run */, RIC_FS_P0 = 24 * 25 / 2, RIC_FS_P = NPSYM - RIC_FS_P0, RIC_FS = 256;
static_assert(8 + RIC_FS_P + 16 <= RIC_FS, "forward-start record");
constexpr int NPART = 272;              // per-pair partial sums (16) + forces for k_curv (27) + point-group results (121) + PT_SIG (10) + pose-row results (88)
constexpr int PT_FORCE = 16;            // Fp[3], Fv[6], Fc[6][3]
constexpr int PT_SIG = 169;             // k_eval -> k_curv: c1 = 2 sig sig'' |e|^2, dpp[3], 2 sig sig' De^T e [6] (exact curvature of sig^2 |e|^2)
// results of the collision-point rows (k_points), read by k_eval where it needs them
constexpr int PT_SIDE = 48, SD_CD = 0 /*[5][7]*/, SD_CD5 = 35 /*7*/, SD_HQQ = 42 /*28*/, SD_GQ = 70 /*[3][7]*/,
              SD_DD = 91 /*6*/, SD_GD = 97 /*[3][6]*/, SD_KKT = 115 /*cmax csum cmin zsum prim nrows*/, SD_END = 121;
// results of the pose rows and of the output-space cost (k_pose), read by k_eval: pose-space Hessian / slack couplings / gradients
constexpr int PT_POSE = 184, PZ_M6 = 0 /*21*/, PZ_MS = 21 /*[3][6]*/, PZ_SS = 39 /*3*/, PZ_BP0 = 42 /*6*/, PZ_BP1 = 48, PZ_BPZ = 54,
              PZ_BS0 = 60 /*3*/, PZ_BS1 = 63, PZ_BSZ = 66, PZ_BV = 69 /*6: d f / d v*/, PZ_KKT = 75 /*6, as SD_KKT*/, PZ_VANG = 81 /*3*/, PZ_END = 84;
static_assert(PT_SIDE + SD_END <= PT_SIG && PT_SIG + 10 <= PT_POSE && PT_POSE + PZ_END <= NPART, "partials layout");
enum { PT_CMAX = 0, PT_CSUM, PT_CMIN, PT_ZSUM, PT_PRIM, PT_THETA, PT_LOGS, PT_NROWS, PT_FVAL,
       PT_AP, PT_AD, PT_DBAR, PT_DPHIF, PT_F1, PT_TH1, PT_LS1 };

// per-pair record of k_mult (written over the stage record once the solve is over), read by k_mult_sweep
constexpr int M_CQ = 0, M_CDQ = 7, M_CDDQ = 14, M_GANG = 21 /*[3][7]*/, M_ZX = 42 /*[3][7]*/, M_BZ = 63 /*6*/, M_BZV = 69 /*6*/,
              M_GRS = 75, M_GPS = 76, M_END = 77;
static_assert(M_END <= HREC, "multiplier record fits the stage record");

enum { ST_EVAL = 0, ST_STEP = 1, ST_TRIAL = 2, ST_DONE = 3 };

// per-instance solver state (AoS, one per instance)
struct InstState {
    int state, it, status, nfilt, hess_mode, bt, armijo, tries;
    int flip, stall;             // stall: iterations since the KKT error last improved by 10 %; flip: which copy of the double-buffered arrays (t / t_t, zeta / zeta_t) holds the iterate (cur_* below)
    double mu, alpha, ad, ap, hreg, err_prev, filt_mu, theta_max, theta_min;
    int gn_skip, gn_back;        // iterations for which the exact Hessian is not tried (after a Gauss-Newton fallback) / current back-off
    int ksel, pad_;              // copy of the gains the forward recursion reads (0: the slot's own; > 0: a speculative attempt's, ric_krec)
    double dw_last, err_best;    // last successful inertia correction delta_w (0 = none yet); best KKT error so far
    double f0, th0, ls0, D, phi0, fk;
    double filt_th[8], filt_phi[8];
};

// PipeArgsT<0>: plain pointers (host side, launch arguments); PipeArgsT<1>: the same layout with
// global-address-space pointers, which is how the device code reads it
template <int DEV> struct PtrT {
    typedef double* D; typedef const double* CD; typedef int* I; typedef const int* CI;
    typedef struct InstState* S; typedef const RobotConst* RC;
};
template <> struct PtrT<1> {
    typedef GD D; typedef GCD CD; typedef GI I; typedef GCI CI;
    typedef BMPC_AS1 struct InstState* S; typedef GRC RC;
};

template <int DEV> struct ListsT {           // work lists (slot ids) with their counters
    typename PtrT<DEV>::I eval, step, trial, eval_next, trial_next;
    typename PtrT<DEV>::I curv;          // slots whose instance is in hess_mode this super-step (k_points appends, k_curv consumes)
    typename PtrT<DEV>::I done, admit;   // slots whose instance finished (to retire) / slots that got a new instance (to initialise)
    typename PtrT<DEV>::I cnt;   // [0] n_eval [1] n_step [2] n_trial [3] n_eval_next [4] n_trial_next [5] finished (cumulative)
                                 // [6] next input row to admit [7] retired (cumulative) [8] n_done [9] n_admit [10] n_curv
                                 // [11] / [12] workgroups (= instance-iterations) run by bmpc_k_ric / bmpc_k_ric_lat (cumulative)
};
constexpr int NCNT = 14;

template <int DEV> struct PipeArgsT {
    typedef PtrT<DEV> PT;
    int B, N;
    int natt, pad0_;                     // speculative factorisation attempts per instance in k_ric_att / k_ric_sel (<= RIC_NATT)
    SolverOpts o;
    typename PT::RC rc;
    typename PT::CD x0, lbx, ubx, p;
    typename PT::D x, f, viol, g;
    typename PT::I iters, status;
    // workspace
    // element (field f, slot b, stage k) of an SoA array lives at  f * NP + b * SS + (k - 1);  the record of (b, k) at
    // b * HS + (k - 1) * HREC (KS, KREC for the gains).  Two layouts (pipe_carve): field-major (NP = pairs of the whole pool,
    // SS = N-1) and slot-major (all arrays of a slot in one block: NP = N-1, SS = HS = KS = block size)
    size_t NP, SS, HS, KS;
    typename PT::D zeta, zeta_t, dz;     // 41 fields
    typename PT::D t, t_t, z, dt, z_t;   // NSLOT fields (dt holds t + dt, k_step's `c`)
    typename PT::D hrec;                 // HREC doubles per pair
    typename PT::D krec;                 // KREC doubles per pair
    typename PT::D dx1;                  // [B][32] step of x_1 (k_ric -> k_fwd)
    typename PT::D kspec, fspec;         // speculative attempts: [slot][RIC_NATT-1][(N-1) KREC] gains, [slot][RIC_NATT][RIC_FS] forward-start records
    typename PT::D part;                 // NPART fields
    typename PT::S st;                   // [slots]
    typename PT::I src;                  // [slots] input / output row of the instance in the slot (streaming: B rows
                                         // pass through fewer slots, a slot is refilled when its instance has retired)
    ListsT<DEV> L;
    typename PT::CI tbl;                 // scatter table of the stage record (3 ints per field)
    typename PT::D prof;                 // diagnostic builds (-DBMPC_PROFILE): phase cycle sums, else unused
    typename PT::D lam_g, lam_x;         // multiplier outputs of k_mult / k_mult_sweep ([B][n_g], [B][n_w]) or null
    typename PT::CI cont;                // closed loop (bmpc_loop_run_async): [rows] 1 = the row has another problem ready (its slot
                                         // is re-admitted with the same row when it retires), else null
};
typedef PipeArgsT<0> PipeArgsH;          // host view
typedef PipeArgsT<1> PipeArgs;           // device view (same layout)
static_assert(sizeof(PipeArgsH) == sizeof(PipeArgs), "host/device argument layouts differ");
typedef BMPC_AS1 InstState* GST;

// The iterate's slacks, row multipliers and zeta live in t / z / zeta or in t_t / z_t / zeta_t, per instance (InstState.flip):
// k_trial writes the trial point (and the updated multipliers, which do not depend on the primal step length) into the other
// copy and an accepted trial becomes the iterate by flipping the bit -- no copy pass, no update pass.
template <class AT> BMPC_INL auto cur_t(const AT& A, int flip) -> decltype(A.t) { return flip ? A.t_t : A.t; }
template <class AT> BMPC_INL auto oth_t(const AT& A, int flip) -> decltype(A.t) { return flip ? A.t : A.t_t; }
template <class AT> BMPC_INL auto cur_z(const AT& A, int flip) -> decltype(A.z) { return flip ? A.z_t : A.z; }
template <class AT> BMPC_INL auto oth_z(const AT& A, int flip) -> decltype(A.z) { return flip ? A.z : A.z_t; }
template <class AT> BMPC_INL auto cur_zeta(const AT& A, int flip) -> decltype(A.zeta) { return flip ? A.zeta_t : A.zeta; }
template <class AT> BMPC_INL auto oth_zeta(const AT& A, int flip) -> decltype(A.zeta) { return flip ? A.zeta : A.zeta_t; }
template <class AT> BMPC_INL size_t pair_of(const AT& A, int b, int k) { return (size_t)b * A.SS + (k - 1); }      // SoA offset
template <class AT> BMPC_INL size_t hrec_of(const AT& A, int b, int k) { return (size_t)b * A.HS + (size_t)(k - 1) * HREC; }
template <class AT> BMPC_INL size_t krec_of(const AT& A, int b, int k) { return (size_t)b * A.KS + (size_t)(k - 1) * KREC; }

// workspace of `cap` slots: size in doubles, and the array bases inside it
constexpr size_t PIPE_SOA_FIELDS = 3 * (size_t)NZ + 5 * (size_t)NSLOT + NPART;
inline size_t pipe_slot_block(int N) { return ((PIPE_SOA_FIELDS + HREC + KREC) * (size_t)(N - 1) + 15) / 16 * 16 + 16; }
inline size_t pipe_np_field_major(int cap, int N) { return ((size_t)cap * (N - 1) + 63) / 64 * 64 + 64; }
inline size_t pipe_workspace_doubles(int cap, int N, int slot_major) {
    return (slot_major ? pipe_slot_block(N) * (size_t)cap + 64
                       : (PIPE_SOA_FIELDS + HREC + KREC) * pipe_np_field_major(cap, N)) + (size_t)cap * NX
           + (size_t)cap * ((size_t)(RIC_NATT - 1) * (N - 1) * KREC + (size_t)RIC_NATT * RIC_FS);
}
template <class AT> inline void pipe_carve(AT& A, double* w, int cap, int N, int slot_major) {
    const size_t S = (size_t)(N - 1);
    if (slot_major) { A.NP = S; A.SS = A.HS = A.KS = pipe_slot_block(N); }
    else { A.NP = pipe_np_field_major(cap, N); A.SS = S; A.HS = S * HREC; A.KS = S * KREC; }
    const size_t NP = A.NP;
    double* w0 = w;
    A.zeta = w; w += NZ * NP; A.zeta_t = w; w += NZ * NP; A.dz = w; w += NZ * NP;
    A.t = w; w += NSLOT * NP; A.t_t = w; w += NSLOT * NP; A.z = w; w += NSLOT * NP; A.dt = w; w += NSLOT * NP;
    A.z_t = w; w += NSLOT * NP;
    A.part = w; w += NPART * NP;
    if (slot_major) w = w0 + ((size_t)(w - w0) + 15) / 16 * 16;      // records 128-byte aligned inside the slot block
    A.hrec = w; w += HREC * NP; A.krec = w; w += KREC * NP;
    A.dx1 = slot_major ? w0 + pipe_slot_block(N) * (size_t)cap + 64 : w;
    A.kspec = A.dx1 + (size_t)cap * NX;
    A.fspec = A.kspec + (size_t)cap * (RIC_NATT - 1) * (size_t)(N - 1) * KREC;
}

// parameters of the instances of a wavefront are staged in LDS (every thread reads ~500 of them):
// pointer type of the staged copy
constexpr int EM_DOUBLES_C = 16 * 66 + 64;     // = EM_DOUBLES (emitter tile, below)
constexpr int IPW_MAX = 8;       // instances per wavefront (bounds the LDS staging area)
BMPC_HD int ipw_of(int N) { int i = 64 / (N - 1); return i < IPW_MAX ? i : IPW_MAX; }

// lanes -> pairs inside a wave: ipw_of(N) instances per wave, lane = li*(N-1) + (k-1)
struct PairMap { int b, k, li; bool valid; size_t pi; };
BMPC_INL PairMap pair_map(const PipeArgs& A, GCI list, int count, int wave, int lane) {
    const int S = A.N - 1, ipw = ipw_of(A.N);
    int li = lane / S, kk = lane - li * S;
    int e = wave * ipw + li;
    PairMap m;
    m.valid = (li < ipw) && (e < count);
    if (!m.valid) { e = wave * ipw; kk = 0; li = 0; }      // dummy work on a valid pair, no stores
    m.li = li;
    m.b = list[e]; m.k = kk + 1;
    m.pi = pair_of(A, m.b, m.k);
    return m;
}
BMPC_HD int waves_for(int N, int count) { int ipw = ipw_of(N); return (count + ipw - 1) / ipw; }

// stage the parameter vectors of this wavefront's instances: lds_par[li][NPAR]; returns the calling
// lane's copy.  All 64 lanes must call it (contains a barrier).
// BATCHED: three instances at a time, all loads issued before the first LDS store (the plain loop is kept for A/B runs)
// activity masks of the halfspace sets of the staged instances (PL_MASKJ, PL_MASKE): computed once per wavefront; the row walkers
// test a bit instead of four parameters per row, and the visitors load the row data (t, z, c) of constraint rows only -- on
// BASELINE configs[2] a fifth of the 90 collision-point slots is padding (12 of 15 rows per set on average)
template <int NTHR = 64>
BMPC_INL void stage_masks(LDSD* lds_par, int ipw, int wave, int lane, int count) {
    for (int e = lane; e < ipw * 10; e += NTHR) {
        const int li = e / 10, c = e - 10 * li;
        if (wave * ipw + li < count) {
            PGP pg = lds_par + li * NPARL;
            unsigned mk = 0;
            for (int rr = 0; rr < 15; rr++) {
                PGP a = c < 6 ? pg + P_ASETJ + 45 * c : pg + P_ASET + 45 * (c - 6);
                const double bb = c < 6 ? pg[P_BSETJ + rr * 6 + c] + pg[P_SLACKS0 + c] : pg[P_BSET + rr * 4 + (c - 6)];
                if (!(a[rr] == 0 && a[rr + 15] == 0 && a[rr + 30] == 0 && bb > 0)) mk |= 1u << rr;
            }
            lds_par[li * NPARL + NPAR + c] = (double)mk;
        }
    }
    BMPC_SYNC();
}
// NTHR: threads of the workgroup (all of them must call it; `lane` = thread index in the workgroup)
template <bool BATCHED = true, int NTHR = 64>
BMPC_INL PGP stage_params(const PipeArgs& A, GCI list, int count, int wave, int lane, const PairMap& m, LDSD* lds_par) {
    const int ipw = ipw_of(A.N);
    if constexpr (!BATCHED) {
        for (int li = 0; li < ipw; li++) {
            const int e = wave * ipw + li;
            if (e < count) {
                GCD src = A.p + (size_t)A.src[list ? list[e] : e] * NPAR;
                for (int i = lane; i < NPAR; i += NTHR) lds_par[li * NPARL + i] = src[i];
            }
        }
        BMPC_SYNC();
        stage_masks<NTHR>(lds_par, ipw, wave, lane, count);
        return lds_par + m.li * NPARL;
    }
    // three instances at a time: all their loads are issued before the first LDS store (one memory round trip per chunk)
    constexpr int NJ = (NPAR + NTHR - 1) / NTHR;
    for (int l0 = 0; l0 < ipw; l0 += 3) {
        double v[3][NJ];
        BMPC_UNROLL
        for (int c = 0; c < 3; c++) {
            const int li = l0 + c, e = wave * ipw + li;
            const bool on = li < ipw && e < count;
            GCD src = A.p + (size_t)A.src[on ? (list ? list[e] : e) : (list ? list[wave * ipw] : wave * ipw)] * NPAR;
            BMPC_UNROLL
            for (int j = 0; j < NJ; j++) { const int i = lane + NTHR * j; v[c][j] = (i < NPAR) ? src[i] : 0.0; }
        }
        BMPC_UNROLL
        for (int c = 0; c < 3; c++) {
            const int li = l0 + c, e = wave * ipw + li;
            if (li < ipw && e < count)
                BMPC_UNROLL
                for (int j = 0; j < NJ; j++) { const int i = lane + NTHR * j; if (i < NPAR) lds_par[li * NPARL + i] = v[c][j]; }
        }
    }
    BMPC_SYNC();
    stage_masks<NTHR>(lds_par, ipw, wave, lane, count);
    return lds_par + m.li * NPARL;
}
constexpr int TRIAL_COMB = 5;      // partial results of one part of a trial point (k_trial with four wavefronts per group of pairs)
BMPC_HD size_t trial_lds_doubles(int N, int nw) { return (size_t)ipw_of(N) * NPARL + IPW_MAX + (nw > 1 ? (size_t)nw * 64 * TRIAL_COMB : 0); }
BMPC_HD size_t pair_lds_doubles(int N, bool with_tile) { return (size_t)ipw_of(N) * NPARL + (with_tile ? EM_DOUBLES_C + 8 : 0); }

// ------------------------------------------------------------------------------------------
// coalesced AoS output of per-thread records: 16 fields at a time through an LDS tile
// ------------------------------------------------------------------------------------------
constexpr int EM_LD = 66;                      // tile row stride (doubles): conflict-free both ways
constexpr int EM_DOUBLES = 16 * EM_LD + 64;    // tile + per-lane record base (as double)
static_assert(EM_DOUBLES == EM_DOUBLES_C, "emitter tile size");
// HOLE: fields [h0, h1) of the record belong to another writer (k_eval without the chained block, tail regime) -- they pass through the
// tile like the others but are not stored
template <bool HOLE> struct EmitterT {
    LDSD* tile;        // [16][EM_LD]
    GD out;            // record array base
    int lane, f;
    int h0, h1;
    BMPC_INL void init(LDSD* lds, GD out_, int lane_, size_t rec, bool valid, int f0 = 0, int hole0 = 0, int hole1 = 0) {
        tile = lds; out = out_; lane = lane_; f = f0;     // f0: multiple of 16; rec: offset of this lane's record in `out`
        h0 = hole0; h1 = hole1;
        // publish every lane's record base (exact in a double: < 2^53) for the transposed store;
        // -1 = this lane must not store
        lds[16 * EM_LD + lane] = valid ? (double)rec : -1.0;
    }
    BMPC_INL void put(double v) {
        tile[(f & 15) * EM_LD + lane] = v;
        f++;
        if ((f & 15) == 0) flush();
    }
    BMPC_INL void flush() {
        BMPC_SYNC();
        const int ff = lane & 15, p0 = lane >> 4, c0 = f - 16;
        bool mine = true;
        if constexpr (HOLE) mine = (c0 + ff < h0) || (c0 + ff >= h1);
        BMPC_UNROLL
        for (int i = 0; i < 16; i++) {
            int pr = 4 * i + p0;
            double base = tile[16 * EM_LD + pr];
            double v = tile[ff * EM_LD + pr];
            if (base >= 0.0 && mine) out[(size_t)base + c0 + ff] = v;
        }
        BMPC_SYNC();
    }
    BMPC_INL void pad_to(int n) { while (f < n) put(0.0); }
};
typedef EmitterT<false> Emitter;
// the chained block alone (k_eval's chain part, tail regime): straight into the lane's record
struct DirectEmitter {
    GD out; bool valid; int f;
    BMPC_INL void put(double v) { if (valid) out[f] = v; f++; }
};

}  // namespace bmpc
