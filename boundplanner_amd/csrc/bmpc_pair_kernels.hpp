// bmpc_pair_kernels.hpp -- the thread-per-(instance, stage) kernels of the pipeline:
//   k_init   initial iterate from x0 (BoundMPC.py:412-416 warm/cold start vector), row slacks
//   k_points collision-point rows; k_pose reference / error context, cost, pose rows -> pose-space sums (side array)
//   k_eval   kinematic columns, chaining of the pose-space sums, diagonal rows, defect -> the stage record for the Riccati sweep
//   k_step   row steps of the Newton direction, fraction-to-boundary partials, merit derivative
//   k_trial  line-search trial points (+ the multipliers' update), filter test, backtracking
//   k_out    solution in the reference layout (casadi_ocp_formulation.py:89-101), g, violation
// Bodies are plain functions of (args, wave, lane, lds) so that tests/emu can run them on the host.
#pragma once
#include "bmpc_pipeline.hpp"

namespace bmpc {

BMPC_INL constexpr int sym7(int i, int j) { return i <= j ? (i * 7 - i * (i - 1) / 2 + (j - i)) : (j * 7 - j * (j - 1) / 2 + (i - j)); }
BMPC_INL constexpr int sym3(int i, int j) { return i <= j ? (i * 3 - i * (i - 1) / 2 + (j - i)) : (j * 3 - j * (j - 1) / 2 + (i - j)); }

// everything the pair kernels need about one stage point
struct StagePoint {
    double zeta[NZ], y[NZ];
    KinT K;
    double Jl[3][7];
    SegCtx C;
};

BMPC_INL void load_zeta(GCD arr, size_t NP, size_t pi, double* z) {
    BMPC_UNROLL
    for (int i = 0; i < NZ; i++) z[i] = arr[(size_t)i * NP + pi];
}

// kinematics + context at zeta (values only; Jacobian G computed by the caller when needed)
BMPC_INL void stage_point(const PipeArgs& A, PGP pg, const double* iw0, int k, const DynC dc, StagePoint& S) {
    nat_all(S.zeta, dc, S.y);
    kin_chain(A.rc, S.y + Z_Q, S.K);
    kin_jlin(S.K, S.Jl);
    kin_vel(S.K, S.Jl, S.y + Z_DQ, S.C.v);
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) { S.C.pose[a] = S.K.pee[a]; S.C.pose[3 + a] = S.y[Z_PI + a] + 0.5 * dc.dt * S.C.v[3 + a]; }
    seg_ctx_eval(pg, A.N, k, S.y, iw0, S.C);
}

// dynamics defect of stage k given zeta_{k+1}[0..31]
BMPC_INL void defect_all(const double* z, const double* zn, const double* vang, const DynC d, double* r) {
    BMPC_UNROLL
    for (int i = 0; i < 7; i++) {
        r[Z_Q + i] = z[Z_Q + i] + d.dt * z[Z_DQ + i] + 0.5 * d.dt * d.dt * z[Z_DDQ + i] + d.b3 * z[Z_U + i] - zn[Z_Q + i];
        r[Z_DQ + i] = z[Z_DQ + i] + d.dt * z[Z_DDQ + i] + d.b2 * z[Z_U + i] - zn[Z_DQ + i];
        r[Z_DDQ + i] = z[Z_DDQ + i] + d.b1 * z[Z_U + i] - zn[Z_DDQ + i];
    }
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) r[Z_PI + a] = z[Z_PI + a] + d.dt * vang[a] - zn[Z_PI + a];
    r[Z_RS] = z[Z_RS] + d.dt * z[Z_DRS] - zn[Z_RS];
    r[Z_PS] = z[Z_PS] + d.dt * z[Z_DPS] - zn[Z_PS];
    BMPC_UNROLL
    for (int i = 0; i < 6; i++) r[Z_D + i] = z[Z_D + i] - zn[Z_D + i];
}

// ------------------------------------------------------------------------------------------
// k_init
// ------------------------------------------------------------------------------------------
struct InitVisitor {
    const PipeArgs* A; size_t pi; bool valid;
    double th, ls;
    BMPC_INL void set(int s, double h) {
        double t = fmax(-h, 1e-2);
        th += fabs(h + t); ls += log(t);
        if (!valid) return;
        size_t o = (size_t)s * A->NP + pi;
        A->t[o] = t; A->t_t[o] = t; A->z[o] = 1.0; A->z_t[o] = 1.0;
    }
    BMPC_INL void skip(int s) {
        if (!valid) return;
        size_t o = (size_t)s * A->NP + pi;
        A->t[o] = 1.0; A->t_t[o] = 1.0; A->z[o] = 0.0; A->z_t[o] = 0.0;
    }
    BMPC_INL void diag(int s, int, double, double h) { set(s, h); }
    BMPC_INL void zdiag(int s, int, double, double h) { set(s, h); }
    BMPC_INL void pose(int s, const double*, int, double h) { set(s, h); }
    template <int S0, int CNT> BMPC_INL void group(unsigned) {}
    template <int C> BMPC_INL void point_begin(unsigned) {}
    template <int C> BMPC_INL void point(int s, const double*, double h) { set(s, h); }
    template <int C> BMPC_INL void point_end() {}
};

// over the admit list: slots that just received an instance
BMPC_KBODY void k_init_body(const PipeArgs& A, int wave, int lane, LDSD* lds_par) {
    const int count = A.L.cnt[9], N = A.N;
    if (wave * ipw_of(N) >= count) return;
    PairMap m = pair_map(A, A.L.admit, count, wave, lane);
    const int k = m.k, n_w = 44 * N + 6;
    const DynC dc = make_dync(A.o.dt);
    const size_t row = (size_t)A.src[m.b];
    GCD x0 = A.x0 + row * n_w;
    GCD lbx = A.lbx + row * n_w;
    GCD ubx = A.ubx + row * n_w;
    PGP pg = stage_params(A, A.L.admit, count, wave, lane, m, lds_par);
    double iw0[3];
    BMPC_UNROLL
    for (int c = 0; c < 3; c++) iw0[c] = lbx[28 * N + (3 + c) * N];
    StagePoint S;
    BMPC_UNROLL
    for (int j = 0; j < 7; j++) {
        double uu = x0[21 * N + j * N + k];
        S.zeta[Z_Q + j] = x0[j * N + k] - dc.c3 * uu;
        S.zeta[Z_DQ + j] = x0[7 * N + j * N + k] - dc.c2 * uu;
        S.zeta[Z_DDQ + j] = x0[14 * N + j * N + k] - dc.c1 * uu;
        S.zeta[Z_U + j] = uu;
    }
    double prot[3];
    BMPC_UNROLL
    for (int c = 0; c < 3; c++) { prot[c] = x0[28 * N + (3 + c) * N + k]; S.zeta[Z_PI + c] = prot[c]; }
    {
        double rs = x0[40 * N + 6 + k], drs = x0[41 * N + 6 + k], ps = x0[42 * N + 6 + k], dps = x0[43 * N + 6 + k];
        S.zeta[Z_RS] = rs - dc.dt / 2 * drs; S.zeta[Z_PS] = ps - dc.dt / 2 * dps;
        S.zeta[Z_DRS] = drs; S.zeta[Z_DPS] = dps;
    }
    BMPC_UNROLL
    for (int i = 0; i < 6; i++) S.zeta[Z_D + i] = x0[40 * N + i];
    // pi_k = p_rot_k - dt/2 w(q_k, dq_k)
    nat_all(S.zeta, dc, S.y);
    kin_chain(A.rc, S.y + Z_Q, S.K);
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) {
        double w = 0;
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) w += S.K.zx[j][a] * S.y[Z_DQ + j];
        S.zeta[Z_PI + a] = prot[a] - dc.dt / 2 * w;
    }
    stage_point(A, pg, iw0, k, dc, S);
    if (m.valid)
        BMPC_UNROLL
        for (int i = 0; i < NZ; i++) { A.zeta[(size_t)i * A.NP + m.pi] = S.zeta[i]; A.zeta_t[(size_t)i * A.NP + m.pi] = S.zeta[i]; }
    InitVisitor v{&A, m.pi, m.valid, 0.0, 0.0};
    walk_rows(pg, lbx, ubx, N, k, S.y, S.zeta, S.K, S.C, v);
    if (m.valid)
        for (int s = S_END; s < NSLOT; s++) v.skip(s);
    // f, theta, sum log t of the initial point (later iterates get them from their accepted trial)
    double th = v.th;
    if (k < N - 1) {
        // zeta_{k+1} is not available yet (written by another thread of this launch): rebuild its x part
        double zn[NX];
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) {
            double uu = x0[21 * N + j * N + k + 1];
            zn[Z_Q + j] = x0[j * N + k + 1] - dc.c3 * uu;
            zn[Z_DQ + j] = x0[7 * N + j * N + k + 1] - dc.c2 * uu;
            zn[Z_DDQ + j] = x0[14 * N + j * N + k + 1] - dc.c1 * uu;
        }
        {
            double yq[7], ydq[7];
            BMPC_UNROLL
            for (int j = 0; j < 7; j++) { double uu = x0[21 * N + j * N + k + 1]; yq[j] = zn[Z_Q + j] + dc.c3 * uu; ydq[j] = zn[Z_DQ + j] + dc.c2 * uu; }
            KinT Kn;
            kin_chain(A.rc, yq, Kn);
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) {
                double w = 0;
                BMPC_UNROLL
                for (int j = 0; j < 7; j++) w += Kn.zx[j][a] * ydq[j];
                zn[Z_PI + a] = x0[28 * N + (3 + a) * N + k + 1] - dc.dt / 2 * w;
            }
        }
        zn[Z_RS] = x0[40 * N + 6 + k + 1] - dc.dt / 2 * x0[41 * N + 6 + k + 1];
        zn[Z_PS] = x0[42 * N + 6 + k + 1] - dc.dt / 2 * x0[43 * N + 6 + k + 1];
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) zn[Z_D + i] = x0[40 * N + i];
        double rdef[NX];
        defect_all(S.zeta, zn, S.C.v + 3, dc, rdef);
        BMPC_UNROLL
        for (int i = 0; i < NX; i++) th += fabs(rdef[i]);
    }
    if (k == 1) {
        double x1fix[24];
        x1fix_eval(lbx, N, dc.dt, x1fix);
        BMPC_UNROLL
        for (int i = 0; i < 24; i++) th += fabs(x1fix[i] - S.zeta[i]);
    }
    if (m.valid) {
        GD P = A.part + m.pi;
        P[PT_F1 * A.NP] = S.C.fv; P[PT_TH1 * A.NP] = th; P[PT_LS1 * A.NP] = v.ls;
    }
}

// ------------------------------------------------------------------------------------------
// k_eval
// ------------------------------------------------------------------------------------------
struct RowAcc {   // row data access (accepting the trial values) + KKT partial sums
    const PipeArgs* A; size_t pi; bool valid; double ad;
    GCD tc, zc;                                      // the copies of the slack / multiplier arrays that hold the iterate (cur_t, cur_z)
    double cmax, csum, cmin, zsum, prim, nrows;     // f, theta, sum log t of this point: k_trial / k_init
    BMPC_INL void init(const PipeArgs* A_, size_t pi_, bool valid_, double ad_, GCD tc_, GCD zc_) {
        A = A_; pi = pi_; valid = valid_; ad = ad_; tc = tc_; zc = zc_;
        cmax = 0; csum = 0; cmin = 1e300; zsum = 0; prim = 0; nrows = 0;
    }
    BMPC_INL void row(int s, double h, double& sg, double& r0, double& r1, double& zz) {
        size_t o = (size_t)s * A->NP + pi;
        row_tz(tc[o], zc[o], h, sg, r0, r1, zz);        // read-only here: the accepted trial was made current by the flip bit
    }
    // the same with the slack t and the multiplier z of the row already in registers (RowPre)
    BMPC_INL void row_tz(double t, double z_in, double h, double& sg, double& r0, double& r1, double& zz) {
        zz = z_in;
        r1 = BMPC_RCP(t); sg = zz * r1; r0 = sg * (h + t);
        double c = t * zz;
        cmax = fmax(cmax, c); csum += c; cmin = fmin(cmin, c); zsum += zz;
        prim = fmax(prim, fabs(h + t));
        nrows += 1.0;
    }
};

// Row data (t, z) of the contiguous slots S0 .. S0+CNT-1 and box bounds of the dg positions I0 .. I0+CNT-1, loaded in one
// batch ahead of their use: a thread that loads them where the row is walked waits one memory round trip per row (the
// walk is conditional, the compiler cannot hoist the loads), and those round trips are what bounds the thread-per-pair kernels
template <int S0_, int CNT_> struct RowPre {
    static constexpr int S0 = S0_, CNT = CNT_;
    double t[CNT_], z[CNT_];
    BMPC_INL void load(const PipeArgs& A, GCD tc, GCD zc, size_t pi) {
        BMPC_UNROLL
        for (int i = 0; i < CNT_; i++) { size_t o = (size_t)(S0_ + i) * A.NP + pi; t[i] = tc[o]; z[i] = zc[o]; }
    }
};
template <int I0_, int CNT_> struct BndPre {
    static constexpr int I0 = I0_, CNT = CNT_;
    double ub[CNT_], lb[CNT_];
    BMPC_INL void load(GCD lbx, GCD ubx, int N, int k) {
        BMPC_UNROLL
        for (int i = 0; i < CNT_; i++) {
            const int I = I0_ + i, blk = I / 7, jj = I % 7;
            size_t wi = (size_t)blk * 7 * N + (size_t)jj * N + k;
            ub[i] = ubx[wi]; lb[i] = lbx[wi];
        }
    }
};

struct PointAsm {
    RowAcc* R; const KinT* K;
    double M3[6], mc[3], sc, b30[3], b31[3], b3z[3], bc0, bc1, bcz;
    double Hqq[28], gq0[7], gq1[7], gqz[7], dD[6], gD0[6], gD1[6], gDz[6], cd[6][7], Fc[6][3];
    double pt[15], pz[15];        // row data of the current point's 15 slots, loaded in one batch (see RowPre)
    BMPC_INL void init() {
        BMPC_UNROLL
        for (int i = 0; i < 28; i++) Hqq[i] = 0;
        BMPC_UNROLL
        for (int i = 0; i < 7; i++) { gq0[i] = 0; gq1[i] = 0; gqz[i] = 0; }
    }
    BMPC_INL void skip(int) {}
    template <int C> BMPC_INL void point_begin(unsigned act) {
        BMPC_UNROLL
        for (int i = 0; i < 15; i++) {       // (row data of constraint rows only)
            size_t o = (size_t)(S_COL + 15 * C + i) * R->A->NP + R->pi;
            if ((act >> i) & 1u) { pt[i] = R->tc[o]; pz[i] = R->zc[o]; } else { pt[i] = 1.0; pz[i] = 0.0; }
        }
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) M3[i] = 0;
        BMPC_UNROLL
        for (int i = 0; i < 3; i++) { mc[i] = 0; b30[i] = 0; b31[i] = 0; b3z[i] = 0; }
        sc = 0; bc0 = 0; bc1 = 0; bcz = 0;
    }
    template <int C> BMPC_INL void point(int s, const double* a, double h) {
        double sg, r0, r1, zz;
        R->row_tz(pt[s - (S_COL + 15 * C)], pz[s - (S_COL + 15 * C)], h, sg, r0, r1, zz);
        BMPC_UNROLL
        for (int i = 0; i < 3; i++) {
            BMPC_UNROLL
            for (int j = i; j < 3; j++) M3[sym3(i, j)] += sg * a[i] * a[j];
            mc[i] -= sg * a[i]; b30[i] += r0 * a[i]; b31[i] += r1 * a[i]; b3z[i] += zz * a[i];
        }
        sc += sg; bc0 -= r0; bc1 -= r1; bcz -= zz;
    }
    template <int C> BMPC_INL void point_end() {
        constexpr int nj = PointNJ<C>::value;
        const double* pc = kin_point<C>(*K);
        double Jp[nj][3], T[nj][3];
        BMPC_UNROLL
        for (int i = 0; i < nj; i++) {
            double r[3] = {pc[0] - K->o[i][0], pc[1] - K->o[i][1], pc[2] - K->o[i][2]};
            cross3r(K->zx[i], r, Jp[i]);
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) T[i][a] = M3[sym3(a, 0)] * Jp[i][0] + M3[sym3(a, 1)] * Jp[i][1] + M3[sym3(a, 2)] * Jp[i][2];
        }
        BMPC_UNROLL
        for (int i = 0; i < nj; i++) {
            BMPC_UNROLL
            for (int j = i; j < nj; j++) Hqq[sym7(i, j)] += Jp[i][0] * T[j][0] + Jp[i][1] * T[j][1] + Jp[i][2] * T[j][2];
            gq0[i] += Jp[i][0] * b30[0] + Jp[i][1] * b30[1] + Jp[i][2] * b30[2];
            gq1[i] += Jp[i][0] * b31[0] + Jp[i][1] * b31[1] + Jp[i][2] * b31[2];
            gqz[i] += Jp[i][0] * b3z[0] + Jp[i][1] * b3z[1] + Jp[i][2] * b3z[2];
        }
        BMPC_UNROLL
        for (int i = 0; i < 7; i++) {
            double v = (i < nj) ? (Jp[i < nj ? i : 0][0] * mc[0] + Jp[i < nj ? i : 0][1] * mc[1] + Jp[i < nj ? i : 0][2] * mc[2]) : 0.0;
            cd[C][i] = v;
        }
        dD[C] = sc; gD0[C] = bc0; gD1[C] = bc1; gDz[C] = bcz;
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) Fc[C][a] = b3z[a];
    }
};

// results of k_points for one pair, loaded from the side array where they are needed
struct PointRes {
    GCD base; size_t NP;      // base = A.part + PT_SIDE * NP + pair
    // loaded one batch per consuming phase of k_eval (see RowPre): q-block pieces for p17_emit_all, the q x q block for
    // chain_all, the d-column pieces for dg_emit_all
    double cd5_[7], gq_[21], hqq_[28], dD_[6], gD_[18];
    BMPC_INL double at(int slot) const { return base[(size_t)slot * NP]; }
    BMPC_INL void load_p17() {
        BMPC_UNROLL
        for (int i = 0; i < 7; i++) cd5_[i] = at(SD_CD5 + i);
        BMPC_UNROLL
        for (int i = 0; i < 21; i++) gq_[i] = at(SD_GQ + i);
    }
    BMPC_INL void load_hqq() {
        BMPC_UNROLL
        for (int i = 0; i < 28; i++) hqq_[i] = at(SD_HQQ + i);
    }
    BMPC_INL void load_dg() {
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) dD_[i] = at(SD_DD + i);
        BMPC_UNROLL
        for (int i = 0; i < 18; i++) gD_[i] = at(SD_GD + i);
    }
    BMPC_INL double cd5(int i) const { return cd5_[i]; }
    BMPC_INL double hqq(int e) const { return hqq_[e]; }
    BMPC_INL double gq(int v, int i) const { return gq_[7 * v + i]; }
    BMPC_INL double dD(int c) const { return dD_[c]; }
    BMPC_INL double gD(int v, int c) const { return gD_[6 * v + c]; }
};

struct PoseAsm {
    RowAcc* R;
    double M6[21], mS[3][6], sS[3], bp0[6], bp1[6], bpz[6], bS0[3], bS1[3], bSz[3];
    BMPC_INL void init(const double* Hp, const double* g12) {
        BMPC_UNROLL
        for (int i = 0; i < 21; i++) M6[i] = Hp[i];
        BMPC_UNROLL
        for (int s = 0; s < 3; s++) {
            sS[s] = 0; bS0[s] = 0; bS1[s] = 0; bSz[s] = 0;
            BMPC_UNROLL
            for (int i = 0; i < 6; i++) mS[s][i] = 0;
        }
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) { bp0[i] = g12[i]; bpz[i] = g12[i]; bp1[i] = 0; }
    }
    // NA = number of leading nonzero coefficients (3 for position-only rows)
    template <int NA, int SEL> BMPC_INL void add(const double* a, double h, double t, double z) {
        double sg, r0, r1, zz;
        R->row_tz(t, z, h, sg, r0, r1, zz);
        BMPC_UNROLL
        for (int i = 0; i < NA; i++) {
            BMPC_UNROLL
            for (int j = i; j < NA; j++) M6[sym6(i, j)] += sg * a[i] * a[j];
            bp0[i] += r0 * a[i]; bp1[i] += r1 * a[i]; bpz[i] += zz * a[i];
            if (SEL > 0) mS[SEL > 0 ? SEL - 1 : 0][i] -= sg * a[i];
        }
        if (SEL > 0) { sS[SEL - 1] += sg; bS0[SEL - 1] -= r0; bS1[SEL - 1] -= r1; bSz[SEL - 1] -= zz; }
    }
};

// pose rows of one stage in slot order, for any visitor exposing add<NA, SEL>(a, h, t, z); row data from the preloaded
// groups g1 (S_EE .. S_ROTL+2), (t_phi, z_phi), g2 (S_TSET .. S_TROTL+2)
template <class V, class G1>
BMPC_INL void walk_pose_rows(PGP pg, int N, int k, const double* y, const SegCtx& C, V& v, const G1& g1, double t_phi, double z_phi) {
    static_assert(G1::S0 == S_EE && G1::CNT == 21, "pose row group");
    {
        PGP a = pg + P_ASET + 45 * C.s;
        BMPC_UNROLL
        for (int rr = 0; rr < 15; rr++) {
            double a3[3] = {a[rr], a[rr + 15], a[rr + 30]};
            double bb = pg[P_BSET + rr * 4 + C.s];
            if (!(a3[0] == 0 && a3[1] == 0 && a3[2] == 0 && bb > 0))
                v.template add<3, 1>(a3, a3[0] * C.pose[0] + a3[1] * C.pose[1] + a3[2] * C.pose[2] - bb - y[Z_PS], g1.t[rr], g1.z[rr]);
        }
    }
    BMPC_UNROLL
    for (int m = 0; m < 3; m++) {
        double al[6];
        BMPC_UNROLL
        for (int c = 0; c < 6; c++) al[c] = -C.gs[m][c];
        v.template add<6, 2>(C.gs[m], C.proj[m] - C.ub[m] - y[Z_RS], g1.t[S_ROTU - S_EE + m], g1.z[S_ROTU - S_EE + m]);
        v.template add<6, 2>(al, -(C.proj[m] - C.lb[m] + y[Z_RS]), g1.t[S_ROTL - S_EE + m], g1.z[S_ROTL - S_EE + m]);
    }
    v.template add<3, 0>(C.dpp, C.phi - (C.phiend + 0.005), t_phi, z_phi);
}
template <class V, class G2>
BMPC_INL void walk_pose_rows_term(PGP pg, int N, int k, const double* y, const SegCtx& C, V& v, const G2& g2) {
    static_assert(G2::S0 == S_TSET && G2::CNT == 21, "terminal pose row group");
    const bool term = (k == N - 1);
    if (term) {
        PGP a = pg + P_ASET + 45 * C.n;
        BMPC_UNROLL
        for (int rr = 0; rr < 15; rr++) {
            double an[3] = {a[rr], a[rr + 15], a[rr + 30]};
            double bn = pg[P_BSET + rr * 4 + C.n];
            if (!(an[0] == 0 && an[1] == 0 && an[2] == 0 && bn + pg[P_SLACKS0 + 5] > 0)) {
                double a1 = dot3(an, C.bp1), a2 = dot3(an, C.bp2);
                double bnew = bn - dot3(an, C.pend);
                double a3[3];
                BMPC_UNROLL
                for (int c = 0; c < 3; c++) {
                    double tt = 0;
                    BMPC_UNROLL
                    for (int a_ = 0; a_ < 3; a_++) tt += (a1 * C.bp1[a_] + a2 * C.bp2[a_]) * C.Dep[a_][c];
                    a3[c] = tt;
                }
                v.template add<3, 3>(a3, a1 * C.tz[0] + a2 * C.tz[1] - bnew - C.sl[5], g2.t[rr], g2.z[rr]);
            }
        }
        BMPC_UNROLL
        for (int m = 0; m < 3; m++) {
            double al[6];
            BMPC_UNROLL
            for (int c = 0; c < 6; c++) al[c] = -C.gsn[m][c];
            v.template add<6, 3>(C.gsn[m], C.projn[m] - C.ubn[m] - C.sl[5], g2.t[S_TROTU - S_TSET + m], g2.z[S_TROTU - S_TSET + m]);
            v.template add<6, 3>(al, -(C.projn[m] - C.lbn[m] + C.sl[5]), g2.t[S_TROTL - S_TSET + m], g2.z[S_TROTL - S_TSET + m]);
        }
    }
}

// natural-diagonal rows of DG position I (dg_pos order); calls v.diag(coef, h, t, z) per active row; row data and box
// bounds from the preloaded groups rp (slots) and bp (box positions)
template <int I, class V, class RP, class BP>
BMPC_INL void walk_diag_pos(const RP& rp, const BP& bp, int k, const double* y, V& v) {
    if constexpr (I < 28) {
        constexpr int blk = I / 7, jj = I % 7;
        constexpr int pos = (blk == 0 ? Z_Q : blk == 1 ? Z_DQ : blk == 2 ? Z_DDQ : Z_U) + jj;
        static_assert(I >= BP::I0 && I < BP::I0 + BP::CNT && 2 * I >= RP::S0 && 2 * I + 1 < RP::S0 + RP::CNT, "preloaded range");
        const double ub = bp.ub[I - BP::I0], lb = bp.lb[I - BP::I0];
        if (ub < BIGB) v.diag(1.0, y[pos] - ub, rp.t[2 * I - RP::S0], rp.z[2 * I - RP::S0]);
        if (lb > -BIGB) v.diag(-1.0, lb - y[pos], rp.t[2 * I + 1 - RP::S0], rp.z[2 * I + 1 - RP::S0]);
    } else if constexpr (I < 32) {
        constexpr int pos = (I == 28 ? Z_RS : I == 29 ? Z_DRS : I == 30 ? Z_PS : Z_DPS);
        constexpr int s = S_NONNEG + (I == 28 ? 0 : I == 29 ? 1 : I == 30 ? 2 : 3);
        static_assert(s >= RP::S0 && s < RP::S0 + RP::CNT, "preloaded range");
        v.diag(-1.0, -y[pos], rp.t[s - RP::S0], rp.z[s - RP::S0]);
    } else if constexpr (I < 38) {
        constexpr int s = S_D1 + (I - 32);
        static_assert(s >= RP::S0 && s < RP::S0 + RP::CNT, "preloaded range");
        if (k == 1) v.diag(-1.0, -y[Z_D + I - 32], rp.t[s - RP::S0], rp.z[s - RP::S0]);
    }
}

struct DiagAsm {
    RowAcc* R;
    double D, g0, g1, gz;
    BMPC_INL void begin() { D = 0; g0 = 0; g1 = 0; gz = 0; }
    BMPC_INL void diag(double coef, double h, double t, double z) {
        double sg, r0, r1, zz;
        R->row_tz(t, z, h, sg, r0, r1, zz);
        D += sg; g0 += coef * r0; g1 += coef * r1; gz += coef * zz;
    }
};

// column j of Op = d(pose)/d(q, dq, pi) and Ov = d(v)/d(q, dq, pi)
template <int J>
BMPC_INL void chain_cols(const KinT& K, const double Jl[3][7], const double G[6][7], double hdt, double* cO, double* cV) {
    if constexpr (J < 7) {
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) { cO[a] = Jl[a][J]; cO[3 + a] = hdt * G[3 + a][J]; cV[a] = G[a][J]; cV[3 + a] = G[3 + a][J]; }
    } else if constexpr (J < 14) {
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) { cO[a] = 0; cO[3 + a] = hdt * K.zx[J - 7][a]; cV[a] = Jl[a][J - 7]; cV[3 + a] = K.zx[J - 7][a]; }
    } else {
        BMPC_UNROLL
        for (int a = 0; a < 6; a++) { cO[a] = (a == 3 + (J - 14)) ? 1.0 : 0.0; cV[a] = 0; }
    }
}

struct ChainOut { double g17[3][17]; };

template <int J, int I, class EM>
BMPC_INL void chain_rows(const KinT& K, const double Jl[3][7], const double G[6][7], double hdt, const double* t1,
                         const double* t2, const PointRes& PR, EM& E) {
    if constexpr (I <= J) {
        double cO[6], cV[6];
        chain_cols<I>(K, Jl, G, hdt, cO, cV);
        double s = 0;
        BMPC_UNROLL
        for (int a = 0; a < 6; a++) s += cO[a] * t1[a] + cV[a] * t2[a];
        if constexpr (J < 7) s += PR.hqq(sym7(I, J));
        E.put(s);
        chain_rows<J, I + 1>(K, Jl, G, hdt, t1, t2, PR, E);
    }
}
template <int J, class EM>
BMPC_INL void chain_column(const KinT& K, const double Jl[3][7], const double G[6][7], double hdt, const double* M6,
                           const double* Hv, const PointRes& PR, EM& E) {
    double cO[6], cV[6], t1[6], t2[6];
    chain_cols<J>(K, Jl, G, hdt, cO, cV);
    BMPC_UNROLL
    for (int a = 0; a < 6; a++) {
        double s1 = 0, s2 = 0;
        BMPC_UNROLL
        for (int b = 0; b < 6; b++) { s1 += M6[sym6(a, b)] * cO[b]; s2 += Hv[sym6(a, b)] * cV[b]; }
        t1[a] = s1; t2[a] = s2;
    }
    chain_rows<J, 0>(K, Jl, G, hdt, t1, t2, PR, E);
}
template <int J, class EM>
BMPC_INL void chain_all(const KinT& K, const double Jl[3][7], const double G[6][7], double hdt, const double* M6,
                        const double* Hv, const PointRes& PR, EM& E) {
    if constexpr (J < 17) {
        chain_column<J>(K, Jl, G, hdt, M6, Hv, PR, E);
        chain_all<J + 1>(K, Jl, G, hdt, M6, Hv, PR, E);
    }
}
// P17 blocks: for position I of the (q, dq, pi) block emit the three slack-column couplings, then
// D, g0, g1, gz of that position (its diagonal rows are walked here), 7 fields
template <int I, int IEND, class RP, class BP, class EM>
BMPC_INL void p17_emit_all(const PipeArgs& A, PGP pg, const RP& rp, const BP& bp, int k, const double* y, const KinT& K,
                           const double Jl[3][7], const double G[6][7], double hdt, RowAcc& R, const PointRes& PA,
                           const PoseAsm& P, const double* bv, EM& E) {
    if constexpr (I < IEND) {
        PGP wts = pg + P_W;
        double cO[6], cV[6];
        chain_cols<I>(K, Jl, G, hdt, cO, cV);
        BMPC_UNROLL
        for (int sl = 0; sl < 3; sl++) {
            double s = 0;
            BMPC_UNROLL
            for (int a = 0; a < 6; a++) s += cO[a] * P.mS[sl][a];
            if (sl == 2 && I < 7) s += PA.cd5(I < 7 ? I : 0);
            E.put(s);
        }
        double g0 = 0, g1 = 0, gz = 0;
        BMPC_UNROLL
        for (int a = 0; a < 6; a++) {
            g0 += cO[a] * P.bp0[a] + cV[a] * bv[a];
            g1 += cO[a] * P.bp1[a];
            gz += cO[a] * P.bpz[a] + cV[a] * bv[a];
        }
        constexpr int DI = I < 14 ? I : 38 + (I - 14);      // dg index of this position
        DiagAsm dgv;
        dgv.R = &R;
        dgv.begin();
        if constexpr (DI < 38) walk_diag_pos<DI>(rp, bp, k, y, dgv);
        double D = dgv.D;
        g0 += dgv.g0; g1 += dgv.g1; gz += dgv.gz;
        if constexpr (I < 7) { g0 += PA.gq(0, I); g1 += PA.gq(1, I); gz += PA.gq(2, I); }
        if constexpr (I >= 7 + 2 && I <= 7 + 4) {            // joint-velocity cost (Q9)
            const double w2 = 2 * wts[6], val = w2 * y[Z_DQ + I - 7];
            D += w2; g0 += val; gz += val;
        }
        E.put(D); E.put(g0); E.put(g1); E.put(gz);
        p17_emit_all<I + 1, IEND>(A, pg, rp, bp, k, y, K, Jl, G, hdt, R, PA, P, bv, E);
    }
}

// second-order kinematic terms of the Lagrangian Hessian for generalised forces Fp (on p_ee),
// Fv (on v; WITHOUT the multiplier of the pi dynamics, which k_ric adds), Fc (on the 6 points)
// sa, sbq, sbdq, sc1: the chained rank-2 curvature of the sigmoid-weighted error terms, c1 a a^T + b a^T + a b^T with
// a on q only and b = (bq, bdq, bpi); its q x pi part is emitted by the caller
BMPC_INL void curvature_emit(const KinT& K, const double Jl[3][7], const double* dq, const double* Fp, const double* Fv,
                             const double Fc[6][3], const double* sa, const double* sbq, const double* sbdq, double sc1, Emitter& E) {
    const int njc[6] = {2, 3, 4, 5, 6, 4};
    // q x q block: third derivatives of the kinematics contracted with forces and dq -- symmetric in (a, bq), so the upper
    // triangle is computed and both halves are emitted from it
    double tri[28];
    BMPC_UNROLL
    for (int a = 0; a < 7; a++)
        BMPC_UNROLL
        for (int bq = a; bq < 7; bq++) {
            const int m = a < bq ? a : bq, M = a < bq ? bq : a;
            double cM[3] = {Jl[0][M], Jl[1][M], Jl[2][M]}, zc[3];
            cross3r(K.zx[m], cM, zc);
            double acc = dot3(Fp, zc);
            BMPC_UNROLL
            for (int c = 0; c < 6; c++)
                if (M < njc[c]) {
                    const double* pc = (c < 5) ? K.o[c + 2] : K.pl4;
                    double r[3] = {pc[0] - K.o[M][0], pc[1] - K.o[M][1], pc[2] - K.o[M][2]}, cc[3];
                    cross3r(K.zx[M], r, cc);
                    cross3r(K.zx[m], cc, zc);
                    acc += dot3(Fc[c], zc);
                }
            BMPC_UNROLL
            for (int j = 0; j < 7; j++) {
                const int m1 = a < j ? a : j, M1 = a < j ? j : a;
                double c1[3] = {Jl[0][M1], Jl[1][M1], Jl[2][M1]};
                double t1[3] = {0, 0, 0}, t2[3], dzm[3], dcM[3];
                if (bq < m1) { cross3r(K.zx[bq], K.zx[m1], dzm); cross3r(dzm, c1, t1); }
                const int m2 = bq < M1 ? bq : M1, M2 = bq < M1 ? M1 : bq;
                double c2[3] = {Jl[0][M2], Jl[1][M2], Jl[2][M2]};
                cross3r(K.zx[m2], c2, dcM);
                cross3r(K.zx[m1], dcM, t2);
                double lin = Fv[0] * (t1[0] + t2[0]) + Fv[1] * (t1[1] + t2[1]) + Fv[2] * (t1[2] + t2[2]);
                double ang = 0;
                if (a < j) {
                    double u1[3] = {0, 0, 0}, u2[3] = {0, 0, 0}, tmp[3];
                    if (bq < a) { cross3r(K.zx[bq], K.zx[a], tmp); cross3r(tmp, K.zx[j], u1); }
                    if (bq < j) { cross3r(K.zx[bq], K.zx[j], tmp); cross3r(K.zx[a], tmp, u2); }
                    ang = Fv[3] * (u1[0] + u2[0]) + Fv[4] * (u1[1] + u2[1]) + Fv[5] * (u1[2] + u2[2]);
                }
                acc += dq[j] * (lin + ang);
            }
            tri[sym7(a, bq)] = acc + (sc1 * sa[a] * sa[bq] + sbq[a] * sa[bq] + sa[a] * sbq[bq]);
        }
    BMPC_UNROLL
    for (int a = 0; a < 7; a++)
        BMPC_UNROLL
        for (int bq = 0; bq < 7; bq++) E.put(tri[sym7(a, bq)]);
    BMPC_UNROLL
    for (int i = 0; i < 7; i++)
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) {
            const int m = i < j ? i : j, M = i < j ? j : i;
            double cM[3] = {Jl[0][M], Jl[1][M], Jl[2][M]}, zc[3];
            cross3r(K.zx[m], cM, zc);
            double acc = dot3(Fv, zc);
            if (i < j) { double zz[3]; cross3r(K.zx[i], K.zx[j], zz); acc += dot3(Fv + 3, zz); }
            E.put(acc + sa[i] * sbdq[j]);
        }
}

// DG entries: position I of the dg order
template <int I, class RP, class BP, class EM>
BMPC_INL void dg_emit_all(const PipeArgs& A, PGP pg, const RP& rp, const BP& bp, int k, bool term,
                          const double* y, RowAcc& R, const PointRes& PA, const PoseAsm& PO, EM& E) {
    if constexpr (I < 38) {
        PGP wts = pg + P_W;
        DiagAsm dgv;
        dgv.R = &R;
        dgv.begin();
        walk_diag_pos<I>(rp, bp, k, y, dgv);
        double D = dgv.D, g0 = dgv.g0, g1 = dgv.g1, gz = dgv.gz;
        // slack columns of the pose rows (ps, rs, d5) and of the point rows (d_c)
        if constexpr (I == 30) { D += PO.sS[0]; g0 += PO.bS0[0]; g1 += PO.bS1[0]; gz += PO.bSz[0]; }
        if constexpr (I == 28) { D += PO.sS[1]; g0 += PO.bS0[1]; g1 += PO.bS1[1]; gz += PO.bSz[1]; }
        if constexpr (I >= 32 && I < 38) {
            constexpr int c = I - 32;
            D += PA.dD(c); g0 += PA.gD(0, c); g1 += PA.gD(1, c); gz += PA.gD(2, c);
            if constexpr (c == 5) { D += PO.sS[2]; g0 += PO.bS0[2]; g1 += PO.bS1[2]; gz += PO.bSz[2]; }
        }
        // direct quadratic cost terms (natural coordinates)
        {
            double w2 = 0, extra = 0;
            bool has = false;
            if constexpr (I >= 21 && I < 28) { w2 = 2 * wts[7]; has = true; }
            else if constexpr (I == 28 || I == 30) { w2 = 2 * wts[9]; has = true; }
            else if constexpr (I == 29 || I == 31) { w2 = 2 * wts[10]; has = true; }
            else if constexpr (I >= 32 && I < 38) {
                constexpr int i = I - 32;
                w2 = term ? (2 * wts[10] + (i != 4 ? 2 * wts[8] : 0.0)) : 0.0;
                extra = (term && i != 4) ? 2 * wts[8] * pg[P_SLACKS0 + i] : 0.0;
                has = true;
            }
            if (has) {
                double val = w2 * y[dg_pos_c(I)] + extra;
                D += w2; g0 += val; gz += val;
            }
        }
        E.put(D); E.put(g0); E.put(g1); E.put(gz);
        dg_emit_all<I + 1>(A, pg, rp, bp, k, term, y, R, PA, PO, E);
    }
}

// ------------------------------------------------------------------------------------------
// k_pose (round 4): reference / error context, output-space cost gradient and Hessian, the pose rows (EE set, orientation bounds,
// path-parameter cap, terminal rows) -> pose-space Hessian M6, slack couplings, gradients: 84 doubles per pair in the side array,
// which k_eval chains through the kinematic columns.  Split off k_eval because the context (~150 doubles) and the kinematic columns
// (111) together with the accumulators do not fit one wavefront's register file; neither kernel needs the other's big state.
// lds: staged parameter vectors only.
// ------------------------------------------------------------------------------------------
BMPC_KBODY void k_pose_body(const PipeArgs& A, int wave, int lane, LDSD* lds_par) {
    const int count = A.L.cnt[0], N = A.N;
    if (wave * ipw_of(N) >= count) return;
    PairMap m = pair_map(A, A.L.eval, count, wave, lane);
    const int k = m.k, n_w = 44 * N + 6;
    const bool term = (k == N - 1);
    const DynC dc = make_dync(A.o.dt);
    GCD lbx = A.lbx + (size_t)A.src[m.b] * n_w;
    PGP pg = stage_params(A, A.L.eval, count, wave, lane, m, lds_par);
    double iw0[3];
    BMPC_UNROLL
    for (int c = 0; c < 3; c++) iw0[c] = lbx[28 * N + (3 + c) * N];
    StagePoint S;
    const int flip = A.st[m.b].flip;
    GCD zc = cur_zeta(A, flip), tc = cur_t(A, flip), zcur = cur_z(A, flip);
    load_zeta(zc, A.NP, m.pi, S.zeta);
    // row data of the pose rows: in flight while the kinematics are evaluated
    RowPre<S_EE, 21> rp_pose;
    rp_pose.load(A, tc, zcur, m.pi);
    const double t_phi = tc[(size_t)S_PHI * A.NP + m.pi], z_phi = zcur[(size_t)S_PHI * A.NP + m.pi];
    stage_point(A, pg, iw0, k, dc, S);
    double g12[12], Hp[21];
    cost_grad12(pg, S.C, term, g12);
    {
        double HvX[21];
        cost_hess(pg, S.C, term, Hp, HvX);
    }
    RowAcc R;
    R.init(&A, m.pi, m.valid, 0.0, tc, zcur);
    // (KKT partial sums of the pose rows only: k_points runs beside this kernel, k_eval combines the two)
    PoseAsm PO;
    PO.R = &R;
    PO.init(Hp, g12);
    walk_pose_rows(pg, N, k, S.y, S.C, PO, rp_pose, t_phi, z_phi);
    {
        RowPre<S_TSET, 21> rp_term;          // one batch of loads per row group, right before the group is walked
        rp_term.load(A, tc, zcur, m.pi);
        walk_pose_rows_term(pg, N, k, S.y, S.C, PO, rp_term);
    }
    if (!m.valid) return;
    const double hdt = 0.5 * dc.dt;
    // generalised forces for the second-order kinematic terms (k_curv)
    if (A.o.hess == 2) {
        GD F = A.part + (size_t)PT_FORCE * A.NP + m.pi;
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) F[(size_t)a * A.NP] = PO.bpz[a];
        BMPC_UNROLL
        for (int a = 0; a < 6; a++) F[(size_t)(3 + a) * A.NP] = g12[6 + a] + (a >= 3 ? hdt * PO.bpz[a] : 0.0);
        // what the Gauss-Newton model of r = sig(phi) e leaves out of the Hessian of sig^2 (|e_r|^2 + |e_p|^2)
        // (casadi_ocp_formulation.py:272-276): 2 sig [ sig'' |e|^2 dphi dphi^T + sig' (ge dphi^T + dphi ge^T) ], ge = De^T e
        GD Sg = A.part + (size_t)PT_SIG * A.NP + m.pi;
        const double c2 = 2 * S.C.sig * S.C.dsig;
        Sg[0] = 2 * S.C.sig * (60.0 * S.C.dsig * (1.0 - 2.0 * S.C.sig)) * S.C.er2ep2;
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) Sg[(size_t)(1 + a) * A.NP] = S.C.dpp[a];
        BMPC_UNROLL
        for (int bb = 0; bb < 6; bb++) {
            double ge = S.C.Der[0][bb] * S.C.er[0] + S.C.Der[1][bb] * S.C.er[1] + S.C.Der[2][bb] * S.C.er[2];
            if (bb < 3) ge += S.C.Dep[0][bb] * S.C.ep[0] + S.C.Dep[1][bb] * S.C.ep[1] + S.C.Dep[2][bb] * S.C.ep[2];
            Sg[(size_t)(4 + bb) * A.NP] = c2 * ge;
        }
    }
    GD Pz = A.part + (size_t)PT_POSE * A.NP + m.pi;
    BMPC_UNROLL
    for (int i = 0; i < 21; i++) Pz[(size_t)(PZ_M6 + i) * A.NP] = PO.M6[i];
    BMPC_UNROLL
    for (int sl = 0; sl < 3; sl++) {
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) Pz[(size_t)(PZ_MS + 6 * sl + i) * A.NP] = PO.mS[sl][i];
        Pz[(size_t)(PZ_SS + sl) * A.NP] = PO.sS[sl];
        Pz[(size_t)(PZ_BS0 + sl) * A.NP] = PO.bS0[sl]; Pz[(size_t)(PZ_BS1 + sl) * A.NP] = PO.bS1[sl]; Pz[(size_t)(PZ_BSZ + sl) * A.NP] = PO.bSz[sl];
    }
    BMPC_UNROLL
    for (int i = 0; i < 6; i++) {
        Pz[(size_t)(PZ_BP0 + i) * A.NP] = PO.bp0[i]; Pz[(size_t)(PZ_BP1 + i) * A.NP] = PO.bp1[i]; Pz[(size_t)(PZ_BPZ + i) * A.NP] = PO.bpz[i];
        Pz[(size_t)(PZ_BV + i) * A.NP] = g12[6 + i];
    }
    Pz[(size_t)(PZ_KKT + 0) * A.NP] = R.cmax; Pz[(size_t)(PZ_KKT + 1) * A.NP] = R.csum; Pz[(size_t)(PZ_KKT + 2) * A.NP] = R.cmin;
    Pz[(size_t)(PZ_KKT + 3) * A.NP] = R.zsum; Pz[(size_t)(PZ_KKT + 4) * A.NP] = R.prim; Pz[(size_t)(PZ_KKT + 5) * A.NP] = R.nrows;
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) Pz[(size_t)(PZ_VANG + a) * A.NP] = S.C.v[3 + a];
}

// lds: EM_DOUBLES doubles per wave
// MODE 0: the whole record (A/B runs: BMPC_EVAL_SPLIT_WGS=0).  The product runs two wavefronts side by side -- in the tail regime a
// launch costs its single-thread latency, in the bulk regime the two lighter bodies spill less --: MODE 1 everything but the chained (q, dq, pi) block -- a third of the arithmetic, with no row of its own (no KKT partial
// sum depends on it) --, MODE 2 that block alone, straight into the record.  Same expressions entry for entry, and the library is
// built with -ffp-contract=on (contraction per source expression, whatever else the surrounding body computes): bitwise the record of MODE 0.
template <int MODE>
BMPC_KBODY void k_eval_body(const PipeArgs& A, int wave, int lane, LDSD* lds) {
    const int count = A.L.cnt[0], N = A.N;
    if (wave * ipw_of(N) >= count) return;
    PairMap m = pair_map(A, A.L.eval, count, wave, lane);
    const int k = m.k, n_w = 44 * N + 6;
    const bool term = (k == N - 1);
    const DynC dc = make_dync(A.o.dt);
    GCD lbx = A.lbx + (size_t)A.src[m.b] * n_w;
    GCD ubx = A.ubx + (size_t)A.src[m.b] * n_w;
    PGP pg = stage_params(A, A.L.eval, count, wave, lane, m, lds + EM_DOUBLES + 8);
    if constexpr (MODE == 2) {
        struct { double zeta[NZ], y[NZ]; KinT K; double Jl[3][7]; } S;
        const int flip = A.st[m.b].flip;
        load_zeta(cur_zeta(A, flip), A.NP, m.pi, S.zeta);
        nat_all(S.zeta, dc, S.y);
        kin_chain(A.rc, S.y + Z_Q, S.K);
        kin_jlin(S.K, S.Jl);
        double G[6][7];
        kin_G(S.K, S.Jl, S.y + Z_DQ, G);
        PointRes PA;
        PA.base = A.part + (size_t)PT_SIDE * A.NP + m.pi; PA.NP = A.NP;
        GCD Pz = A.part + (size_t)PT_POSE * A.NP + m.pi;
        double Hv[21], M6[21];
        cost_hess_v(pg, N, k, Hv);
        BMPC_UNROLL
        for (int i = 0; i < 21; i++) M6[i] = Pz[(size_t)(PZ_M6 + i) * A.NP];
        PA.load_hqq();
        DirectEmitter D;
        D.out = A.hrec + hrec_of(A, m.b, m.k) + F_H17; D.valid = m.valid; D.f = 0;
        chain_all<0>(S.K, S.Jl, G, 0.5 * dc.dt, M6, Hv, PA, D);
        return;
    }
    const double ad = A.st[m.b].ad;
    EmitterT<MODE == 1> E;
    E.init(lds, A.hrec, lane, hrec_of(A, m.b, m.k), m.valid, 0, F_H17, F_EW);
    BMPC_SYNC();
    // kinematic columns at the iterate (the reference / error context and the pose rows are k_pose's: their results come from the side array)
    struct { double zeta[NZ], y[NZ]; KinT K; double Jl[3][7]; } S;
    const int flip = A.st[m.b].flip;
    GCD zc = cur_zeta(A, flip), tc = cur_t(A, flip), zcur = cur_z(A, flip);
    load_zeta(zc, A.NP, m.pi, S.zeta);
    nat_all(S.zeta, dc, S.y);
    kin_chain(A.rc, S.y + Z_Q, S.K);
    kin_jlin(S.K, S.Jl);
    double G[6][7];
    kin_G(S.K, S.Jl, S.y + Z_DQ, G);
    RowAcc R;
    R.init(&A, m.pi, m.valid, ad, tc, zcur);
    // ---- results of k_pose (pose-space Hessian, slack couplings, gradients): loaded phase by phase, right before their use ----
    PoseAsm PO;
    PO.R = &R;
    GCD Pz = A.part + (size_t)PT_POSE * A.NP + m.pi;
    {   // KKT partial sums so far: collision-point rows (k_points) and pose rows (k_pose)
        GCD Sd = A.part + (size_t)(PT_SIDE + SD_KKT) * A.NP + m.pi;
        R.cmax = fmax(Sd[0], Pz[(size_t)(PZ_KKT + 0) * A.NP]); R.csum = Sd[A.NP] + Pz[(size_t)(PZ_KKT + 1) * A.NP];
        R.cmin = fmin(Sd[2 * A.NP], Pz[(size_t)(PZ_KKT + 2) * A.NP]); R.zsum = Sd[3 * A.NP] + Pz[(size_t)(PZ_KKT + 3) * A.NP];
        R.prim = fmax(Sd[4 * A.NP], Pz[(size_t)(PZ_KKT + 4) * A.NP]); R.nrows = Sd[5 * A.NP] + Pz[(size_t)(PZ_KKT + 5) * A.NP];
    }
    // ---- collision-point results of k_points: q x d columns first, the rest where it is needed ----
    PointRes PA;
    PA.base = A.part + (size_t)PT_SIDE * A.NP + m.pi; PA.NP = A.NP;
    {
        double cdv[35];
        BMPC_UNROLL
        for (int i = 0; i < 35; i++) cdv[i] = PA.at(SD_CD + i);
        BMPC_UNROLL
        for (int i = 0; i < 35; i++) E.put(cdv[i]);
    }
    const double hdt = 0.5 * dc.dt;
    // ---- slack-column couplings + gradients + diagonal rows of the 17 chained positions ----
    {
        double bvv[6];
        BMPC_UNROLL
        for (int sl = 0; sl < 3; sl++)
            BMPC_UNROLL
            for (int i = 0; i < 6; i++) PO.mS[sl][i] = Pz[(size_t)(PZ_MS + 6 * sl + i) * A.NP];
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) {
            PO.bp0[i] = Pz[(size_t)(PZ_BP0 + i) * A.NP]; PO.bp1[i] = Pz[(size_t)(PZ_BP1 + i) * A.NP]; PO.bpz[i] = Pz[(size_t)(PZ_BPZ + i) * A.NP];
            bvv[i] = Pz[(size_t)(PZ_BV + i) * A.NP];
        }
        PA.load_p17();
        {
            RowPre<0, 14> rp_a;              // rows + bounds of the q box
            BndPre<0, 7> bp_a;
            rp_a.load(A, tc, zcur, m.pi); bp_a.load(lbx, ubx, N, k);
            p17_emit_all<0, 7>(A, pg, rp_a, bp_a, k, S.y, S.K, S.Jl, G, hdt, R, PA, PO, bvv, E);
        }
            {
            RowPre<14, 14> rp_a;             // ... of the dq box (the pi positions have no rows)
            BndPre<7, 7> bp_a;
            rp_a.load(A, tc, zcur, m.pi); bp_a.load(lbx, ubx, N, k);
            p17_emit_all<7, 17>(A, pg, rp_a, bp_a, k, S.y, S.K, S.Jl, G, hdt, R, PA, PO, bvv, E);
        }
    }
    // ---- chained (q, dq, pi) block ----
    if constexpr (MODE == 1) {
        BMPC_UNROLL
        for (int i = 0; i < F_EW - F_H17; i++) E.put(0.0);      // (the chain part's fields: pass through the tile, not stored)
    } else {
        double Hv[21];
        cost_hess_v(pg, N, k, Hv);
        BMPC_UNROLL
        for (int i = 0; i < 21; i++) PO.M6[i] = Pz[(size_t)(PZ_M6 + i) * A.NP];
        PA.load_hqq();
        // register-resident copies of the kinematic columns for this block (they are read ~20 times each here; the originals
        // were parked in scratch while the pose / diagonal rows needed the registers)
        BMPC_UNROLL
        for (int a = 0; a < 3; a++)
            BMPC_UNROLL
            for (int j = 0; j < 7; j++) { BMPC_PIN(S.Jl[a][j]); BMPC_PIN(S.K.zx[j][a]); }
        BMPC_UNROLL
        for (int a = 0; a < 6; a++)
            BMPC_UNROLL
            for (int j = 0; j < 7; j++) BMPC_PIN(G[a][j]);
        BMPC_UNROLL
        for (int i = 0; i < 21; i++) BMPC_PIN(PO.M6[i]);
        chain_all<0>(S.K, S.Jl, G, hdt, PO.M6, Hv, PA, E);
    }
    // ---- dynamics linearisation data (angular Jacobian, joint axes, suffix sums of the angular velocity): emitted here, while the
    // kinematic columns are still in registers; nothing below needs them ----
    BMPC_UNROLL
    for (int a = 0; a < 3; a++)
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) E.put(G[3 + a][j]);
    BMPC_UNROLL
    for (int a = 0; a < 3; a++)
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) E.put(S.K.zx[j][a]);
    {
        double sz[3] = {0, 0, 0}, sufz[7][3];
        BMPC_UNROLL
        for (int j = 6; j >= 0; j--)
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) { sz[a] += S.K.zx[j][a] * S.y[Z_DQ + j]; sufz[j][a] = sz[a]; }
        // sufz[m] for m = 1..7 (sufz[7] = 0)
        BMPC_UNROLL
        for (int mm = 1; mm < 8; mm++)
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) E.put(mm < 7 ? sufz[mm < 7 ? mm : 0][a] : 0.0);
    }
    // The iterate is read again (and its natural form recomputed) for the rest of the kernel instead of being carried
    // through the chained block: there the register file is needed for the kinematic columns (K, Jl, G), which the register
    // allocator otherwise parks in scratch memory and reloads for every entry of the block
    double zeta2[NZ], y2[NZ];
    load_zeta(zc, A.NP, m.pi, zeta2);
    nat_all(zeta2, dc, y2);
    // ---- remaining diagonal rows + gradients: ddq, u, rs, drs, ps, dps, d ----
    RowPre<28, 40> rp_b;                     // rows + bounds of the ddq, u box, the slack rows, the zeta-diagonal rows
    BndPre<14, 14> bp_b;
    rp_b.load(A, tc, zcur, m.pi); bp_b.load(lbx, ubx, N, k);
    BMPC_UNROLL
    for (int sl = 0; sl < 3; sl++) {
        PO.sS[sl] = Pz[(size_t)(PZ_SS + sl) * A.NP];
        PO.bS0[sl] = Pz[(size_t)(PZ_BS0 + sl) * A.NP]; PO.bS1[sl] = Pz[(size_t)(PZ_BS1 + sl) * A.NP]; PO.bSz[sl] = Pz[(size_t)(PZ_BSZ + sl) * A.NP];
    }
    PA.load_dg();
    dg_emit_all<14>(A, pg, rp_b, bp_b, k, term, y2, R, PA, PO, E);
    // ---- zeta-diagonal rows (k == 1) ----
    {
        double sg2[2] = {0, 0}, r2[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        if (k == 1) {
            BMPC_UNROLL
            for (int i = 0; i < 2; i++) {
                double sg, r0, r1, zz;
                R.row_tz(rp_b.t[S_RS1 - 28 + i], rp_b.z[S_RS1 - 28 + i], -zeta2[i ? Z_PS : Z_RS], sg, r0, r1, zz);
                sg2[i] = sg; r2[0][i] = r0; r2[1][i] = r1; r2[2][i] = zz;
            }
        }
        E.put(sg2[0]); E.put(sg2[1]);
        BMPC_UNROLL
        for (int v = 0; v < 3; v++) { E.put(r2[v][0]); E.put(r2[v][1]); }
    }
    double prim = R.prim;
    {
        double rdef[NX];
        if (!term) {
            double zn[NX];
            BMPC_UNROLL
            for (int i = 0; i < NX; i++) zn[i] = zc[(size_t)i * A.NP + m.pi + 1];
            double vang[3];
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) vang[a] = Pz[(size_t)(PZ_VANG + a) * A.NP];
            defect_all(zeta2, zn, vang, dc, rdef);
            BMPC_UNROLL
            for (int i = 0; i < NX; i++) prim = fmax(prim, fabs(rdef[i]));
        } else {
            BMPC_UNROLL
            for (int i = 0; i < NX; i++) rdef[i] = 0;
        }
        BMPC_UNROLL
        for (int i = 0; i < NX; i++) E.put(rdef[i]);
    }
    E.pad_to(512);
    if (k == 1) {
        double x1fix[24];
        x1fix_eval(lbx, N, dc.dt, x1fix);
        BMPC_UNROLL
        for (int i = 0; i < 24; i++) prim = fmax(prim, fabs(x1fix[i] - zeta2[i]));
    }
    if (m.valid) {
        GD P = A.part + m.pi;
        P[PT_CMAX * A.NP] = R.cmax; P[PT_CSUM * A.NP] = R.csum; P[PT_CMIN * A.NP] = R.cmin; P[PT_ZSUM * A.NP] = R.zsum;
        P[PT_PRIM * A.NP] = prim; P[PT_NROWS * A.NP] = R.nrows;
    }
}

// ------------------------------------------------------------------------------------------
// k_points: the 90 collision-point rows of a pair (ocp :323-330): barrier terms chained through the point
// Jacobians into the q x q block, the q x d columns and the gradients; results go to the side array.
// A kernel of its own because together with the pose / chain work of k_eval it does not fit the
// register file (the spills were what bound k_eval).
// ------------------------------------------------------------------------------------------
BMPC_KBODY void k_points_body(const PipeArgs& A, int wave, int lane, LDSD* lds_par) {
    const int count = A.L.cnt[0], N = A.N;
    if (wave * ipw_of(N) >= count) return;
    PairMap m = pair_map(A, A.L.eval, count, wave, lane);
    // instances in hess_mode get their second-order terms from k_curv: list them (any order: per-pair arithmetic only)
    if (m.valid && m.k == 1 && A.o.hess == 2 && A.st[m.b].hess_mode) {
        int pos = BMPC_ATOMIC_INC(A.L.cnt + 10);
        A.L.curv[pos] = m.b;
    }
    const DynC dc = make_dync(A.o.dt);
    PGP pg = stage_params(A, A.L.eval, count, wave, lane, m, lds_par);
    double zeta[NZ], y[NZ];
    const int flip = A.st[m.b].flip;
    load_zeta(cur_zeta(A, flip), A.NP, m.pi, zeta);
    nat_all(zeta, dc, y);
    KinT K;
    kin_chain(A.rc, y + Z_Q, K);
    SegCtx C;
    BMPC_UNROLL
    for (int i = 0; i < 6; i++) C.sl[i] = pg[P_SLACKS0 + i] + y[Z_D + i];
    RowAcc R;
    R.init(&A, m.pi, m.valid, 0.0, cur_t(A, flip), cur_z(A, flip));
    PointAsm PA;
    PA.R = &R; PA.K = &K;
    PA.init();
    walk_points<PointAsm, 0>(pg, K, C.sl, PA);
    if (!m.valid) return;
    GD Sd = A.part + (size_t)PT_SIDE * A.NP + m.pi;
    BMPC_UNROLL
    for (int c = 0; c < 5; c++)
        BMPC_UNROLL
        for (int i = 0; i < 7; i++) Sd[(size_t)(SD_CD + 7 * c + i) * A.NP] = PA.cd[c][i];
    BMPC_UNROLL
    for (int i = 0; i < 7; i++) {
        Sd[(size_t)(SD_CD5 + i) * A.NP] = PA.cd[5][i];
        Sd[(size_t)(SD_GQ + i) * A.NP] = PA.gq0[i]; Sd[(size_t)(SD_GQ + 7 + i) * A.NP] = PA.gq1[i]; Sd[(size_t)(SD_GQ + 14 + i) * A.NP] = PA.gqz[i];
    }
    BMPC_UNROLL
    for (int i = 0; i < 28; i++) Sd[(size_t)(SD_HQQ + i) * A.NP] = PA.Hqq[i];
    BMPC_UNROLL
    for (int c = 0; c < 6; c++) {
        Sd[(size_t)(SD_DD + c) * A.NP] = PA.dD[c];
        Sd[(size_t)(SD_GD + c) * A.NP] = PA.gD0[c]; Sd[(size_t)(SD_GD + 6 + c) * A.NP] = PA.gD1[c]; Sd[(size_t)(SD_GD + 12 + c) * A.NP] = PA.gDz[c];
    }
    Sd[(size_t)(SD_KKT + 0) * A.NP] = R.cmax; Sd[(size_t)(SD_KKT + 1) * A.NP] = R.csum; Sd[(size_t)(SD_KKT + 2) * A.NP] = R.cmin;
    Sd[(size_t)(SD_KKT + 3) * A.NP] = R.zsum; Sd[(size_t)(SD_KKT + 4) * A.NP] = R.prim; Sd[(size_t)(SD_KKT + 5) * A.NP] = R.nrows;
    if (A.o.hess == 2) {
        GD F = A.part + (size_t)PT_FORCE * A.NP + m.pi;
        BMPC_UNROLL
        for (int c = 0; c < 6; c++)
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) F[(size_t)(9 + 3 * c + a) * A.NP] = PA.Fc[c][a];
    }
}

// ------------------------------------------------------------------------------------------
// k_curv: second-order kinematic terms of the Lagrangian Hessian (record fields F_CQQ, F_CQD) for the
// instances in hess_mode (the list k_points built: wavefronts made of such instances only); forces from k_eval.
// ------------------------------------------------------------------------------------------
BMPC_KBODY void k_curv_body(const PipeArgs& A, int wave, int lane, LDSD* lds) {
    const int count = A.L.cnt[10], N = A.N;
    if (A.o.hess != 2 || wave * ipw_of(N) >= count) return;
    PairMap m = pair_map(A, A.L.curv, count, wave, lane);
    const DynC dc = make_dync(A.o.dt);
    Emitter E;
    E.init(lds, A.hrec, lane, hrec_of(A, m.b, m.k), m.valid, F_CQP);
    BMPC_SYNC();
    double zeta[NZ], y[NZ];
    load_zeta(cur_zeta(A, A.st[m.b].flip), A.NP, m.pi, zeta);
    nat_all(zeta, dc, y);
    KinT K;
    double Jl[3][7];
    kin_chain(A.rc, y + Z_Q, K);
    kin_jlin(K, Jl);
    double Fp[3], Fv[6], Fc[6][3];
    GCD F = A.part + (size_t)PT_FORCE * A.NP + m.pi;
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) Fp[a] = F[(size_t)a * A.NP];
    BMPC_UNROLL
    for (int a = 0; a < 6; a++) Fv[a] = F[(size_t)(3 + a) * A.NP];
    BMPC_UNROLL
    for (int c = 0; c < 6; c++)
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) Fc[c][a] = F[(size_t)(9 + 3 * c + a) * A.NP];
    // exact curvature of the sigmoid-weighted error terms (scalars from k_eval), chained through d(pose)/d(q, dq, pi)
    double sa[7], sbq[7], sbdq[7], sc1;
    {
        GCD Sg = A.part + (size_t)PT_SIG * A.NP + m.pi;
        double dpp[3], gec[6];
        sc1 = Sg[0];
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) dpp[a] = Sg[(size_t)(1 + a) * A.NP];
        BMPC_UNROLL
        for (int a = 0; a < 6; a++) gec[a] = Sg[(size_t)(4 + a) * A.NP];
        double G[6][7];
        kin_G(K, Jl, y + Z_DQ, G);
        const double hdt = 0.5 * dc.dt;
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) {
            sa[j] = dpp[0] * Jl[0][j] + dpp[1] * Jl[1][j] + dpp[2] * Jl[2][j];
            sbq[j] = gec[0] * Jl[0][j] + gec[1] * Jl[1][j] + gec[2] * Jl[2][j]
                     + hdt * (gec[3] * G[3][j] + gec[4] * G[4][j] + gec[5] * G[5][j]);
            sbdq[j] = hdt * (gec[3] * K.zx[j][0] + gec[4] * K.zx[j][1] + gec[5] * K.zx[j][2]);
        }
        BMPC_UNROLL
        for (int i = 0; i < 7; i++)
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) E.put(sa[i] * gec[3 + a]);
    }
    curvature_emit(K, Jl, y + Z_DQ, Fp, Fv, Fc, sa, sbq, sbdq, sc1, E);
    E.pad_to(HREC);
}

// line-search start of one instance (fraction-to-boundary step lengths, merit derivative) from the per-pair partials of
// k_step, summed in pair order (fixed order -> reproducible)
BMPC_INL void ls0_instance(const PipeArgs& A, int b) {
    const int N = A.N;
    GST st = A.st + b;
    GCD P = A.part + pair_of(A, b, 1);
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
    // the search starts at ls_alpha_mem times the step length the previous iteration ended with (still in st->alpha)
    st->alpha = (A.o.ls_alpha_mem > 0 && st->it > 0) ? fmin(ap, A.o.ls_alpha_mem * st->alpha) : ap;
    st->bt = 0; st->armijo = 0;
    st->state = ST_TRIAL;
    int pos = BMPC_ATOMIC_INC(A.L.cnt + 2);
    A.L.trial[pos] = b;
}

// filter acceptance test of one instance's trial point (Waechter & Biegler 2006, Sec. 2.3) from the per-pair partials of
// k_trial; returns true when the trial becomes the iterate.  A rejected instance goes to the next super-step's trial list
// only when k_trial does not try again itself (`requeue`)
// (f1, th1, ls1: objective, infeasibility and sum log t of the trial point with step length st->alpha)
BMPC_INL bool ls_decide(const PipeArgs& A, int b, double f1, double th1, double ls1, bool requeue) {
    GST st = A.st + b;
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
        st->f0 = f1; st->th0 = th1; st->ls0 = ls1;      // merit pieces of the accepted point
        if (!acc) st->alpha = 1e300;                     // no acceptable step found (the last trial is kept): no step-length memory
        st->flip ^= 1;                                   // the trial copy of t / zeta is the iterate now (cur_t, cur_zeta)
        st->it += 1;
        // the exact Hessian close to a solution, or when the Gauss-Newton model has stopped making progress
        int want = (A.o.hess == 2 && (st->err_prev < A.o.hess_switch || (A.o.inertia == 2 && st->stall >= A.o.stall_n))) ? 1 : 0;
        if (A.o.gn_backoff > 0 && want && st->gn_skip > 0) { want = 0; st->gn_skip -= 1; }
        st->hess_mode = want;
        st->state = ST_EVAL;
        int pos = BMPC_ATOMIC_INC(A.L.cnt + 3);
        A.L.eval_next[pos] = b;
        return true;
    }
    st->alpha = 0.5 * alpha; st->bt += 1;
    if (requeue) {
        int pos = BMPC_ATOMIC_INC(A.L.cnt + 4);
        A.L.trial_next[pos] = b;
    }
    return false;
}
BMPC_INL bool ls_instance(const PipeArgs& A, int b, bool requeue) {      // the trial point's pieces from the per-pair partials, in pair order
    GCD P = A.part + pair_of(A, b, 1);
    double f1 = 0, th1 = 0, ls1 = 0;
    for (int k = 0; k < A.N - 1; k++) { f1 += P[PT_F1 * A.NP + k]; th1 += P[PT_TH1 * A.NP + k]; ls1 += P[PT_LS1 * A.NP + k]; }
    return ls_decide(A, b, f1, th1, ls1, requeue);
}

// ------------------------------------------------------------------------------------------
// k_step: row steps for the Newton direction dz
// ------------------------------------------------------------------------------------------
struct StepVisitor {
    const PipeArgs* A; size_t pi; bool valid;
    double mu, tau;
    const double* dy;     // natural step
    const double* dzt;    // zeta step
    double dloc[6], dpt[6][3];
    GCD tc, zc;                                      // cur_t, cur_z of the instance
    double gt[ROW_GROUP_MAX], gz[ROW_GROUP_MAX];     // slack t and multiplier z of the current row group, loaded in one batch
    double rp, rdn, rdd, dbar;                        // max(-dt/t), max(-dz/z) as the fraction rdn / rdd (one division per pair
                                                      // instead of one per row), -mu sum dt/t over the rows of this pair
    template <int S0, int CNT> BMPC_INL void group(unsigned act) {      // bit i of act: slot S0 + i may be a constraint row (else: not loaded)
        static_assert(CNT <= ROW_GROUP_MAX, "row group size");
        BMPC_UNROLL
        for (int i = 0; i < CNT; i++) {
            size_t o = (size_t)(S0 + i) * A->NP + pi;
            if ((act >> i) & 1u) { gt[i] = tc[o]; gz[i] = zc[o]; } else { gt[i] = 1.0; gz[i] = 0.0; }
        }
    }
    BMPC_INL void fin(int s, double h, double adot) {
        // row step: t + dt = c = -h - a.d (stored in A.dt: k_trial subtracts t again), dz_row = (mu - t z - z dt) / t = (mu - z c) / t
        // (not stored: k_trial recomputes it from t, c, z where it updates the multipliers), fraction-to-boundary ratios, barrier
        // part of the merit derivative
        const double c = -h - adot, t = gt[s - row_group_base(s)], z = gz[s - row_group_base(s)];
        const double rt = BMPC_RCP(t), dti = c - t, dzi = (mu - z * c) * rt;
        if (valid) { size_t o = (size_t)s * A->NP + pi; A->dt[o] = c; }
        rp = fmax(rp, -dti * rt);
        if (-dzi * rdd > rdn * z) { rdn = -dzi; rdd = z; }          // -dzi / z > rdn / rdd  (z, rdd > 0)
        dbar -= mu * dti * rt;
    }
    BMPC_INL void skip(int) {}
    BMPC_INL void diag(int s, int pos, double coef, double h) { fin(s, h, coef * dy[pos]); }
    BMPC_INL void zdiag(int s, int pos, double coef, double h) { fin(s, h, coef * dzt[pos]); }
    BMPC_INL void pose(int s, const double* a, int sel, double h) {
        double adot = 0;
        BMPC_UNROLL
        for (int c = 0; c < 6; c++) adot += a[c] * dloc[c];
        if (sel == 1) adot -= dy[Z_PS];
        else if (sel == 2) adot -= dy[Z_RS];
        else if (sel == 3) adot -= dy[Z_D + 5];
        fin(s, h, adot);
    }
    template <int C> BMPC_INL void point_begin(unsigned act) { group<S_COL + 15 * C, 15>(act); }
    template <int C> BMPC_INL void point(int s, const double* a, double h) {
        fin(s, h, a[0] * dpt[C][0] + a[1] * dpt[C][1] + a[2] * dpt[C][2] - dy[Z_D + C]);
    }
    template <int C> BMPC_INL void point_end() {}
};

template <int C>
BMPC_INL void point_dirs(const KinT& K, const double* dyq, double dpt[6][3]) {
    if constexpr (C < 6) {
        constexpr int nj = PointNJ<C>::value;
        const double* pc = kin_point<C>(K);
        double s[3] = {0, 0, 0};
        BMPC_UNROLL
        for (int i = 0; i < nj; i++) {
            double r[3] = {pc[0] - K.o[i][0], pc[1] - K.o[i][1], pc[2] - K.o[i][2]}, c[3];
            cross3r(K.zx[i], r, c);
            s[0] += c[0] * dyq[i]; s[1] += c[1] * dyq[i]; s[2] += c[2] * dyq[i];
        }
        dpt[C][0] = s[0]; dpt[C][1] = s[1]; dpt[C][2] = s[2];
        point_dirs<C + 1>(K, dyq, dpt);
    }
}

BMPC_KBODY void k_step_body(const PipeArgs& A, int wave, int lane, LDSD* lds_par) {
    const int count = A.L.cnt[1], N = A.N;
    if (wave * ipw_of(N) >= count) return;
    PairMap m = pair_map(A, A.L.step, count, wave, lane);
    const int k = m.k, n_w = 44 * N + 6;
    const bool term = (k == N - 1);
    const DynC dc = make_dync(A.o.dt);
    GCD lbx = A.lbx + (size_t)A.src[m.b] * n_w;
    GCD ubx = A.ubx + (size_t)A.src[m.b] * n_w;
    PGP pg = stage_params(A, A.L.step, count, wave, lane, m, lds_par);
    PGP wts = pg + P_W;
    const double mu = A.st[m.b].mu;
    double iw0[3];
    BMPC_UNROLL
    for (int c = 0; c < 3; c++) iw0[c] = lbx[28 * N + (3 + c) * N];
    StagePoint S;
    const int flip = A.st[m.b].flip;
    load_zeta(cur_zeta(A, flip), A.NP, m.pi, S.zeta);
    stage_point(A, pg, iw0, k, dc, S);
    double G[6][7], g12[12];
    kin_G(S.K, S.Jl, S.y + Z_DQ, G);
    cost_grad12(pg, S.C, term, g12);
    double dzt[NZ], dy[NZ];
    load_zeta(A.dz, A.NP, m.pi, dzt);
    nat_all(dzt, dc, dy);
    StepVisitor V;
    V.A = &A; V.pi = m.pi; V.valid = m.valid; V.mu = mu; V.tau = fmax(0.99, 1.0 - mu); V.tc = cur_t(A, flip); V.zc = cur_z(A, flip);
    V.dy = dy; V.dzt = dzt; V.rp = 0.0; V.rdn = 0.0; V.rdd = 1.0; V.dbar = 0.0;
    double dv[6];
    BMPC_UNROLL
    for (int a = 0; a < 6; a++) {
        double s = 0;
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) s += G[a][j] * dy[Z_Q + j] + (a < 3 ? S.Jl[a][j] : S.K.zx[j][a - 3]) * dy[Z_DQ + j];
        dv[a] = s;
    }
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) {
        double s = 0;
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) s += S.Jl[a][j] * dy[Z_Q + j];
        V.dloc[a] = s;
        V.dloc[3 + a] = dy[Z_PI + a] + 0.5 * dc.dt * dv[3 + a];
    }
    point_dirs<0>(S.K, dy + Z_Q, V.dpt);
    // directional derivative of f
    double dphi_f = 0;
    BMPC_UNROLL
    for (int a = 0; a < 6; a++) dphi_f += g12[a] * V.dloc[a] + g12[6 + a] * dv[a];
    BMPC_UNROLL
    for (int j = 2; j <= 4; j++) dphi_f += 2 * wts[6] * S.y[Z_DQ + j] * dy[Z_DQ + j];
    BMPC_UNROLL
    for (int j = 0; j < 7; j++) dphi_f += 2 * wts[7] * S.y[Z_U + j] * dy[Z_U + j];
    dphi_f += 2 * wts[9] * S.y[Z_RS] * dy[Z_RS] + 2 * wts[10] * S.y[Z_DRS] * dy[Z_DRS] +
              2 * wts[9] * S.y[Z_PS] * dy[Z_PS] + 2 * wts[10] * S.y[Z_DPS] * dy[Z_DPS];
    if (term)
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) {
            double gg = 2 * wts[10] * S.y[Z_D + i] + (i != 4 ? 2 * wts[8] * (pg[P_SLACKS0 + i] + S.y[Z_D + i]) : 0.0);
            dphi_f += gg * dy[Z_D + i];
        }
    walk_rows(pg, lbx, ubx, N, k, S.y, S.zeta, S.K, S.C, V);
    if (m.valid) {
        GD P = A.part + m.pi;
        P[PT_DPHIF * A.NP] = dphi_f;
        P[PT_AP * A.NP] = (V.rp > 0) ? V.tau / V.rp : 1.0; P[PT_AD * A.NP] = (V.rdn > 0) ? V.tau / (V.rdn / V.rdd) : 1.0; P[PT_DBAR * A.NP] = V.dbar;
    }
    // the pairs of an instance are lanes of this wavefront: line-search start per instance (a kernel of its own, then part
    // of the streaming row kernel k_rowstep, before the row steps moved in here)
    BMPC_FENCE_SYNC();
    if (m.valid && m.k == 1) ls0_instance(A, m.b);
}

// ------------------------------------------------------------------------------------------
// k_trial: zeta + alpha dz, t + alpha dt -> f, theta, sum log t
// ------------------------------------------------------------------------------------------
struct TrialVisitor {
    const PipeArgs* A; size_t pi; bool valid;
    double alpha;
    GCD tc; GD tn_out;                                // cur_t (read) and oth_t (the trial slacks go there) of the instance
    // the multipliers' step does not depend on the primal step length: the first trial of an iteration also writes
    // z + alpha_dual dz_row, dz_row = (mu - z c) / t, into the other copy of z (what the streaming kernel k_accept did with a
    // dz_row array that k_step stored) -- whichever trial is accepted, the flip bit makes it current together with t and zeta
    bool dual; double ad, mu; GCD zc; GD zn_out;
    double gt[ROW_GROUP_MAX], gc[ROW_GROUP_MAX];      // slack t and t + dt (k_step's c) of the current row group, one batch of loads
    double thr;                                       // row part of theta
    double lp; int le;                                // sum log t of the trial point as log of the running product lp * 2^le
                                                      // (mantissa renormalised every row: one log per pair instead of one per row)
    template <int S0, int CNT> BMPC_INL void group(unsigned act) {      // bit i of act: slot S0 + i may be a constraint row (else: not loaded)
        static_assert(CNT <= ROW_GROUP_MAX, "row group size");
        // (z is loaded with t and c, one memory round trip per group, also by the later trials of a search, which do not use it)
        double gz[CNT];
        BMPC_UNROLL
        for (int i = 0; i < CNT; i++) {
            size_t o = (size_t)(S0 + i) * A->NP + pi;
            if ((act >> i) & 1u) { gt[i] = tc[o]; gc[i] = A->dt[o]; gz[i] = zc[o]; } else { gt[i] = 1.0; gc[i] = 1.0; gz[i] = 0.0; }
        }
        if (dual) {
            // a slot is live iff z > 0 (inactive slots keep t = 1, z = 0 in both copies from k_init); done here, on the whole
            // group at once, so that no multiplier stays in a register while the group's rows are walked
            BMPC_UNROLL
            for (int i = 0; i < CNT; i++)
                if (gz[i] > 0.0) zn_out[(size_t)(S0 + i) * A->NP + pi] = gz[i] + ad * ((mu - gz[i] * gc[i]) * BMPC_RCP(gt[i]));
        }
    }
    BMPC_INL void fin(int s, double h) {
        // trial slack t + alpha dt (reset to -h where that is larger), its share of theta and of the barrier term
        const double t = gt[s - row_group_base(s)];
        double tn = t + alpha * (gc[s - row_group_base(s)] - t);
        if (A->o.slack_reset) tn = fmax(tn, -h);      // slack reset: never below what closes the row at the trial point
        thr += fabs(h + tn);
        int e;
        lp = frexp(lp * tn, &e); le += e;
        if (valid) tn_out[(size_t)s * A->NP + pi] = tn;
    }
    BMPC_INL void skip(int) {}
    BMPC_INL void diag(int s, int, double, double h) { fin(s, h); }
    BMPC_INL void zdiag(int s, int, double, double h) { fin(s, h); }
    BMPC_INL void pose(int s, const double*, int, double h) { fin(s, h); }
    template <int C> BMPC_INL void point_begin(unsigned act) { group<S_COL + 15 * C, 15>(act); }
    template <int C> BMPC_INL void point(int s, const double*, double h) { fin(s, h); }
    template <int C> BMPC_INL void point_end() {}
};

// One part (ROLES: stage.hpp WR_*) of the trial point of pair (b, k): trial slacks, multiplier update, the part's share of theta and of
// sum log t; WR_POSE also writes the trial zeta and adds the dynamics defect and the objective.
struct TrialPart { double f, th, thr, lp; int le; };
// t_out / zeta_out: where the trial slacks and the trial zeta go, indexed like the workspace arrays (field * A.NP + pi); null = the other
// copy of the double-buffered arrays
template <int ROLES>
BMPC_INL void trial_part(const PipeArgs& A, PGP pg, int b, int k, size_t pi, bool live, bool dual, double alpha, int flip, TrialPart& R,
                         GD t_out = nullptr, GD zeta_out = nullptr) {
    const int N = A.N, n_w = 44 * N + 6;
    const bool term = (k == N - 1);
    const DynC dc = make_dync(A.o.dt);
    GCD lbx = A.lbx + (size_t)A.src[b] * n_w;
    GCD ubx = A.ubx + (size_t)A.src[b] * n_w;
    GCD zc = cur_zeta(A, flip);
    StagePoint S;
    BMPC_UNROLL
    for (int i = 0; i < NZ; i++) S.zeta[i] = zc[(size_t)i * A.NP + pi] + alpha * A.dz[(size_t)i * A.NP + pi];
    if constexpr ((ROLES & WR_POSE) != 0) {
        GD zo = zeta_out ? zeta_out : oth_zeta(A, flip);
        if (live)
            BMPC_UNROLL
            for (int i = 0; i < NZ; i++) zo[(size_t)i * A.NP + pi] = S.zeta[i];
    }
    nat_all(S.zeta, dc, S.y);
    TrialVisitor V;
    V.A = &A; V.pi = pi; V.valid = live; V.alpha = alpha; V.thr = 0.0; V.lp = 1.0; V.le = 0; V.tc = cur_t(A, flip); V.tn_out = t_out ? t_out : oth_t(A, flip);
    V.dual = dual; V.ad = A.st[b].ad; V.mu = A.st[b].mu; V.zc = cur_z(A, flip); V.zn_out = oth_z(A, flip);
    // The walk in three phases, each with its own inputs: the box rows need the natural coordinates only and run BEFORE kinematics
    // and reference context exist; the pose rows need the context; the collision points need the joint origins only.  (One walk
    // over everything kept coordinates, context and kinematics alive through all row groups: the kernel's spills.)
    if constexpr ((ROLES & WR_BOX) != 0) walk_rows<TrialVisitor, WR_BOX>(pg, lbx, ubx, N, k, S.y, S.zeta, S.K, S.C, V);
    BMPC_SCHED_FENCE();
    if constexpr ((ROLES & (WR_POSE | WR_PT0 | WR_PT1)) != 0) kin_chain(A.rc, S.y + Z_Q, S.K);
    R.th = 0.0; R.f = 0.0;
    if constexpr ((ROLES & WR_POSE) != 0) {
        double iw0[3];
        BMPC_UNROLL
        for (int c = 0; c < 3; c++) iw0[c] = lbx[28 * N + (3 + c) * N];
        kin_jlin(S.K, S.Jl);
        kin_vel(S.K, S.Jl, S.y + Z_DQ, S.C.v);
        // dynamics / initial-state part of theta: here, before the reference context, so that zeta is dead while that is alive
        double th = 0;
        if (!term) {
            double zn[NX], rdef[NX];
            BMPC_UNROLL
            for (int i = 0; i < NX; i++) zn[i] = zc[(size_t)i * A.NP + pi + 1] + alpha * A.dz[(size_t)i * A.NP + pi + 1];
            defect_all(S.zeta, zn, S.C.v + 3, dc, rdef);
            BMPC_UNROLL
            for (int i = 0; i < NX; i++) th += fabs(rdef[i]);
        }
        if (k == 1) {
            double x1fix[24];
            x1fix_eval(lbx, N, dc.dt, x1fix);
            BMPC_UNROLL
            for (int i = 0; i < 24; i++) th += fabs(x1fix[i] - S.zeta[i]);
        }
        BMPC_SCHED_FENCE();
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) { S.C.pose[a] = S.K.pee[a]; S.C.pose[3 + a] = S.y[Z_PI + a] + 0.5 * dc.dt * S.C.v[3 + a]; }
        seg_ctx_eval(pg, A.N, k, S.y, iw0, S.C);          // (= stage_point)
        walk_rows<TrialVisitor, WR_POSE>(pg, lbx, ubx, N, k, S.y, S.zeta, S.K, S.C, V);
        R.th = th; R.f = S.C.fv;
    }
    BMPC_SCHED_FENCE();
    if constexpr ((ROLES & (WR_PT0 | WR_PT1)) != 0) walk_rows<TrialVisitor, ROLES & (WR_PT0 | WR_PT1)>(pg, lbx, ubx, N, k, S.y, S.zeta, S.K, S.C, V);
    R.thr = V.thr; R.lp = V.lp; R.le = V.le;
}

// NW = 1: one wavefront per group of pairs walks all rows (rounds 1-4).  NW = 4 (round 4): a workgroup of four wavefronts per group
// of pairs, each wavefront one part of the walk -- pose rows + defect + objective / box rows / collision points 0-2 / 3-5.  No
// wavefront holds the natural coordinates, the reference context AND the point positions: the parts fit half the register file (two
// workgroups per CU instead of one wavefront per SIMD that spills), and a pair's dependent memory round trips -- one per row group --
// run side by side in the four wavefronts instead of one after the other.  tid = thread index in the workgroup.
template <int NW>
BMPC_KBODY void k_trial_body_t(const PipeArgs& A, int wave, int tid, LDSD* lds_par) {
    static_assert(NW == 1 || NW == 4, "parts of the trial walk");
    const int count = A.L.cnt[2], N = A.N;
    if (wave * ipw_of(N) >= count) return;
    const int lane = tid & 63, role = tid >> 6;
    const PairMap m = pair_map(A, A.L.trial, count, wave, lane);
    const int ipw = ipw_of(N);
    (void)stage_params<true, 64 * NW>(A, A.L.trial, count, wave, tid, m, lds_par);
    // The pairs of an instance are lanes of this wavefront, so the filter test of a trial point needs nobody else: the
    // wavefront backtracks by itself -- trial, test, half the step length, again (at most 1 + trial_repeats trials per
    // super-step) -- until each of its instances has an accepted point.  Rounds 1-2 (and trial_repeats = 0) tested in k_accept
    // and gave a rejected instance its next trial one super-step later, after the evaluation and Riccati passes of everybody
    // else.  Scheduling only: an instance sees the same sequence of trials either way.
    LDSD* ended = lds_par + (size_t)ipw * NPARL;         // [IPW_MAX] 1: the instance's line search has ended (or no such instance)
    LDSD* comb = ended + IPW_MAX;                        // NW == 4: [part][lane][TRIAL_COMB] partial results of the parts
    if (tid < IPW_MAX) ended[tid] = (tid < ipw && wave * ipw + tid < count) ? 0.0 : 1.0;
    BMPC_SYNC();
    const int flip = A.st[m.b].flip;                     // (flipped by the accepting test: read once, the instance is dead then)
    for (int round = 0;; round++) {
        // the lane's coordinates pass through an opaque copy every round: everything below is recomputed from them, so the
        // compiler cannot hoist the (hundreds of) round-invariant parameter loads out of the loop into registers it does not have
        int b = m.b, k = m.k, li = m.li;
        BMPC_OPAQUE_I(b); BMPC_OPAQUE_I(k); BMPC_OPAQUE_I(li);
        const size_t pi = pair_of(A, b, k);
        PGP pg = lds_par + li * NPARL;
        const bool live = m.valid && ended[li] == 0.0;
        const double alpha = live ? A.st[b].alpha : 0.0;
        TrialPart R;
        if constexpr (NW == 1) trial_part<WR_ALL>(A, pg, b, k, pi, live, live && round == 0, alpha, flip, R);
        else {
            TrialPart Q;
#ifdef BMPC_TRIAL_ONLY      // (register-need experiments: one part only)
            Q.thr = 0; Q.lp = 1; Q.le = 0; Q.th = 0; Q.f = 0;
            if (role == 0) trial_part<BMPC_TRIAL_ONLY>(A, pg, b, k, pi, live, live && round == 0, alpha, flip, Q);
#else
            if (role == 0) trial_part<WR_POSE>(A, pg, b, k, pi, live, live && round == 0, alpha, flip, Q);
            else if (role == 1) trial_part<WR_BOX>(A, pg, b, k, pi, live, live && round == 0, alpha, flip, Q);
            else if (role == 2) trial_part<WR_PT0>(A, pg, b, k, pi, live, live && round == 0, alpha, flip, Q);
            else trial_part<WR_PT1>(A, pg, b, k, pi, live, live && round == 0, alpha, flip, Q);
#endif
            LDSD* c = comb + (size_t)(role * 64 + lane) * TRIAL_COMB;
            c[0] = Q.thr; c[1] = Q.lp; c[2] = (double)Q.le;
            if (role == 0) { c[3] = Q.th; c[4] = Q.f; }
            BMPC_SYNC();
            // the parts' shares in part order (fixed: the result does not depend on which wavefront finished first)
            R.thr = 0.0; R.lp = 1.0; R.le = 0;
            if (role == 0) {
                BMPC_UNROLL
                for (int r = 0; r < NW; r++) {
                    const LDSD* cr = comb + (size_t)(r * 64 + lane) * TRIAL_COMB;
                    R.thr += cr[0]; R.lp *= cr[1]; R.le += (int)cr[2];
                }
                R.th = c[3]; R.f = c[4];
            }
        }
        if (live && role == 0) {
            GD P = A.part + pi;
            P[PT_F1 * A.NP] = R.f; P[PT_TH1 * A.NP] = R.th + R.thr;
            P[PT_LS1 * A.NP] = log(R.lp) + (double)R.le * 0.69314718055994530942;
        }
        const bool last = round >= A.o.trial_repeats;
        BMPC_FENCE_SYNC();
        if (live && role == 0 && k == 1 && (ls_instance(A, b, last) || last)) ended[li] = 1.0;
        BMPC_FENCE_SYNC();
        bool left = false;
        for (int q = 0; q < ipw; q++) left = left || (ended[q] == 0.0);
        if (!left) break;
    }
}
BMPC_KBODY void k_trial_body(const PipeArgs& A, int wave, int lane, LDSD* lds_par) { k_trial_body_t<1>(A, wave, lane, lds_par); }

// ------------------------------------------------------------------------------------------
// k_trial_spec (tail regime: fewer groups of pairs than CUs): the step lengths of a line search are known before any of them
// is tried -- alpha, alpha / 2, alpha / 4, ... -- and their acceptance tests do not depend on each other (the filter changes only
// when a point is accepted).  A workgroup of TRIAL_SPEC wavefronts per group of pairs evaluates that many trial points side by side,
// the wavefront of the first one into the other copy of the double-buffered arrays as k_trial does, the others into candidate
// records (the gain copies of the speculative Riccati attempts: dead once k_fwd has run); the lane of stage 1 then walks the tests
// in the order of the sequential search (ls_decide, the same code), and the candidate of the first accepted step length is copied
// over.  Every trial point and every test is the arithmetic k_trial would have done, one after the other: bitwise the same
// iterate, at one trial's latency (+ a copy) instead of up to ten.  Slot-major layout only (candidate records are indexed like a slot's
// own arrays).  Always ends the search (trial_repeats is scheduling only).  tid = thread index in the workgroup.
// ------------------------------------------------------------------------------------------
constexpr int TRIAL_SPEC = 4;
static_assert(TRIAL_SPEC - 1 <= RIC_NATT - 1 && NSLOT + NZ <= KREC, "candidate trial points fit the speculative gain copies");
static_assert(TRIAL_SPEC * 64 * 3 + IPW_MAX <= TRIAL_SPEC * 64 * TRIAL_COMB, "k_trial_spec's LDS fits trial_lds_doubles(N, 4)");
BMPC_KBODY void k_trial_spec_body(const PipeArgs& A, int wave, int tid, LDSD* lds_par) {
    const int count = A.L.cnt[2], N = A.N;
    if (wave * ipw_of(N) >= count) return;
    const int lane = tid & 63, role = tid >> 6;
    const PairMap m = pair_map(A, A.L.trial, count, wave, lane);
    const int ipw = ipw_of(N), S = N - 1;
    (void)stage_params<true, 64 * TRIAL_SPEC>(A, A.L.trial, count, wave, tid, m, lds_par);
    LDSD* ended = lds_par + (size_t)ipw * NPARL;         // [IPW_MAX] 1: the instance's line search has ended (or no such instance)
    LDSD* comb = ended + IPW_MAX;                        // [trial][lane][3]: f, theta, sum log t of the pair at that trial point
    LDSD* win = comb + TRIAL_SPEC * 64 * 3;              // [IPW_MAX] which trial of this round became the iterate (-1: none)
    if (tid < IPW_MAX) ended[tid] = (tid < ipw && wave * ipw + tid < count) ? 0.0 : 1.0;
    BMPC_SYNC();
    const int flip = A.st[m.b].flip;                     // (flipped by the accepting test: read once, the instance is dead then)
    for (int round = 0;; round++) {
        int b = m.b, k = m.k, li = m.li;
        BMPC_OPAQUE_I(b); BMPC_OPAQUE_I(k); BMPC_OPAQUE_I(li);
        const size_t pi = pair_of(A, b, k);
        PGP pg = lds_par + li * NPARL;
        const bool live = m.valid && ended[li] == 0.0;
        const int bt0 = live ? A.st[b].bt : 0;                              // index of this round's first trial in the search
        const bool mine = live && bt0 + role <= 9;                          // (a search has at most ten trials)
        double alpha = live ? A.st[b].alpha : 0.0;
        for (int r = 0; r < role; r++) alpha *= 0.5;                        // (exactly what the sequential search's halvings give)
        // candidate record of (slot, role): [field][stage], fields 0 .. NSLOT-1 = t, then zeta
        GD cand = A.kspec + ((size_t)b * (RIC_NATT - 1) + (size_t)(role > 0 ? role - 1 : 0)) * ((size_t)S * KREC);
        GD t_out = role > 0 ? cand + (size_t)(k - 1) - pi : nullptr;
        GD zeta_out = role > 0 ? cand + (size_t)NSLOT * S + (size_t)(k - 1) - pi : nullptr;
        TrialPart R;
        trial_part<WR_ALL>(A, pg, b, k, pi, mine, mine && role == 0 && round == 0, alpha, flip, R, t_out, zeta_out);
        {
            LDSD* c = comb + (size_t)(role * 64 + lane) * 3;
            c[0] = R.f; c[1] = R.th + R.thr; c[2] = log(R.lp) + (double)R.le * 0.69314718055994530942;
        }
        if (tid < IPW_MAX) win[tid] = -1.0;
        BMPC_FENCE_SYNC();
        if (live && role == 0 && k == 1) {
            // the tests in the order of the sequential search; the pieces of a trial point summed in pair order, as ls_instance does
            for (int j = 0; j < TRIAL_SPEC && bt0 + j <= 9; j++) {
                double f1 = 0, th1 = 0, ls1 = 0;
                for (int kk = 0; kk < S; kk++) {
                    const LDSD* c = comb + (size_t)(j * 64 + lane + kk) * 3;
                    f1 += c[0]; th1 += c[1]; ls1 += c[2];
                }
                if (ls_decide(A, b, f1, th1, ls1, false)) { win[li] = (double)j; ended[li] = 1.0; break; }
            }
        }
        BMPC_FENCE_SYNC();
        // the accepted candidate becomes the other copy (trial 0 is there already); the workgroup's wavefronts share the fields
        if (m.valid) {
            const int w = (int)win[li];
            if (w > 0) {
                GCD src = A.kspec + ((size_t)b * (RIC_NATT - 1) + (size_t)(w - 1)) * ((size_t)S * KREC) + (size_t)(k - 1);
                GD to = oth_t(A, flip), zo = oth_zeta(A, flip);
                for (int f = role; f < NSLOT; f += TRIAL_SPEC) to[(size_t)f * A.NP + pi] = src[(size_t)f * S];
                for (int f = role; f < NZ; f += TRIAL_SPEC) zo[(size_t)f * A.NP + pi] = src[(size_t)(NSLOT + f) * S];
            }
        }
        bool left = false;
        for (int q = 0; q < ipw; q++) left = left || (ended[q] == 0.0);
        if (!left) break;
        BMPC_FENCE_SYNC();           // (win / comb are rewritten in the next round)
    }
}

// ------------------------------------------------------------------------------------------
// k_out: x in the reference layout, constraint vector g, violation partials
// ------------------------------------------------------------------------------------------
struct OutVisitor {
    double viol;
    GD gi;               // inequality rows of this stage in the reference order, or null
    PGP pg; const SegCtx* C; const double* y; bool term;
    BMPC_INL void rowv(int s, double h, bool lower) {
        if (h > 1e-6) viol += h;
        if (gi) gi[s - S_EE] = lower ? -h : h;
    }
    BMPC_INL void skip(int s) {
        if (!gi || s < S_EE || s >= S_END || (s >= S_TSET && !term)) return;
        double v;
        if (s < S_ROTU) v = -pg[P_BSET + (s - S_EE) * 4 + C->s] - y[Z_PS];
        else if (s < S_PHI) { int c = (s - S_COL) / 15, rr = (s - S_COL) - 15 * c; v = -pg[P_BSETJ + rr * 6 + c] - C->sl[c]; }
        else v = -pg[P_BSET + (s - S_TSET) * 4 + C->n] - C->sl[5];
        gi[s - S_EE] = v;
    }
    BMPC_INL void diag(int, int, double, double) {}     // box bounds are not part of g / viol
    BMPC_INL void zdiag(int, int, double, double) {}
    BMPC_INL void pose(int s, const double*, int, double h) { rowv(s, h, (s >= S_ROTL && s < S_COL) || (s >= S_TROTL)); }
    template <int S0, int CNT> BMPC_INL void group(unsigned) {}
    template <int C_> BMPC_INL void point_begin(unsigned) {}
    template <int C_> BMPC_INL void point(int s, const double*, double h) { rowv(s, h, false); }
    template <int C_> BMPC_INL void point_end() {}
};

BMPC_KBODY void k_out_body(const PipeArgs& A, int wave, int lane, LDSD* lds_par) {
    const int count = A.L.cnt[8], N = A.N;          // the done list: instances that finished since the last retirement
    if (wave * ipw_of(N) >= count) return;
    PairMap m = pair_map(A, A.L.done, count, wave, lane);
    PGP pg = stage_params(A, A.L.done, count, wave, lane, m, lds_par);
    if (!m.valid) return;
    const int k = m.k, n_w = 44 * N + 6;
    const size_t b = (size_t)A.src[m.b];            // output row of the instance
    const bool term = (k == N - 1);
    const size_t pi = m.pi;
    const DynC dc = make_dync(A.o.dt);
    GCD lbx = A.lbx + b * n_w;
    GCD ubx = A.ubx + b * n_w;
    double iw0[3];
    BMPC_UNROLL
    for (int c = 0; c < 3; c++) iw0[c] = lbx[28 * N + (3 + c) * N];
    StagePoint S;
    GCD zc = cur_zeta(A, A.st[m.b].flip);
    load_zeta(zc, A.NP, pi, S.zeta);           // the iterate the instance finished on
    stage_point(A, pg, iw0, k, dc, S);
    GD x = A.x + b * n_w;
    BMPC_UNROLL
    for (int j = 0; j < 7; j++) {
        x[j * N + k] = S.y[Z_Q + j]; x[7 * N + j * N + k] = S.y[Z_DQ + j];
        x[14 * N + j * N + k] = S.y[Z_DDQ + j]; x[21 * N + j * N + k] = S.y[Z_U + j];
    }
    BMPC_UNROLL
    for (int c = 0; c < 6; c++) { x[28 * N + c * N + k] = S.C.pose[c]; x[34 * N + c * N + k] = S.C.v[c]; }
    x[40 * N + 6 + k] = S.y[Z_RS]; x[41 * N + 6 + k] = S.y[Z_DRS];
    x[42 * N + 6 + k] = S.y[Z_PS]; x[43 * N + 6 + k] = S.y[Z_DPS];
    if (k == 1) {
        // stage-0 column: the pins (lbx == ubx there), rs_0 / ps_0 from the eliminated stage-0 slacks
        BMPC_UNROLL
        for (int blk = 0; blk < 4; blk++)
            BMPC_UNROLL
            for (int j = 0; j < 7; j++) x[blk * 7 * N + j * N] = lbx[blk * 7 * N + j * N];
        BMPC_UNROLL
        for (int c = 0; c < 6; c++) { x[28 * N + c * N] = lbx[28 * N + c * N]; x[34 * N + c * N] = lbx[34 * N + c * N]; }
        x[40 * N + 6] = S.zeta[Z_RS]; x[41 * N + 6] = 0.0; x[42 * N + 6] = S.zeta[Z_PS]; x[43 * N + 6] = 0.0;
    }
    if (term)
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) x[40 * N + i] = S.y[Z_D + i];
    // violation as BoundMPC.py:613-615 (g rows only, 1e-6 dead band) and the g vector
    GD g = A.g ? A.g + b * (147 * (N - 1) + 21) : nullptr;
    OutVisitor V;
    V.viol = 0; V.gi = g ? g + 35 * (N - 1) + 112 * (k - 1) : nullptr; V.pg = pg; V.C = &S.C; V.y = S.y; V.term = term;
    walk_rows(pg, lbx, ubx, N, k, S.y, S.zeta, S.K, S.C, V);
    double rdef[NX];
    if (!term) {
        double zn[NX];
        BMPC_UNROLL
        for (int i = 0; i < NX; i++) zn[i] = zc[(size_t)i * A.NP + pi + 1];
        defect_all(S.zeta, zn, S.C.v + 3, dc, rdef);
        BMPC_UNROLL
        for (int i = 0; i < Z_D; i++) { double r = fabs(rdef[i]); if (r > 1e-6) V.viol += r; }
        if (g) {
            GD ge = g + 35 * k;
            BMPC_UNROLL
            for (int i = 0; i < 35; i++) ge[i] = 0.0;
            BMPC_UNROLL
            for (int i = 0; i < 21; i++) ge[i] = rdef[i];
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) ge[24 + a] = rdef[Z_PI + a];
            ge[33] = rdef[Z_RS]; ge[34] = rdef[Z_PS];
        }
    }
    if (k == 1) {
        double x1fix[24];
        x1fix_eval(lbx, N, dc.dt, x1fix);
        BMPC_UNROLL
        for (int i = 0; i < 24; i++) { double r = fabs(x1fix[i] - S.zeta[i]); if (r > 1e-6) V.viol += r; }
        if (g) {
            BMPC_UNROLL
            for (int i = 0; i < 35; i++) g[i] = 0.0;
            BMPC_UNROLL
            for (int i = 0; i < 21; i++) g[i] = x1fix[i] - S.zeta[i];
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) g[24 + a] = x1fix[Z_PI + a] - S.zeta[Z_PI + a];
        }
    }
    A.part[PT_F1 * A.NP + pi] = V.viol;      // reduced per instance by k_fin
}

// ------------------------------------------------------------------------------------------
// k_mult: multipliers of the full-space NLP in the reference's layout and CasADi's sign convention
// (grad f + J_g^T lam_g + lam_x = 0; BoundMPC.py:638-645 reads sol["lam_g"], sol["lam_x"]).  Thread per pair, at the
// final point of the instance: inequality rows of g and bound rows of x get the row multipliers z of the barrier method;
// the equality rows (35 per stage transition, casadi_ocp_formulation.py:145-164) follow from stationarity in the one
// variable each of them carries a -1 on -- an adjoint sweep backwards over the stages (k_mult_sweep) for which this
// kernel leaves the stage-local terms in a per-pair record.  Algebra: oracle/bmpc_solve.c recover_multipliers.
// ------------------------------------------------------------------------------------------
struct MultVisitor {
    const PipeArgs* A; size_t pi; int N, k;
    GCD zc;              // cur_z of the instance
    GD lg;               // inequality rows of this stage in lam_g
    GD lx;               // lam_x of the instance
    double bz[6], Fc[6][3], sPS, sRS, sD[6], lxq[7], lxdq[7], lxddq[7], znn[4], z1[2];
    BMPC_INL double zrow(int s) const { return zc[(size_t)s * A->NP + pi]; }
    BMPC_INL void skip(int s) {
        if (s >= S_EE && s < (k == N - 1 ? S_END : S_TSET)) lg[s - S_EE] = 0.0;
    }
    BMPC_INL void diag(int s, int pos, double coef, double) {
        const double z = zrow(s);
        if (s < S_NONNEG) {
            const int blk = s / 14, jj = (s / 2) % 7;
            lx[(size_t)blk * 7 * N + (size_t)jj * N + k] += coef * z;       // both bound rows of a variable add up
            if (blk == 0) lxq[jj] += coef * z; else if (blk == 1) lxdq[jj] += coef * z; else if (blk == 2) lxddq[jj] += coef * z;
        } else if (s < S_RS1) {
            const int m = s - S_NONNEG;                                          // rs, drs, ps, dps >= 0
            lx[(size_t)(40 + m) * N + 6 + k] = -z;
            znn[m] = z;
        } else lx[(size_t)40 * N + (s - S_D1)] = -z;                              // dslacks >= 0 (k == 1 only)
    }
    BMPC_INL void zdiag(int s, int, double, double) { z1[s - S_RS1] = zrow(s); }
    BMPC_INL void pose(int s, const double* a, int sel, double) {
        const double z = zrow(s);
        const bool lower = (s >= S_ROTL && s < S_COL) || (s >= S_TROTL);
        lg[s - S_EE] = lower ? -z : z;
        BMPC_UNROLL
        for (int c = 0; c < 6; c++) bz[c] += z * a[c];
        if (sel == 1) sPS -= z; else if (sel == 2) sRS -= z; else if (sel == 3) sD[5] -= z;
    }
    template <int S0, int CNT> BMPC_INL void group(unsigned) {}
    template <int C> BMPC_INL void point_begin(unsigned) {}
    template <int C> BMPC_INL void point(int s, const double* a, double) {
        const double z = zrow(s);
        lg[s - S_EE] = z;
        Fc[C][0] += z * a[0]; Fc[C][1] += z * a[1]; Fc[C][2] += z * a[2];
        sD[C] -= z;
    }
    template <int C> BMPC_INL void point_end() {}
};

template <int C>
BMPC_INL void point_forces_to_q(const KinT& K, const double Fc[6][3], double* cq) {
    if constexpr (C < 6) {
        constexpr int nj = PointNJ<C>::value;
        const double* pc = kin_point<C>(K);
        BMPC_UNROLL
        for (int i = 0; i < nj; i++) {
            double r[3] = {pc[0] - K.o[i][0], pc[1] - K.o[i][1], pc[2] - K.o[i][2]}, c[3];
            cross3r(K.zx[i], r, c);
            cq[i] += c[0] * Fc[C][0] + c[1] * Fc[C][1] + c[2] * Fc[C][2];
        }
        point_forces_to_q<C + 1>(K, Fc, cq);
    }
}

BMPC_KBODY void k_mult_body(const PipeArgs& A, int wave, int lane, LDSD* lds_par) {
    const int count = A.B, N = A.N;
    if (wave * ipw_of(N) >= count) return;
    PairMap m;
    {
        const int S_ = N - 1, ipw = ipw_of(N);
        int li = lane / S_, kk = lane - li * S_;
        int e = wave * ipw + li;
        m.valid = (li < ipw) && (e < count);
        if (!m.valid) { e = wave * ipw; kk = 0; li = 0; }
        m.b = e; m.k = kk + 1; m.li = li; m.pi = pair_of(A, m.b, m.k);
    }
    PGP pg = stage_params(A, (GCI)nullptr, count, wave, lane, m, lds_par);
    if (!m.valid) return;
    const int k = m.k, n_w = 44 * N + 6, n_g = 147 * (N - 1) + 21;
    const size_t b = (size_t)A.src[m.b];
    const bool term = (k == N - 1);
    const size_t pi = m.pi;
    const DynC dc = make_dync(A.o.dt);
    GCD lbx = A.lbx + b * n_w;
    GCD ubx = A.ubx + b * n_w;
    PGP wts = pg + P_W;
    double iw0[3];
    BMPC_UNROLL
    for (int c = 0; c < 3; c++) iw0[c] = lbx[28 * N + (3 + c) * N];
    StagePoint S;
    load_zeta(cur_zeta(A, A.st[m.b].flip), A.NP, pi, S.zeta);     // the final point of the instance
    stage_point(A, pg, iw0, k, dc, S);
    double G[6][7], g12[12];
    kin_G(S.K, S.Jl, S.y + Z_DQ, G);
    cost_grad12(pg, S.C, term, g12);
    GD lx = A.lam_x + b * n_w;
    // bound multipliers of this stage's columns start from zero (p and v are unbounded)
    BMPC_UNROLL
    for (int f = 0; f < 40; f++) lx[(size_t)f * N + k] = 0.0;
    MultVisitor V;
    V.A = &A; V.pi = pi; V.N = N; V.k = k; V.zc = cur_z(A, A.st[m.b].flip);
    V.lg = A.lam_g + b * n_g + 35 * (N - 1) + 112 * (k - 1);
    V.lx = lx;
    V.sPS = 0; V.sRS = 0; V.z1[0] = 0; V.z1[1] = 0;
    BMPC_UNROLL
    for (int c = 0; c < 6; c++) { V.bz[c] = 0; V.sD[c] = 0; V.Fc[c][0] = 0; V.Fc[c][1] = 0; V.Fc[c][2] = 0; }
    BMPC_UNROLL
    for (int j = 0; j < 7; j++) { V.lxq[j] = 0; V.lxdq[j] = 0; V.lxddq[j] = 0; }
    BMPC_UNROLL
    for (int i = 0; i < 4; i++) V.znn[i] = 0;
    walk_rows(pg, lbx, ubx, N, k, S.y, S.zeta, S.K, S.C, V);
    double bz[6], bzv[6], fa[3];
    BMPC_UNROLL
    for (int c = 0; c < 6; c++) { bz[c] = V.bz[c] + g12[c]; bzv[c] = g12[6 + c]; }
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) fa[a] = bzv[3 + a] + 0.5 * dc.dt * bz[3 + a];     // stage-local part of lam_v (angular)
    double cq[7], cdq[7];
    BMPC_UNROLL
    for (int j = 0; j < 7; j++) {
        double sq = V.lxq[j], sd = V.lxdq[j];
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) {
            sq += S.Jl[a][j] * bz[a] + G[a][j] * bzv[a] + G[3 + a][j] * fa[a];
            sd += S.Jl[a][j] * bzv[a] + S.K.zx[j][a] * fa[a];
        }
        if (j >= 2 && j <= 4) sd += 2 * wts[6] * S.y[Z_DQ + j];                      // joint-velocity cost (Q9)
        cq[j] = sq; cdq[j] = sd;
    }
    point_forces_to_q<0>(S.K, V.Fc, cq);
    GD rec = A.hrec + hrec_of(A, m.b, m.k);
    BMPC_UNROLL
    for (int j = 0; j < 7; j++) { rec[M_CQ + j] = cq[j]; rec[M_CDQ + j] = cdq[j]; rec[M_CDDQ + j] = V.lxddq[j]; }
    BMPC_UNROLL
    for (int a = 0; a < 3; a++)
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) { rec[M_GANG + 7 * a + j] = G[3 + a][j]; rec[M_ZX + 7 * a + j] = S.K.zx[j][a]; }
    BMPC_UNROLL
    for (int c = 0; c < 6; c++) { rec[M_BZ + c] = bz[c]; rec[M_BZV + c] = bzv[c]; }
    rec[M_GRS] = 2 * wts[9] * S.y[Z_RS] + V.sRS - V.znn[0];
    rec[M_GPS] = 2 * wts[9] * S.y[Z_PS] + V.sPS - V.znn[2];
    if (k == 1) {
        // stage-0 column: zero here, filled from stationarity by k_mult_sweep
        BMPC_UNROLL
        for (int f = 0; f < 40; f++) lx[(size_t)f * N] = 0.0;
        BMPC_UNROLL
        for (int f = 0; f < 4; f++) lx[(size_t)(40 + f) * N + 6] = 0.0;
    }
}

}  // namespace bmpc
