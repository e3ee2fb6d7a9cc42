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

#ifndef BMPC_AS1
#define BMPC_AS1
#endif

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
// LDS copy of the parameter vector: [0,275) verbatim, then a_set_joints (270) and b_set_joints (90);
// the EE sets a_set/b_set stay in global memory (read by two row slots per stage only)
constexpr int SP_ASETJ = 275, SP_BSETJ = 545, NSP = 635;

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

struct RobotConst {           // chain constants of the handle's robot (include/boundmpc.h bmpc_robot), rotations precomputed on the host
    double jxyz[7][3];
    double jrot[7][9];
    double ee_xyz[3];
    double ee_rot[9];
    double l4c_xyz[3];
    // limits and collision-sphere radii: used by the device-resident loop (bounds rows, collision sets)
    double q_lo[7], q_hi[7], dq_max[7], ddq_max, u_max, colsize[6];
};

typedef BMPC_AS1 double* GD;               // pointers into global memory (device code of the pipeline)
typedef BMPC_AS1 const double* GCD;
typedef BMPC_AS1 int* GI;
typedef BMPC_AS1 const int* GCI;

typedef BMPC_AS1 const RobotConst* GRC;

struct KernelArgs {
    int B;
    SolverOpts o;
    const RobotConst* rc;
    const double *x0, *lbx, *ubx, *p;   // [B][n_w] x3, [B][875]
    double *x, *f, *viol, *g;           // [B][n_w], [B], [B], [B][n_g] or null
    int *iters, *status;
    double* ws;                         // scratch: nblocks * ws_doubles(N)
    int* counter;                       // next instance to hand out (zeroed before every launch)
    double* prof;                       // diagnostic builds: per-block phase cycle sums (16 per block) or null
};

struct Inst {            // per-instance arguments, passed by value (registers)
    int N, b;
    const double *lbx, *ubx, *pg;   // this instance's rows of lbx / ubx / p
    double* prof;                   // diagnostic builds only
};

constexpr int WS_EVAL = 624;   // >= EVAL_DOUBLES, per-stage cache of the evaluation block
BMPC_HD int ws_doubles(int N) { return N * (3 * ZPAD + 5 * NSLOT + NU * NX + 32 + WS_EVAL); }

// ------------------------------------------------------------------------------------------
// LDS carve-up (doubles)
// ------------------------------------------------------------------------------------------
// LDS layout: constexpr offsets (in doubles) from the workgroup's LDS base
constexpr int O_P = 0;
constexpr int O_W = O_P + (NX * LDP);
constexpr int O_sp = O_W + (NZ * LDW);
constexpr int O_zeta = O_sp + (NSP);
constexpr int O_znext = O_zeta + (ZPAD);
// ---- evaluation block (contiguous: cached per stage in the scratch slab, see EVAL_DOUBLES) ----
constexpr int O_yz = O_znext + (ZPAD);
constexpr int O_J = O_yz + (ZPAD);
constexpr int O_G = O_J + (42);
constexpr int O_Jp = O_G + (42);
constexpr int O_zax = O_Jp + (126);
constexpr int O_pc = O_zax + (21);
constexpr int O_rc = O_pc + (18);
constexpr int O_kin = O_rc + (160);
constexpr int EVAL_DOUBLES = ZPAD + 42 + 42 + 126 + 21 + 18 + 160 + 160;
constexpr int O_g0 = O_kin + (160);
constexpr int O_g1 = O_g0 + (ZPAD);
constexpr int O_gz = O_g1 + (ZPAD);
constexpr int O_lam = O_gz + (ZPAD);
constexpr int O_pv0 = O_lam + (NX);
constexpr int O_pv1 = O_pv0 + (NX);
constexpr int O_vt0 = O_pv1 + (NX);
constexpr int O_vt1 = O_vt0 + (NX);
constexpr int O_rdef = O_vt1 + (NX);
constexpr int O_Op = O_rdef + (NX);
constexpr int O_Ov = O_Op + (102);
constexpr int O_T1 = O_Ov + (102);
constexpr int O_T2 = O_T1 + (102);
constexpr int O_Hp = O_T2 + (102);
constexpr int O_Hv = O_Hp + (36);
constexpr int O_mS = O_Hv + (36);
constexpr int O_sS = O_mS + (18);
constexpr int O_bp0 = O_sS + (3);
constexpr int O_bp1 = O_bp0 + (6);
constexpr int O_bpz = O_bp1 + (6);
constexpr int O_bv = O_bpz + (6);
constexpr int O_bS0 = O_bv + (6);
constexpr int O_bS1 = O_bS0 + (3);
constexpr int O_bSz = O_bS1 + (3);
constexpr int O_M3 = O_bSz + (3);
constexpr int O_mc = O_M3 + (54);
constexpr int O_sc = O_mc + (18);
constexpr int O_b30 = O_sc + (6);
constexpr int O_b31 = O_b30 + (18);
constexpr int O_b3z = O_b31 + (18);
constexpr int O_bc0 = O_b3z + (18);
constexpr int O_bc1 = O_bc0 + (6);
constexpr int O_bcz = O_bc1 + (6);
constexpr int O_rowS = O_bcz + (6);
constexpr int O_rowA = O_rowS + (4 * NSLOT);
constexpr int O_rowSl = O_rowA + (NPOSE * 6);
constexpr int O_kf = O_rowSl + (NPOSE);
constexpr int O_red = O_kf + (32);
constexpr int O_dx = O_red + (BMPC_NT);
constexpr int O_dxn = O_dx + (NX);
constexpr int O_dloc = O_dxn + (NX);
constexpr int O_dpt = O_dloc + (16);
constexpr int O_x1fix = O_dpt + (24);
constexpr int O_r0 = O_x1fix + (24);
constexpr int O_misc = O_r0 + (NX);
constexpr int O_rob = O_misc + (64);
constexpr int LDS_DOUBLES = O_rob + (96);
constexpr int O_Kl = O_Op;
constexpr int O_Y = O_rowA;
constexpr int O_Et = O_rowA + NZ * 3;

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
template <class PA, class PB> BMPC_INL double dot3(PA a, PB b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class PA, class PB> BMPC_INL void cross3(PA a, PB b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
template <class PA, class PB> BMPC_INL void mat3mul(PA A, PB B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
template <class PA, class PB> BMPC_INL void mat3vec(PA A, PB v, double* r) {
    for (int i = 0; i < 3; i++) r[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

// workgroup-wide reductions through LDS (fixed summation order -> reproducible)
BMPC_DEV double wg_sum(double v, LDSD* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = 0;
    for (int i = 0; i < BMPC_NT; i++) s += red[i];
    return s;
}
BMPC_DEV double wg_max(double v, LDSD* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = red[0];
    for (int i = 1; i < BMPC_NT; i++) s = fmax(s, red[i]);
    return s;
}
BMPC_DEV double wg_min(double v, LDSD* red, int lane) {
    BMPC_SYNC(); red[lane] = v; BMPC_SYNC();
    double s = red[0];
    for (int i = 1; i < BMPC_NT; i++) s = fmin(s, red[i]);
    return s;
}

// sparse column structure of [As Bs] (constant part of the stage dynamics) for zeta column c:
// returns number of (x-row, coefficient) pairs.
struct DynC { double dt, b1, b2, b3, c1, c2, c3; };
struct PhiCol { int i0, i1, i2; double c0, c1, c2; };   // by value: stays in registers
BMPC_INL PhiCol phi_col(int c, const DynC d) {
    PhiCol r;
    r.i0 = c; r.i1 = c; r.i2 = c; r.c0 = 1.0; r.c1 = 0.0; r.c2 = 0.0;
    if (c < Z_DQ) {
    } else if (c < Z_DDQ) { r.i0 = c - 7; r.c0 = d.dt; r.i1 = c; r.c1 = 1.0; }
    else if (c < Z_PI) { r.i0 = c - 14; r.c0 = 0.5 * d.dt * d.dt; r.i1 = c - 7; r.c1 = d.dt; r.i2 = c; r.c2 = 1.0; }
    else if (c < Z_U) {
    } else if (c < Z_DRS) { int j = c - Z_U; r.i0 = Z_Q + j; r.c0 = d.b3; r.i1 = Z_DQ + j; r.c1 = d.b2; r.i2 = Z_DDQ + j; r.c2 = d.b1; }
    else if (c == Z_DRS) { r.i0 = Z_RS; r.c0 = d.dt; r.i1 = Z_RS; r.i2 = Z_RS; }
    else { r.i0 = Z_PS; r.c0 = d.dt; r.i1 = Z_PS; r.i2 = Z_PS; }
    return r;
}

// kinematics for the stand-alone FK kernel (registers; the solver uses the LDS phases of stage_eval)
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

// ------------------------------------------------------------------------------------------
// stage evaluation shared by every pass: short lane-parallel phases, all results in LDS
// ------------------------------------------------------------------------------------------
// rc[] layout (per-stage context in LDS)
constexpr int RC_POSE = 0, RC_PROJ = 6, RC_PROJN = 9, RC_GS = 12, RC_GSN = 30, RC_PHI = 48,
              RC_UB = 49, RC_LB = 52, RC_UBN = 55, RC_LBN = 58, RC_DPP = 61, RC_PHIEND = 64,
              RC_TZ = 65 /*z1,z2*/, RC_BP1 = 67, RC_BP2 = 70, RC_DEP = 73, RC_PEND = 82,
              RC_SL = 85 /*sl0+d (6)*/, RC_FVAL = 91, RC_V = 92 /*v6*/, RC_PROT = 98, RC_SEG = 101 /*s,n*/,
              RC_ER = 103, RC_EP = 106, RC_SIG = 109, RC_DSIG = 110, RC_DPHI = 111, RC_VO = 112 /*6*/,
              RC_DPSI = 118, RC_DDPSI = 119, RC_ER2EP2 = 120, RC_DWVO = 121, RC_DER = 122 /*18*/,
              RC_GSR = 140 /*raw gs 18*/;
// kin[] layout
constexpr int RB_XYZ = 0, RB_ROT = 21, RB_EE = 84, RB_L4C = 87;
constexpr int KN_O = 0 /*7x3*/, KN_PEE = 21, KN_CS = 24 /*cos7,sin7*/, KN_SUFC = 38 /*8x3*/, KN_SUFZ = 62,
              KN_PREZ = 86, KN_R1 = 110 /*3x6*/, KN_R2 = 128, KN_G12 = 146 /*12*/, KN_END = 158;

BMPC_INL double nat_from_zeta(const LDSD* z, int i, const DynC d) {
    if (i < Z_DQ) return z[i] + d.c3 * z[Z_U + i];
    if (i < Z_DDQ) return z[i] + d.c2 * z[Z_U + i - 7];
    if (i < Z_PI) return z[i] + d.c1 * z[Z_U + i - 14];
    if (i == Z_RS) return z[i] + 0.5 * d.dt * z[Z_DRS];
    if (i == Z_PS) return z[i] + 0.5 * d.dt * z[Z_DPS];
    return z[i];
}

#define TABP(off, seg, c) sp[(off) + (c) * 4 + (seg)]

// Returns the stage cost value (same on every lane).  Publishes to LDS: yz, kinematics (J, G, Jp,
// zax, pc), the row context rc[], the output-space cost gradient kin[KN_G12..] and, when want_h,
// the Gauss-Newton/convex cost Hessian blocks Hp/Hv plus the initial group gradients bp0/bpz/bp1/bv.
BMPC_NOINL double stage_eval(const Inst I, LDSD* lds, const DynC dc, int k, int lane, int mode,
                           const double* iw0) {
    // mode 1: evaluate the point (everything but the Hessian blocks); mode 2: Hessian blocks only, the
    // evaluation block [O_yz, O_yz + EVAL_DOUBLES) having been loaded from the cache; mode 3: both
    const bool want_h = (mode & 2) != 0;
    const int N = I.N;
    const LDSD* sp = (lds + O_sp);
    LDSD* rc = (lds + O_rc);
    LDSD* kn = (lds + O_kin);
    const bool term = (k == N - 1);
    if (mode & 1) {
    // ---- E0: natural variables, sin/cos ----
    if (lane < NZ) (lds + O_yz)[lane] = nat_from_zeta((lds + O_zeta), lane, dc);
    if (lane < 7) { double q = nat_from_zeta((lds + O_zeta), lane, dc); kn[KN_CS + lane] = cos(q); kn[KN_CS + 7 + lane] = sin(q); }
    BMPC_SYNC();
    // ---- E1: kinematic chain (every lane, small live set; lane 0 publishes) ----
    {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}, Rn[9], tmp[3];
        const LDSD* rob = (lds + O_rob);
#pragma unroll
        for (int i = 0; i < 7; i++) {
            mat3vec(R, rob + RB_XYZ + 3 * i, tmp);
            for (int a = 0; a < 3; a++) t[a] += tmp[a];
            mat3mul(R, rob + RB_ROT + 9 * i, Rn);
            if (lane == 0)
                for (int a = 0; a < 3; a++) { kn[KN_O + 3 * i + a] = t[a]; (lds + O_zax)[3 * i + a] = Rn[3 * a + 2]; }
            double c = kn[KN_CS + i], s = kn[KN_CS + 7 + i];
            for (int a = 0; a < 3; a++) {
                R[3 * a] = Rn[3 * a] * c + Rn[3 * a + 1] * s;
                R[3 * a + 1] = Rn[3 * a + 1] * c - Rn[3 * a] * s;
                R[3 * a + 2] = Rn[3 * a + 2];
            }
            if (i == 3) {
                mat3vec(R, rob + RB_L4C, tmp);
                if (lane == 0) for (int a = 0; a < 3; a++) (lds + O_pc)[15 + a] = t[a] + tmp[a];
            }
        }
        mat3vec(R, rob + RB_EE, tmp);
        if (lane == 0) for (int a = 0; a < 3; a++) kn[KN_PEE + a] = t[a] + tmp[a];
    }
    BMPC_SYNC();
    // ---- E2: Jacobian columns, collision points and their Jacobians ----
    {
        if (lane < 42) {
            int a = lane / 7, i = lane % 7;
            double v;
            if (a < 3) {
                int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
                double r1 = kn[KN_PEE + a1] - kn[KN_O + 3 * i + a1], r2 = kn[KN_PEE + a2] - kn[KN_O + 3 * i + a2];
                v = (lds + O_zax)[3 * i + a1] * r2 - (lds + O_zax)[3 * i + a2] * r1;
            } else v = (lds + O_zax)[3 * i + a - 3];
            (lds + O_J)[lane] = v;
        } else if (lane < 57) {
            int e = lane - 42;
            (lds + O_pc)[e] = kn[KN_O + 6 + e];     // pc[c] = o[c+2], c < 5
        }
        const int nj[6] = {2, 3, 4, 5, 6, 4};
        for (int e = lane; e < 126; e += BMPC_NT) {
            int c = e / 21, a = (e % 21) / 7, i = e % 7;
            double v = 0;
            if (i < nj[c]) {
                int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
                double p1 = (c < 5) ? kn[KN_O + 3 * (c + 2) + a1] : (lds + O_pc)[15 + a1];
                double p2 = (c < 5) ? kn[KN_O + 3 * (c + 2) + a2] : (lds + O_pc)[15 + a2];
                double r1 = p1 - kn[KN_O + 3 * i + a1], r2 = p2 - kn[KN_O + 3 * i + a2];
                v = (lds + O_zax)[3 * i + a1] * r2 - (lds + O_zax)[3 * i + a2] * r1;
            }
            (lds + O_Jp)[e] = v;
        }
    }
    BMPC_SYNC();
    // ---- E3: v = J dq, prefix/suffix sums for G ----
    if (lane < 6) {
        double s = 0;
        for (int j = 0; j < 7; j++) s += (lds + O_J)[7 * lane + j] * (lds + O_yz)[Z_DQ + j];
        rc[RC_V + lane] = s;
    } else if (lane < 9) {
        int a = lane - 6;
        double sc = 0, sz = 0, pz = 0;
        kn[KN_SUFC + 21 + a] = 0; kn[KN_SUFZ + 21 + a] = 0; kn[KN_PREZ + a] = 0;
        for (int j = 6; j >= 0; j--) {
            sc += (lds + O_J)[7 * a + j] * (lds + O_yz)[Z_DQ + j]; sz += (lds + O_zax)[3 * j + a] * (lds + O_yz)[Z_DQ + j];
            kn[KN_SUFC + 3 * j + a] = sc; kn[KN_SUFZ + 3 * j + a] = sz;
        }
        for (int j = 0; j < 7; j++) { pz += (lds + O_zax)[3 * j + a] * (lds + O_yz)[Z_DQ + j]; kn[KN_PREZ + 3 * (j + 1) + a] = pz; }
    }
    BMPC_SYNC();
    // ---- E4: G = d(J dq)/dq, pose ----
    if (lane < 42) {
        int a = lane / 7, i = lane % 7;
        double v;
        if (a < 3) {
            int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
            // (z_i x sufc_i)[a] + (prez_i x c_i)[a]
            v = (lds + O_zax)[3 * i + a1] * kn[KN_SUFC + 3 * i + a2] - (lds + O_zax)[3 * i + a2] * kn[KN_SUFC + 3 * i + a1];
            v += kn[KN_PREZ + 3 * i + a1] * (lds + O_J)[7 * a2 + i] - kn[KN_PREZ + 3 * i + a2] * (lds + O_J)[7 * a1 + i];
        } else {
            int b = a - 3, a1 = (b + 1) % 3, a2 = (b + 2) % 3;
            v = (lds + O_zax)[3 * i + a1] * kn[KN_SUFZ + 3 * (i + 1) + a2] - (lds + O_zax)[3 * i + a2] * kn[KN_SUFZ + 3 * (i + 1) + a1];
        }
        (lds + O_G)[lane] = v;
    } else if (lane < 48) {
        int a = lane - 42;
        double v = (a < 3) ? kn[KN_PEE + a] : (lds + O_yz)[Z_PI + a - 3] + 0.5 * dc.dt * rc[RC_V + a];
        rc[RC_POSE + a] = v;
        if (a >= 3) rc[RC_PROT + a - 3] = v;
    }
    BMPC_SYNC();
    }
    // ---- E5: reference / error scalars (bound_mpc_functions.py:85-390), every lane ----
    int s = 0;
    if ((double)k > sp[P_SPLIT + 1]) s = 1;
    if ((double)k > sp[P_SPLIT + 2]) s = 2;
    const int n = (sp[P_SPLIT + 1] == (double)N) ? 1 : ((sp[P_SPLIT + 2] == (double)N) ? 2 : 3);
    double fv = 0.0;
    if (!(mode & 1)) fv = rc[RC_FVAL];
    if (mode & 1) {
        const bool iw_param = ((double)k <= sp[P_SPLIT + 1]);
        double dpp[3], dpr[3], d[3], tmp[3], delta[3], er[3], ep[3], jrdpr[3], vv[6];
        for (int a = 0; a < 3; a++) { dpp[a] = TABP(P_DPREF, s, a); dpr[a] = TABP(P_DPREF, s, 3 + a); d[a] = rc[RC_POSE + a] - TABP(P_PREF, s, a); }
        for (int a = 0; a < 6; a++) vv[a] = rc[RC_V + a];
        double phil = dot3(d, dpp);
        double phi = phil + sp[P_PHISW + s];
        double dphi = dot3(vv, dpp);
        for (int a = 0; a < 3; a++) ep[a] = d[a] - dpp[a] * phil;
        for (int a = 0; a < 3; a++) tmp[a] = rc[RC_POSE + 3 + a] - iw0[a];
        for (int a = 0; a < 3; a++) delta[a] = sp[P_JACL + a] * tmp[0] + sp[P_JACL + 3 + a] * tmp[1] + sp[P_JACL + 6 + a] * tmp[2];
        for (int a = 0; a < 3; a++) tmp[a] = dpr[a] * phil + TABP(P_PREF, s, 3 + a) - (iw_param ? sp[P_IWREF + a] : TABP(P_PREF, s, 3 + a));
        for (int a = 0; a < 3; a++) delta[a] -= sp[P_JACR + a] * tmp[0] + sp[P_JACR + 3 + a] * tmp[1] + sp[P_JACR + 6 + a] * tmp[2];
        for (int a = 0; a < 3; a++) {
            er[a] = sp[P_DTAU + 3 * s + a] + delta[a];
            jrdpr[a] = sp[P_JACR + a] * dpr[0] + sp[P_JACR + 3 + a] * dpr[1] + sp[P_JACR + 6 + a] * dpr[2];
        }
        double br1[3], br2[3], dpn[3], v1[3], v2[3], v3[3];
        for (int a = 0; a < 3; a++) {
            br1[a] = TABP(P_BR1, s, a); br2[a] = TABP(P_BR2, s, a); dpn[a] = TABP(P_DPN, s, a);
            v1[a] = TABP(P_V1, s, a); v2[a] = TABP(P_V2, s, a); v3[a] = TABP(P_V3, s, a);
        }
        double sc1 = dot3(delta, v1), scp = dot3(delta, v2), sc2 = dot3(delta, v3);
        double eo1[3], epar[3], eo2[3];
        for (int a = 0; a < 3; a++) {
            eo1[a] = sp[P_DTAU_O1 + 3 * s + a] + sc1 * br1[a];
            epar[a] = sp[P_DTAU_PAR + 3 * s + a] + scp * dpn[a];
            eo2[a] = sp[P_DTAU_O2 + 3 * s + a] + sc2 * br2[a];
        }
        double proj[3] = {dot3(br1, eo1), dot3(dpn, epar), dot3(br2, eo2)};
        double br1n[3], br2n[3], dpnn[3];
        for (int a = 0; a < 3; a++) { br1n[a] = TABP(P_BR1, s + 1, a); br2n[a] = TABP(P_BR2, s + 1, a); dpnn[a] = TABP(P_DPN, s + 1, a); }
        double projn[3] = {dot3(br1n, eo1), dot3(dpnn, epar), dot3(br2n, eo2)};
        double e = exp(-60.0 * (phi - (sp[P_PHIMAX] - 0.05)));
        double sig = 1.0 / (1.0 + e), dsig = 60.0 * sig * (1.0 - sig);
        // stage cost value (ocp :268-299, 360; objective_function :393-428)
        const LDSD* wts = sp + P_W;
        double er2 = dot3(er, er), ep2 = dot3(ep, ep);
        double vo[6], dWvo = 0;
        for (int a = 0; a < 6; a++) { vo[a] = vv[a] - dphi * TABP(P_DPREF, s, a); dWvo += TABP(P_DPREF, s, a) * (a < 3 ? wts[2] : wts[3]) * vo[a]; }
        double dphid = sp[P_XPHID] - phi;
        double rt = sqrt(dphid * dphid + 0.01);
        fv = sig * sig * (er2 + ep2) + wts[1] * dot3(epar, epar);
        fv += wts[2] * (vo[0] * vo[0] + vo[1] * vo[1] + vo[2] * vo[2]) + wts[3] * (vo[3] * vo[3] + vo[4] * vo[4] + vo[5] * vo[5]);
        fv += wts[4] * (rt - 0.1) + wts[5] * (sp[P_XPHID + 1] - dphi) * (sp[P_XPHID + 1] - dphi);
        fv += wts[0] * ep2 + wts[1] / 50.0 * (dot3(eo1, eo1) + dot3(eo2, eo2));
        if (term)
            for (int a = 0; a < 6; a++) fv += 100.0 * vv[a] * vv[a];
        for (int j = 2; j <= 4; j++) fv += wts[6] * (lds + O_yz)[Z_DQ + j] * (lds + O_yz)[Z_DQ + j];
        for (int j = 0; j < 7; j++) fv += wts[7] * (lds + O_yz)[Z_U + j] * (lds + O_yz)[Z_U + j];
        fv += wts[9] * (lds + O_yz)[Z_RS] * (lds + O_yz)[Z_RS] + wts[10] * (lds + O_yz)[Z_DRS] * (lds + O_yz)[Z_DRS] +
              wts[9] * (lds + O_yz)[Z_PS] * (lds + O_yz)[Z_PS] + wts[10] * (lds + O_yz)[Z_DPS] * (lds + O_yz)[Z_DPS];
        if (term)
            for (int i = 0; i < 6; i++) {
                double sl = sp[P_SLACKS0 + i] + (lds + O_yz)[Z_D + i];
                if (i != 4) fv += wts[8] * sl * sl;
                fv += wts[10] * (lds + O_yz)[Z_D + i] * (lds + O_yz)[Z_D + i];
            }
        if (lane == 0) {
            double bp1[3], bp2[3];
            for (int m = 0; m < 3; m++) {
                bp1[m] = TABP(P_BP1, s, m); bp2[m] = TABP(P_BP2, s, m);
                rc[RC_PROJ + m] = proj[m]; rc[RC_PROJN + m] = projn[m];
                rc[RC_UB + m] = TABP(P_ERB, s, m); rc[RC_LB + m] = TABP(P_ERB, s, 3 + m);
                rc[RC_UBN + m] = TABP(P_ERB, s + 1, m); rc[RC_LBN + m] = TABP(P_ERB, s + 1, 3 + m);
                rc[RC_DPP + m] = dpp[m]; rc[RC_BP1 + m] = bp1[m]; rc[RC_BP2 + m] = bp2[m];
                rc[RC_PEND + m] = TABP(P_PREF, s + 1, m);
                rc[RC_ER + m] = er[m]; rc[RC_EP + m] = ep[m];
            }
            rc[RC_PHI] = phi; rc[RC_PHIEND] = sp[P_PHISW + n];
            rc[RC_TZ] = dot3(bp1, ep); rc[RC_TZ + 1] = dot3(bp2, ep);
            for (int i = 0; i < 6; i++) { rc[RC_SL + i] = sp[P_SLACKS0 + i] + (lds + O_yz)[Z_D + i]; rc[RC_VO + i] = vo[i]; }
            rc[RC_FVAL] = fv; rc[RC_SEG] = (double)s; rc[RC_SEG + 1] = (double)n;
            rc[RC_SIG] = sig; rc[RC_DSIG] = dsig; rc[RC_DPHI] = dphi;
            rc[RC_DPSI] = -wts[4] * dphid / rt; rc[RC_DDPSI] = wts[4] * 0.01 / (rt * rt * rt);
            rc[RC_ER2EP2] = er2 + ep2; rc[RC_DWVO] = dWvo;
        }
        // ---- E6 (same phase, disjoint outputs): Dep, Der, raw/scaled gs ----
        if (lane < 9) { int a = lane / 3, b = lane % 3; rc[RC_DEP + lane] = (a == b ? 1.0 : 0.0) - dpp[a] * dpp[b]; }
        else if (lane < 27) {
            int e = lane - 9, a = e / 6, b = e % 6;
            rc[RC_DER + e] = (b < 3) ? -jrdpr[a] * dpp[b] : sp[P_JACL + 3 * (b - 3) + a];
        } else if (lane < 45) {
            int e = lane - 27, m = e / 6, b = e % 6;
            const double* vm = (m == 0) ? v1 : (m == 1 ? v2 : v3);
            double g;
            if (b < 3) g = -dot3(vm, jrdpr) * dpp[b];
            else { int bb = b - 3; g = sp[P_JACL + 3 * bb] * vm[0] + sp[P_JACL + 3 * bb + 1] * vm[1] + sp[P_JACL + 3 * bb + 2] * vm[2]; }
            double nb = (m == 0) ? dot3(br1, br1) : (m == 1 ? dot3(dpn, dpn) : dot3(br2, br2));
            double cc = (m == 0) ? dot3(br1n, br1) : (m == 1 ? dot3(dpnn, dpn) : dot3(br2n, br2));
            rc[RC_GSR + e] = g; rc[RC_GS + e] = nb * g; rc[RC_GSN + e] = cc * g;
        }
    }
    BMPC_SYNC();
    // ---- E7: output-space cost gradient g12 and the residual Jacobians R1, R2 ----
    {
        const LDSD* wts = sp + P_W;
        const double sig = rc[RC_SIG], dsig = rc[RC_DSIG];
        if (lane < 6) {
            int b = lane;
            double s1 = 0;
            for (int a = 0; a < 3; a++) s1 += rc[RC_DER + 6 * a + b] * rc[RC_ER + a];
            double gp = 2 * sig * sig * s1;
            if (b < 3) {
                double s2 = 0;
                for (int a = 0; a < 3; a++) s2 += rc[RC_DEP + 3 * a + b] * rc[RC_EP + a];
                gp += 2 * (sig * sig + wts[0]) * s2 + (2 * sig * dsig * rc[RC_ER2EP2] + rc[RC_DPSI]) * rc[RC_DPP + b];
            }
            gp += 2 * wts[1] * rc[RC_PROJ + 1] * rc[RC_GSR + 6 + b] +
                  2 * (wts[1] / 50.0) * (rc[RC_PROJ] * rc[RC_GSR + b] + rc[RC_PROJ + 2] * rc[RC_GSR + 12 + b]);
            kn[KN_G12 + b] = gp;
        } else if (lane < 12) {
            int b = lane - 6;
            double gv = 2 * (b < 3 ? wts[2] : wts[3]) * rc[RC_VO + b];
            if (b < 3) gv += (-2 * rc[RC_DWVO] - 2 * wts[5] * (sp[P_XPHID + 1] - rc[RC_DPHI])) * rc[RC_DPP + b];
            if (term) gv += 200.0 * rc[RC_V + b];
            kn[KN_G12 + 6 + b] = gv;
        } else if (want_h && lane < 48) {
            int e = lane - 12, which = e / 18, a = (e % 18) / 6, b = e % 6;
            double dphib = (b < 3) ? rc[RC_DPP + b] : 0.0, v;
            if (which == 0) v = sig * rc[RC_DER + 6 * a + b] + rc[RC_ER + a] * dsig * dphib;
            else v = (b < 3 ? sig * rc[RC_DEP + 3 * a + b] : 0.0) + rc[RC_EP + a] * dsig * dphib;
            kn[(which ? KN_R2 : KN_R1) + 6 * a + b] = v;
        }
    }
    BMPC_SYNC();
    // ---- E8: Hessian blocks Hp, Hv and initial group gradients ----
    if (want_h) {
        const LDSD* wts = sp + P_W;
        double w_vp = wts[2], w_vr = wts[3];
        for (int e = lane; e < 72; e += BMPC_NT) {
            if (e < 36) {
                int i = e / 6, j = e % 6;
                double h = 0;
                for (int a = 0; a < 3; a++) h += kn[KN_R1 + 6 * a + i] * kn[KN_R1 + 6 * a + j] + kn[KN_R2 + 6 * a + i] * kn[KN_R2 + 6 * a + j];
                h *= 2;
                // RC_GS = |b|^2 gs: 2 w |b|^2 gs_i gs_j = 2 w RC_GS_i * raw_j
                h += 2 * wts[1] * rc[RC_GS + 6 + i] * rc[RC_GSR + 6 + j];
                h += 2 * (wts[1] / 50.0) * (rc[RC_GS + i] * rc[RC_GSR + j] + rc[RC_GS + 12 + i] * rc[RC_GSR + 12 + j]);
                if (i < 3 && j < 3) {
                    double dd = 0;
                    for (int a = 0; a < 3; a++) dd += rc[RC_DEP + 3 * a + i] * rc[RC_DEP + 3 * a + j];
                    h += 2 * wts[0] * dd + rc[RC_DDPSI] * rc[RC_DPP + i] * rc[RC_DPP + j];
                }
                (lds + O_Hp)[e] = h;
            } else {
                int i = (e - 36) / 6, j = (e - 36) % 6;
                double dWd = 0;
                for (int a = 0; a < 6; a++) { double da = TABP(P_DPREF, s, a); dWd += da * da * (a < 3 ? w_vp : w_vr); }
                double wi = (i < 3 ? w_vp : w_vr), wj = (j < 3 ? w_vp : w_vr);
                double di = (i < 3) ? rc[RC_DPP + i] : 0.0, dj = (j < 3) ? rc[RC_DPP + j] : 0.0;
                double h = (i == j ? wi : 0.0) - wi * TABP(P_DPREF, s, i) * dj - di * wj * TABP(P_DPREF, s, j) + di * dj * dWd;
                h = 2 * h + 2 * wts[5] * di * dj;
                if (term && i == j) h += 200.0;
                (lds + O_Hv)[e - 36] = h;
            }
        }
        if (lane < 6) { double g = kn[KN_G12 + lane]; (lds + O_bp0)[lane] = g; (lds + O_bpz)[lane] = g; (lds + O_bp1)[lane] = 0; (lds + O_bv)[lane] = kn[KN_G12 + 6 + lane]; }
    }
    BMPC_SYNC();
    return fv;
}

// One inequality row (slot) of stage k: h value, and a compact description of its gradient.
// kind: 0 inactive, 1 natural-diagonal (pos, coef), 2 zeta-diagonal (pos, coef), 3 pose row
// (a6 + slack selector 0 none / 1 ps / 2 rs / 3 d5, coefficient -1), 4 point row (point c, a3, -d_c)
struct Row { int kind, pos, sel; double coef, h, a[6]; };

BMPC_NOINL void row_eval(const Inst I, LDSD* lds, int k, int s, Row& r) {
    const int N = I.N;
    const LDSD* sp = (lds + O_sp);
    const double* pg = I.pg;   // EE sets live in global memory
    const LDSD* rc = (lds + O_rc);
    r.kind = 0; r.pos = 0; r.sel = 0; r.coef = 0; r.h = 0;
    for (int c = 0; c < 6; c++) r.a[c] = 0;
    if (s < S_NONNEG) {                       // box bounds on q,dq,ddq,u (BoundMPC.py:171-186,544-589)
        int j = s >> 1, blk = j / 7, jj = j - 7 * blk;
        int pos = (blk == 0 ? Z_Q : blk == 1 ? Z_DQ : blk == 2 ? Z_DDQ : Z_U) + jj;
        size_t wi = (size_t)blk * 7 * N + (size_t)jj * N + k;
        if ((s & 1) == 0) {
            double ub = I.ubx[wi];
            if (ub < BIGB) { r.kind = 1; r.pos = pos; r.coef = 1.0; r.h = (lds + O_yz)[pos] - ub; }
        } else {
            double lb = I.lbx[wi];
            if (lb > -BIGB) { r.kind = 1; r.pos = pos; r.coef = -1.0; r.h = lb - (lds + O_yz)[pos]; }
        }
    } else if (s < S_RS1) {                   // rs, drs, ps, dps >= 0 (Q6)
        int m = s - S_NONNEG;
        int pos = (m == 0 ? Z_RS : m == 1 ? Z_DRS : m == 2 ? Z_PS : Z_DPS);
        r.kind = 1; r.pos = pos; r.coef = -1.0; r.h = -(lds + O_yz)[pos];
    } else if (s < S_D1) {                    // stage-0 slacks: rs~_1, ps~_1 >= 0
        if (k == 1) { int pos = (s == S_RS1) ? Z_RS : Z_PS; r.kind = 2; r.pos = pos; r.coef = -1.0; r.h = -(lds + O_zeta)[pos]; }
    } else if (s < S_EE) {                    // dslacks >= 0
        if (k == 1) { int pos = Z_D + (s - S_D1); r.kind = 1; r.pos = pos; r.coef = -1.0; r.h = -(lds + O_yz)[pos]; }
    } else if (s < S_ROTU) {                  // EE in current set (ocp :304)
        int rr = s - S_EE, sg = (int)rc[RC_SEG];
        const double* a = pg + P_ASET + 45 * sg;
        double a0 = a[rr], a1 = a[rr + 15], a2 = a[rr + 30], bb = pg[P_BSET + rr * 4 + sg];
        if (!(a0 == 0 && a1 == 0 && a2 == 0 && bb > 0)) {
            r.kind = 3; r.sel = 1; r.a[0] = a0; r.a[1] = a1; r.a[2] = a2;
            r.h = a0 * rc[RC_POSE] + a1 * rc[RC_POSE + 1] + a2 * rc[RC_POSE + 2] - bb - (lds + O_yz)[Z_PS];
        }
    } else if (s < S_COL) {                   // orientation bounds (ocp :308-321)
        int m = s - S_ROTU;
        bool lower = m >= 3;
        if (lower) m -= 3;
        r.kind = 3; r.sel = 2;
        double sgn = lower ? -1.0 : 1.0;
        for (int c = 0; c < 6; c++) r.a[c] = sgn * rc[RC_GS + 6 * m + c];
        r.h = lower ? -(rc[RC_PROJ + m] - rc[RC_LB + m] + (lds + O_yz)[Z_RS]) : (rc[RC_PROJ + m] - rc[RC_UB + m] - (lds + O_yz)[Z_RS]);
    } else if (s < S_PHI) {                   // collision points (ocp :323-330)
        int c = (s - S_COL) / 15, rr = (s - S_COL) - 15 * c;
        const LDSD* a = sp + SP_ASETJ + 45 * c;
        double a0 = a[rr], a1 = a[rr + 15], a2 = a[rr + 30], bb = sp[SP_BSETJ + rr * 6 + c];
        if (!(a0 == 0 && a1 == 0 && a2 == 0 && bb + sp[P_SLACKS0 + c] > 0)) {
            r.kind = 4; r.pos = c; r.a[0] = a0; r.a[1] = a1; r.a[2] = a2;
            r.h = a0 * (lds + O_pc)[3 * c] + a1 * (lds + O_pc)[3 * c + 1] + a2 * (lds + O_pc)[3 * c + 2] - bb - rc[RC_SL + c];
        }
    } else if (s == S_PHI) {                  // phi cap (ocp :332)
        r.kind = 3; r.sel = 0;
        for (int c = 0; c < 3; c++) r.a[c] = rc[RC_DPP + c];
        r.h = rc[RC_PHI] - (rc[RC_PHIEND] + 0.005);
    } else if (s < S_TROTU) {                 // terminal next-set rows (ocp :346-358)
        if (k == N - 1) {
            int rr = s - S_TSET, nn = (int)rc[RC_SEG + 1];
            const double* a = pg + P_ASET + 45 * nn;
            double an[3] = {a[rr], a[rr + 15], a[rr + 30]};
            double bn = pg[P_BSET + rr * 4 + nn];
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
