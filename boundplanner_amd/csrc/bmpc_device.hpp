// bmpc_device.hpp -- constants, argument structs and small device helpers shared by the kernels of the batched BoundMPC
// interior-point solver (gfx950): the receding-horizon NLP that /root/reference/bound_planner/BoundMPC/BoundMPC.py:594-603
// hands to CasADi/IPOPT (formulation casadi_ocp_formulation.py:13-421).
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

#ifndef BMPC_LDW
#define BMPC_LDW 42
#endif
constexpr int NX = 32, NU = 9, NZ = 41, LDW = BMPC_LDW;      // LDW: row stride of the stage matrix in LDS (even: 16-byte row reads; odd: no bank conflicts on column walks)
// the value-function Hessian P (32 x 32, symmetric) is held packed in LDS: lower triangle by rows (528 doubles instead of 32 x 33)
constexpr int NPSYM = NX * (NX + 1) / 2;
constexpr int NSLOT = 208;      // inequality-row slots per stage
constexpr int ZPAD = 48;        // padded stage vector length
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
// per-instance block of the staged parameter vectors in LDS (stage_params): the NPAR parameters, then the activity masks of the
// 15-row halfspace sets (bit rr set: row rr is a constraint; padding rows a = 0, b > 0 are none): six collision-point sets, four EE sets
constexpr int PL_MASKJ = NPAR, PL_MASKE = NPAR + 6, NPARL = NPAR + 10;
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
    double mu_floor_k, dw0, inertia_err, ls_alpha_mem;     // include/boundmpc.h bmpc_opts
    int inertia, stall_n, gn_backoff, slack_reset;
    int trial_repeats;   // a rejected line-search trial is repeated (half the step length) up to this many times inside k_trial
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

// kinematics for the stand-alone FK kernel and the closed-loop kernels (the solver kernels have their own: bmpc_stage.hpp)
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
        double c, s;
        BMPC_SINCOS(q[i], s, c);
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

}  // namespace bmpc
