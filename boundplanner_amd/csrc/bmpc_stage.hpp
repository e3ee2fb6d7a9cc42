// bmpc_stage.hpp -- thread-serial stage functions of the batched BoundMPC interior-point solver.
//
// One THREAD evaluates one (instance, stage) pair: kinematic chain of the iiwa14
// (RobotModel.py:146-231 restated from iiwa.urdf), reference/error decomposition
// (bound_mpc_functions.py:85-390), stage cost (casadi_ocp_formulation.py:268-299, 360;
// bound_mpc_functions.py:393-428) and the inequality rows (casadi_ocp_formulation.py:304-380,
// BoundMPC.py:544-589).  Every array index is a compile-time constant after unrolling, so the
// working set lives in registers; parameters are read from global memory (threads of one instance
// read the same address: one request).  The same source compiles on the host for tests/emu.
//
// Algebra and notation: oracle/bmpc_solve.c and bmpc_device.hpp (zeta / natural coordinates).
#pragma once
#include "bmpc_device.hpp"

namespace bmpc {

#define BMPC_UNROLL _Pragma("unroll")
typedef const LDSD* PGP;       // parameter vector of the thread's instance, staged in LDS (bmpc_pipeline.hpp)

// ------------------------------------------------------------------------------------------
// kinematics
// ------------------------------------------------------------------------------------------
struct KinT {
    double o[7][3];    // joint origins
    double zx[7][3];   // joint axes
    double pee[3];     // end effector position
    double pl4[3];     // link4_col_link position (collision point 5)
};

BMPC_INL void kin_chain(GRC rc, const double* q, KinT& K) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    BMPC_UNROLL
    for (int i = 0; i < 7; i++) {
        double Rn[9];
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) t[a] += R[3 * a] * rc->jxyz[i][0] + R[3 * a + 1] * rc->jxyz[i][1] + R[3 * a + 2] * rc->jxyz[i][2];
        BMPC_UNROLL
        for (int a = 0; a < 3; a++)
            BMPC_UNROLL
            for (int b = 0; b < 3; b++)
                Rn[3 * a + b] = R[3 * a] * rc->jrot[i][b] + R[3 * a + 1] * rc->jrot[i][3 + b] + R[3 * a + 2] * rc->jrot[i][6 + b];
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) { K.o[i][a] = t[a]; K.zx[i][a] = Rn[3 * a + 2]; }
        double c, s;
        BMPC_SINCOS(q[i], s, c);
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) {
            R[3 * a] = Rn[3 * a] * c + Rn[3 * a + 1] * s;
            R[3 * a + 1] = Rn[3 * a + 1] * c - Rn[3 * a] * s;
            R[3 * a + 2] = Rn[3 * a + 2];
        }
        if (i == 3) {
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) K.pl4[a] = t[a] + R[3 * a] * rc->l4c_xyz[0] + R[3 * a + 1] * rc->l4c_xyz[1] + R[3 * a + 2] * rc->l4c_xyz[2];
        }
    }
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) K.pee[a] = t[a] + R[3 * a] * rc->ee_xyz[0] + R[3 * a + 1] * rc->ee_xyz[1] + R[3 * a + 2] * rc->ee_xyz[2];
}

// collision point c (RobotModel.py:27-35): joint_3..joint_7 origins, link4_col_link
template <int C> BMPC_INL const double* kin_point(const KinT& K) { return C < 5 ? K.o[C + 2] : K.pl4; }
template <int C> struct PointNJ { static constexpr int value = (C == 0 ? 2 : C == 1 ? 3 : C == 2 ? 4 : C == 3 ? 5 : C == 4 ? 6 : 4); };

BMPC_INL void cross3r(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// linear Jacobian columns Jl[a][i] = (z_i x (pee - o_i))[a]; v = J dq
BMPC_INL void kin_jlin(const KinT& K, double Jl[3][7]) {
    BMPC_UNROLL
    for (int i = 0; i < 7; i++) {
        double r[3] = {K.pee[0] - K.o[i][0], K.pee[1] - K.o[i][1], K.pee[2] - K.o[i][2]}, c[3];
        cross3r(K.zx[i], r, c);
        Jl[0][i] = c[0]; Jl[1][i] = c[1]; Jl[2][i] = c[2];
    }
}
BMPC_INL void kin_vel(const KinT& K, const double Jl[3][7], const double* dq, double* v) {
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) {
        double s = 0, w = 0;
        BMPC_UNROLL
        for (int j = 0; j < 7; j++) { s += Jl[a][j] * dq[j]; w += K.zx[j][a] * dq[j]; }
        v[a] = s; v[3 + a] = w;
    }
}
// G = d(J dq)/dq (6x7)
BMPC_INL void kin_G(const KinT& K, const double Jl[3][7], const double* dq, double G[6][7]) {
    double sufc[8][3], sufz[8][3], prez[8][3];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) { sufc[7][a] = 0; sufz[7][a] = 0; prez[0][a] = 0; }
    BMPC_UNROLL
    for (int j = 6; j >= 0; j--)
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) {
            sufc[j][a] = sufc[j + 1][a] + Jl[a][j] * dq[j];
            sufz[j][a] = sufz[j + 1][a] + K.zx[j][a] * dq[j];
        }
    BMPC_UNROLL
    for (int j = 0; j < 7; j++)
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) prez[j + 1][a] = prez[j][a] + K.zx[j][a] * dq[j];
    BMPC_UNROLL
    for (int i = 0; i < 7; i++) {
        double ci[3] = {Jl[0][i], Jl[1][i], Jl[2][i]}, t1[3], t2[3], t3[3];
        cross3r(K.zx[i], sufc[i], t1);
        cross3r(prez[i], ci, t2);
        cross3r(K.zx[i], sufz[i + 1], t3);
        BMPC_UNROLL
        for (int a = 0; a < 3; a++) { G[a][i] = t1[a] + t2[a]; G[3 + a][i] = t3[a]; }
    }
}
// point Jacobian column i of collision point C: (z_i x (pc - o_i)), zero for i >= nj
template <int C, int I> BMPC_INL void kin_pjcol(const KinT& K, double* c) {
    if (I < PointNJ<C>::value) {
        const double* pc = kin_point<C>(K);
        double r[3] = {pc[0] - K.o[I][0], pc[1] - K.o[I][1], pc[2] - K.o[I][2]};
        cross3r(K.zx[I], r, c);
    } else { c[0] = 0; c[1] = 0; c[2] = 0; }
}

// ------------------------------------------------------------------------------------------
// natural <-> zeta
// ------------------------------------------------------------------------------------------
BMPC_INL void nat_all(const double* z, const DynC d, double* y) {
    BMPC_UNROLL
    for (int i = 0; i < 7; i++) {
        y[Z_Q + i] = z[Z_Q + i] + d.c3 * z[Z_U + i];
        y[Z_DQ + i] = z[Z_DQ + i] + d.c2 * z[Z_U + i];
        y[Z_DDQ + i] = z[Z_DDQ + i] + d.c1 * z[Z_U + i];
        y[Z_U + i] = z[Z_U + i];
    }
    BMPC_UNROLL
    for (int i = 0; i < 3; i++) y[Z_PI + i] = z[Z_PI + i];
    y[Z_RS] = z[Z_RS] + 0.5 * d.dt * z[Z_DRS];
    y[Z_PS] = z[Z_PS] + 0.5 * d.dt * z[Z_DPS];
    BMPC_UNROLL
    for (int i = 0; i < 6; i++) y[Z_D + i] = z[Z_D + i];
    y[Z_DRS] = z[Z_DRS]; y[Z_DPS] = z[Z_DPS];
}

BMPC_INL DynC make_dync(double dt) {
    DynC dc;
    dc.dt = dt; dc.c1 = dt / 2; dc.c2 = dt * dt / 6; dc.c3 = dt * dt * dt / 24;
    dc.b1 = dt; dc.b2 = dt * dt; dc.b3 = 7 * dt * dt * dt / 12;
    return dc;
}

// required pinned part of x_1 (q~, dq~, ddq~, pi) from the stage-0 pins (BoundMPC.py:551-556)
BMPC_INL void x1fix_eval(GCD lbx, int N, double dt, double* x1fix /*24*/) {
    BMPC_UNROLL
    for (int j = 0; j < 7; j++) {
        double q0 = lbx[j * N], dq0 = lbx[7 * N + j * N], ddq0 = lbx[14 * N + j * N], u0 = lbx[21 * N + j * N];
        x1fix[Z_Q + j] = q0 + dt * dq0 + dt * dt / 2 * ddq0 + dt * dt * dt / 8 * u0;
        x1fix[Z_DQ + j] = dq0 + dt * ddq0 + dt * dt / 3 * u0;
        x1fix[Z_DDQ + j] = ddq0 + dt / 2 * u0;
    }
    BMPC_UNROLL
    for (int c = 0; c < 3; c++) x1fix[Z_PI + c] = lbx[28 * N + (3 + c) * N] + dt / 2 * lbx[34 * N + (3 + c) * N];
}

// ------------------------------------------------------------------------------------------
// reference / error context at one stage (bound_mpc_functions.py:85-390)
// ------------------------------------------------------------------------------------------
struct SegCtx {
    int s, n;                    // current segment, "next" selector for phi_end
    double pose[6], v[6];
    double dpp[3], phi, phiend, dphi, sig, dsig;
    double er[3], ep[3], proj[3], projn[3], ub[3], lb[3], ubn[3], lbn[3];
    double gs[3][6], gsn[3][6], gsr[3][6];
    double Dep[3][3], Der[3][6];
    double bp1[3], bp2[3], pend[3], tz[2];
    double sl[6];
    double vo[6], dpsi, ddpsi, er2ep2, dWvo, fv;
};

#define PTAB(off, seg, c) pg[(off) + (c) * 4 + (seg)]

// pose = [p_ee, p_rot], v = J dq; y = natural stage variables; iw0 = pinned stage-0 p_rot
BMPC_INL void seg_ctx_eval(PGP pg, int N, int k, const double* y, const double* iw0, SegCtx& C) {
    const bool term = (k == N - 1);
    int s = 0;
    if ((double)k > pg[P_SPLIT + 1]) s = 1;
    if ((double)k > pg[P_SPLIT + 2]) s = 2;
    const int n = (pg[P_SPLIT + 1] == (double)N) ? 1 : ((pg[P_SPLIT + 2] == (double)N) ? 2 : 3);
    C.s = s; C.n = n;
    const bool iw_param = ((double)k <= pg[P_SPLIT + 1]);
    PGP wts = pg + P_W;
    double dpr[3], d[3], tmp[3], delta[3], jrdpr[3];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) { C.dpp[a] = PTAB(P_DPREF, s, a); dpr[a] = PTAB(P_DPREF, s, 3 + a); d[a] = C.pose[a] - PTAB(P_PREF, s, a); }
    double phil = d[0] * C.dpp[0] + d[1] * C.dpp[1] + d[2] * C.dpp[2];
    C.phi = phil + pg[P_PHISW + s];
    C.dphi = C.v[0] * C.dpp[0] + C.v[1] * C.dpp[1] + C.v[2] * C.dpp[2];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) C.ep[a] = d[a] - C.dpp[a] * phil;
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) tmp[a] = C.pose[3 + a] - iw0[a];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) delta[a] = pg[P_JACL + a] * tmp[0] + pg[P_JACL + 3 + a] * tmp[1] + pg[P_JACL + 6 + a] * tmp[2];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) tmp[a] = dpr[a] * phil + PTAB(P_PREF, s, 3 + a) - (iw_param ? pg[P_IWREF + a] : PTAB(P_PREF, s, 3 + a));
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) delta[a] -= pg[P_JACR + a] * tmp[0] + pg[P_JACR + 3 + a] * tmp[1] + pg[P_JACR + 6 + a] * tmp[2];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) {
        C.er[a] = pg[P_DTAU + 3 * s + a] + delta[a];
        jrdpr[a] = pg[P_JACR + a] * dpr[0] + pg[P_JACR + 3 + a] * dpr[1] + pg[P_JACR + 6 + a] * dpr[2];
    }
    double br1[3], br2[3], dpn[3], v1[3], v2[3], v3[3], br1n[3], br2n[3], dpnn[3];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) {
        br1[a] = PTAB(P_BR1, s, a); br2[a] = PTAB(P_BR2, s, a); dpn[a] = PTAB(P_DPN, s, a);
        v1[a] = PTAB(P_V1, s, a); v2[a] = PTAB(P_V2, s, a); v3[a] = PTAB(P_V3, s, a);
        br1n[a] = PTAB(P_BR1, s + 1, a); br2n[a] = PTAB(P_BR2, s + 1, a); dpnn[a] = PTAB(P_DPN, s + 1, a);
    }
    double sc1 = dot3(delta, v1), scp = dot3(delta, v2), sc2 = dot3(delta, v3);
    double eo1[3], epar[3], eo2[3];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) {
        eo1[a] = pg[P_DTAU_O1 + 3 * s + a] + sc1 * br1[a];
        epar[a] = pg[P_DTAU_PAR + 3 * s + a] + scp * dpn[a];
        eo2[a] = pg[P_DTAU_O2 + 3 * s + a] + sc2 * br2[a];
    }
    C.proj[0] = dot3(br1, eo1); C.proj[1] = dot3(dpn, epar); C.proj[2] = dot3(br2, eo2);
    C.projn[0] = dot3(br1n, eo1); C.projn[1] = dot3(dpnn, epar); C.projn[2] = dot3(br2n, eo2);
    double e = exp(-60.0 * (C.phi - (pg[P_PHIMAX] - 0.05)));
    C.sig = 1.0 / (1.0 + e); C.dsig = 60.0 * C.sig * (1.0 - C.sig);
    double er2 = dot3(C.er, C.er), ep2 = dot3(C.ep, C.ep);
    C.dWvo = 0;
    BMPC_UNROLL
    for (int a = 0; a < 6; a++) {
        double da = PTAB(P_DPREF, s, a);
        C.vo[a] = C.v[a] - C.dphi * da;
        C.dWvo += da * (a < 3 ? wts[2] : wts[3]) * C.vo[a];
    }
    double dphid = pg[P_XPHID] - C.phi;
    double rt = sqrt(dphid * dphid + 0.01);
    double fv = C.sig * C.sig * (er2 + ep2) + wts[1] * dot3(epar, epar);
    fv += wts[2] * (C.vo[0] * C.vo[0] + C.vo[1] * C.vo[1] + C.vo[2] * C.vo[2]) + wts[3] * (C.vo[3] * C.vo[3] + C.vo[4] * C.vo[4] + C.vo[5] * C.vo[5]);
    fv += wts[4] * (rt - 0.1) + wts[5] * (pg[P_XPHID + 1] - C.dphi) * (pg[P_XPHID + 1] - C.dphi);
    fv += wts[0] * ep2 + wts[1] / 50.0 * (dot3(eo1, eo1) + dot3(eo2, eo2));
    if (term)
        BMPC_UNROLL
        for (int a = 0; a < 6; a++) fv += 100.0 * C.v[a] * C.v[a];
    BMPC_UNROLL
    for (int j = 2; j <= 4; j++) fv += wts[6] * y[Z_DQ + j] * y[Z_DQ + j];
    BMPC_UNROLL
    for (int j = 0; j < 7; j++) fv += wts[7] * y[Z_U + j] * y[Z_U + j];
    fv += wts[9] * y[Z_RS] * y[Z_RS] + wts[10] * y[Z_DRS] * y[Z_DRS] + wts[9] * y[Z_PS] * y[Z_PS] + wts[10] * y[Z_DPS] * y[Z_DPS];
    if (term)
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) {
            double sl = pg[P_SLACKS0 + i] + y[Z_D + i];
            if (i != 4) fv += wts[8] * sl * sl;
            fv += wts[10] * y[Z_D + i] * y[Z_D + i];
        }
    C.fv = fv;
    BMPC_UNROLL
    for (int m = 0; m < 3; m++) {
        C.bp1[m] = PTAB(P_BP1, s, m); C.bp2[m] = PTAB(P_BP2, s, m);
        C.ub[m] = PTAB(P_ERB, s, m); C.lb[m] = PTAB(P_ERB, s, 3 + m);
        C.ubn[m] = PTAB(P_ERB, s + 1, m); C.lbn[m] = PTAB(P_ERB, s + 1, 3 + m);
        C.pend[m] = PTAB(P_PREF, s + 1, m);
    }
    C.phiend = pg[P_PHISW + n];
    C.tz[0] = dot3(C.bp1, C.ep); C.tz[1] = dot3(C.bp2, C.ep);
    BMPC_UNROLL
    for (int i = 0; i < 6; i++) C.sl[i] = pg[P_SLACKS0 + i] + y[Z_D + i];
    C.dpsi = -wts[4] * dphid / rt; C.ddpsi = wts[4] * 0.01 / (rt * rt * rt);
    C.er2ep2 = er2 + ep2;
    BMPC_UNROLL
    for (int a = 0; a < 3; a++)
        BMPC_UNROLL
        for (int b = 0; b < 3; b++) C.Dep[a][b] = (a == b ? 1.0 : 0.0) - C.dpp[a] * C.dpp[b];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++)
        BMPC_UNROLL
        for (int b = 0; b < 6; b++) C.Der[a][b] = (b < 3) ? -jrdpr[a] * C.dpp[b] : pg[P_JACL + 3 * (b - 3) + a];
    double nb[3] = {dot3(br1, br1), dot3(dpn, dpn), dot3(br2, br2)};
    double cc[3] = {dot3(br1n, br1), dot3(dpnn, dpn), dot3(br2n, br2)};
    double vmj[3] = {dot3(v1, jrdpr), dot3(v2, jrdpr), dot3(v3, jrdpr)};
    BMPC_UNROLL
    for (int m = 0; m < 3; m++) {
        const double* vm = (m == 0) ? v1 : (m == 1 ? v2 : v3);
        BMPC_UNROLL
        for (int b = 0; b < 6; b++) {
            double g;
            if (b < 3) g = -vmj[m] * C.dpp[b];
            else g = pg[P_JACL + 3 * (b - 3)] * vm[0] + pg[P_JACL + 3 * (b - 3) + 1] * vm[1] + pg[P_JACL + 3 * (b - 3) + 2] * vm[2];
            C.gsr[m][b] = g; C.gs[m][b] = nb[m] * g; C.gsn[m][b] = cc[m] * g;
        }
    }
}

// output-space cost gradient g12 = d f / d(pose, v) (E7)
BMPC_INL void cost_grad12(PGP pg, const SegCtx& C, bool term, double* g12) {
    PGP wts = pg + P_W;
    const double sig = C.sig, dsig = C.dsig;
    BMPC_UNROLL
    for (int b = 0; b < 6; b++) {
        double s1 = C.Der[0][b] * C.er[0] + C.Der[1][b] * C.er[1] + C.Der[2][b] * C.er[2];
        double gp = 2 * sig * sig * s1;
        if (b < 3) {
            double s2 = C.Dep[0][b] * C.ep[0] + C.Dep[1][b] * C.ep[1] + C.Dep[2][b] * C.ep[2];
            gp += 2 * (sig * sig + wts[0]) * s2 + (2 * sig * dsig * C.er2ep2 + C.dpsi) * C.dpp[b];
        }
        gp += 2 * wts[1] * C.proj[1] * C.gsr[1][b] + 2 * (wts[1] / 50.0) * (C.proj[0] * C.gsr[0][b] + C.proj[2] * C.gsr[2][b]);
        g12[b] = gp;
        double gv = 2 * (b < 3 ? wts[2] : wts[3]) * C.vo[b];
        if (b < 3) gv += (-2 * C.dWvo - 2 * wts[5] * (pg[P_XPHID + 1] - C.dphi)) * C.dpp[b];
        if (term) gv += 200.0 * C.v[b];
        g12[6 + b] = gv;
    }
}

// Gauss-Newton/convex output-space Hessians: Hp (pose x pose, 21 packed upper), Hv (v x v)
BMPC_INL constexpr int sym6(int i, int j) { return i <= j ? (i * 6 - i * (i - 1) / 2 + (j - i)) : (j * 6 - j * (j - 1) / 2 + (i - j)); }

BMPC_INL void cost_hess(PGP pg, const SegCtx& C, bool term, double* Hp /*21*/, double* Hv /*21*/) {
    PGP wts = pg + P_W;
    const double sig = C.sig, dsig = C.dsig;
    double R1[3][6], R2[3][6];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++)
        BMPC_UNROLL
        for (int b = 0; b < 6; b++) {
            double dphib = (b < 3) ? C.dpp[b] : 0.0;
            R1[a][b] = sig * C.Der[a][b] + C.er[a] * dsig * dphib;
            R2[a][b] = (b < 3 ? sig * C.Dep[a][b] : 0.0) + C.ep[a] * dsig * dphib;
        }
    const double w_vp = wts[2], w_vr = wts[3];
    double dWd = 0;
    BMPC_UNROLL
    for (int a = 0; a < 6; a++) { double da = PTAB(P_DPREF, C.s, a); dWd += da * da * (a < 3 ? w_vp : w_vr); }
    BMPC_UNROLL
    for (int i = 0; i < 6; i++)
        BMPC_UNROLL
        for (int j = i; j < 6; j++) {
            double h = 0;
            BMPC_UNROLL
            for (int a = 0; a < 3; a++) h += R1[a][i] * R1[a][j] + R2[a][i] * R2[a][j];
            h *= 2;
            h += 2 * wts[1] * C.gs[1][i] * C.gsr[1][j];
            h += 2 * (wts[1] / 50.0) * (C.gs[0][i] * C.gsr[0][j] + C.gs[2][i] * C.gsr[2][j]);
            if (i < 3 && j < 3) {
                double dd = C.Dep[0][i] * C.Dep[0][j] + C.Dep[1][i] * C.Dep[1][j] + C.Dep[2][i] * C.Dep[2][j];
                h += 2 * wts[0] * dd + C.ddpsi * C.dpp[i] * C.dpp[j];
            }
            Hp[sym6(i, j)] = h;
            double wi = (i < 3 ? w_vp : w_vr), wj = (j < 3 ? w_vp : w_vr);
            double di = (i < 3) ? C.dpp[i] : 0.0, dj = (j < 3) ? C.dpp[j] : 0.0;
            double hv = (i == j ? wi : 0.0) - wi * PTAB(P_DPREF, C.s, i) * dj - di * wj * PTAB(P_DPREF, C.s, j) + di * dj * dWd;
            hv = 2 * hv + 2 * wts[5] * di * dj;
            if (term && i == j) hv += 200.0;
            Hv[sym6(i, j)] = hv;
        }
}

// the v x v block alone: it depends on the parameters (segment of the stage, weights) only -- k_eval recomputes it instead of
// carrying the whole context (same expressions as in cost_hess)
BMPC_INL void cost_hess_v(PGP pg, int N, int k, double* Hv /*21*/) {
    PGP wts = pg + P_W;
    const bool term = (k == N - 1);
    int s = 0;
    if ((double)k > pg[P_SPLIT + 1]) s = 1;
    if ((double)k > pg[P_SPLIT + 2]) s = 2;
    const double w_vp = wts[2], w_vr = wts[3];
    double dpp[3];
    BMPC_UNROLL
    for (int a = 0; a < 3; a++) dpp[a] = PTAB(P_DPREF, s, a);
    double dWd = 0;
    BMPC_UNROLL
    for (int a = 0; a < 6; a++) { double da = PTAB(P_DPREF, s, a); dWd += da * da * (a < 3 ? w_vp : w_vr); }
    BMPC_UNROLL
    for (int i = 0; i < 6; i++)
        BMPC_UNROLL
        for (int j = i; j < 6; j++) {
            double wi = (i < 3 ? w_vp : w_vr), wj = (j < 3 ? w_vp : w_vr);
            double di = (i < 3) ? dpp[i < 3 ? i : 0] : 0.0, dj = (j < 3) ? dpp[j < 3 ? j : 0] : 0.0;
            double hv = (i == j ? wi : 0.0) - wi * PTAB(P_DPREF, s, i) * dj - di * wj * PTAB(P_DPREF, s, j) + di * dj * dWd;
            hv = 2 * hv + 2 * wts[5] * di * dj;
            if (term && i == j) hv += 200.0;
            Hv[sym6(i, j)] = hv;
        }
}

// ------------------------------------------------------------------------------------------
// inequality rows.  Visitor interface: the row walker calls, for every ACTIVE slot s,
//   v.diag(s, pos, coef, h)            natural-diagonal row   h = coef*y[pos] + const
//   v.zdiag(s, pos, coef, h)           zeta-diagonal row (k == 1: rs~_1, ps~_1 >= 0)
//   v.pose(s, a6, sel, h)              pose-space row, slack selector 0 none/1 ps/2 rs/3 d5 (coef -1)
//   v.point<C>(s, a3, h)               collision-point row of point C (slack d_C, coef -1)
// and v.skip(s) for inactive slots.  Slot numbering: bmpc_device.hpp (S_*).
// ------------------------------------------------------------------------------------------
// Row groups of walk_rows: before the rows of a group are walked the visitor's group<S0, CNT>() hook runs (the six point
// groups announce themselves through point_begin<C>()), so that a visitor which needs per-row data from memory can load a
// whole group in one batch.  row_group_base(s) = first slot of the group of slot s.
BMPC_HD constexpr int row_group_base(int s) {
    return s < 28 ? 0 : s < S_NONNEG ? 28 : s < S_EE ? S_NONNEG : s < S_COL ? S_EE : s < S_PHI ? S_COL + 15 * ((s - S_COL) / 15) : S_PHI;
}
constexpr int ROW_GROUP_MAX = 28;

template <class V, int C0, int CEND = 6>
BMPC_INL void walk_points(PGP pg, const KinT& K, const double* sl, V& v) {
    if constexpr (C0 < CEND) {
        // (row rr is a constraint unless a == 0 and b + slacks0 > 0: bit rr of the set's activity mask, stage_masks)
        const unsigned act = (unsigned)pg[PL_MASKJ + C0];
        v.template point_begin<C0>(act);
        PGP a = pg + P_ASETJ + 45 * C0;
        const double* pc = kin_point<C0>(K);
        BMPC_UNROLL
        for (int rr = 0; rr < 15; rr++) {
            if ((act >> rr) & 1u) {
                double a3[3] = {a[rr], a[rr + 15], a[rr + 30]};
                double bb = pg[P_BSETJ + rr * 6 + C0];
                v.template point<C0>(S_COL + 15 * C0 + rr, a3, a3[0] * pc[0] + a3[1] * pc[1] + a3[2] * pc[2] - bb - sl[C0]);
            } else v.skip(S_COL + 15 * C0 + rr);
        }
        v.template point_end<C0>();
        walk_points<V, C0 + 1, CEND>(pg, K, sl, v);
    }
}

// ROLES: which parts of the walk run (the wave-specialised kernels give each wavefront of a workgroup one part: the parts need
// different inputs -- natural coordinates only / reference context / collision-point positions -- so no wavefront holds them all)
constexpr int WR_BOX = 1, WR_POSE = 2, WR_PT0 = 4, WR_PT1 = 8, WR_ALL = 15;
template <class V, int ROLES = WR_ALL>
BMPC_INL void walk_rows(PGP pg, GCD lbx, GCD ubx, int N, int k, const double* y,
                        const double* zeta, const KinT& K, const SegCtx& C, V& v) {
    const bool term = (k == N - 1);
  if constexpr ((ROLES & WR_BOX) != 0) {
    // box bounds on q, dq, ddq, u (BoundMPC.py:171-186, 544-589)
    // the bounds of 14 positions are loaded in one batch: the walk is conditional, so loads issued where they are used
    // cost one memory round trip each (the thread-per-pair kernels run one wavefront per SIMD: nothing hides it)
    BMPC_UNROLL
    for (int half = 0; half < 2; half++) {
        if (half == 0) v.template group<0, 28>(~0u); else v.template group<28, 28>(~0u);
        double ubv[14], lbv[14];
        BMPC_UNROLL
        for (int i = 0; i < 14; i++) {
            const int blk = 2 * half + i / 7, jj = i % 7;
            size_t wi = (size_t)blk * 7 * N + (size_t)jj * N + k;
            ubv[i] = ubx[wi]; lbv[i] = lbx[wi];
        }
        BMPC_UNROLL
        for (int i = 0; i < 14; i++) {
            const int blk = 2 * half + i / 7, jj = i % 7;
            const int pos = (blk == 0 ? Z_Q : blk == 1 ? Z_DQ : blk == 2 ? Z_DDQ : Z_U) + jj;
            const int s = 2 * (blk * 7 + jj);
            const double ub = ubv[i], lb = lbv[i];
            if (ub < BIGB) v.diag(s, pos, 1.0, y[pos] - ub); else v.skip(s);
            if (lb > -BIGB) v.diag(s + 1, pos, -1.0, lb - y[pos]); else v.skip(s + 1);
        }
    }
    // rs, drs, ps, dps >= 0 (Q6)
    v.template group<S_NONNEG, S_EE - S_NONNEG>(k == 1 ? ~0u : 0xfu);         // (the eight slots from S_RS1 on exist at stage 1 only)
    v.diag(S_NONNEG + 0, Z_RS, -1.0, -y[Z_RS]);
    v.diag(S_NONNEG + 1, Z_DRS, -1.0, -y[Z_DRS]);
    v.diag(S_NONNEG + 2, Z_PS, -1.0, -y[Z_PS]);
    v.diag(S_NONNEG + 3, Z_DPS, -1.0, -y[Z_DPS]);
    if (k == 1) {
        v.zdiag(S_RS1, Z_RS, -1.0, -zeta[Z_RS]);
        v.zdiag(S_RS1 + 1, Z_PS, -1.0, -zeta[Z_PS]);
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) v.diag(S_D1 + i, Z_D + i, -1.0, -y[Z_D + i]);
    } else {
        BMPC_UNROLL
        for (int i = 0; i < 8; i++) v.skip(S_RS1 + i);
    }
  }
  if constexpr ((ROLES & WR_POSE) != 0) {
    // EE in current set (ocp :304)
    {
        const unsigned act = (unsigned)pg[PL_MASKE + C.s];
        v.template group<S_EE, S_COL - S_EE>(act | ~0x7fffu);
        PGP a = pg + P_ASET + 45 * C.s;
        BMPC_UNROLL
        for (int rr = 0; rr < 15; rr++) {
            if ((act >> rr) & 1u) {
                double a0 = a[rr], a1 = a[rr + 15], a2 = a[rr + 30], bb = pg[P_BSET + rr * 4 + C.s];
                double a6[6] = {a0, a1, a2, 0, 0, 0};
                v.pose(S_EE + rr, a6, 1, a0 * C.pose[0] + a1 * C.pose[1] + a2 * C.pose[2] - bb - y[Z_PS]);
            } else v.skip(S_EE + rr);
        }
    }
    // orientation bounds (ocp :308-321)
    BMPC_UNROLL
    for (int m = 0; m < 3; m++) {
        double au[6], al[6];
        BMPC_UNROLL
        for (int c = 0; c < 6; c++) { au[c] = C.gs[m][c]; al[c] = -C.gs[m][c]; }
        v.pose(S_ROTU + m, au, 2, C.proj[m] - C.ub[m] - y[Z_RS]);
        v.pose(S_ROTL + m, al, 2, -(C.proj[m] - C.lb[m] + y[Z_RS]));
    }
    // (the pose rows of the walk come first, the collision points last: the reference context is dead by then)
    // phi cap (ocp :332)
    if (term) v.template group<S_PHI, S_END - S_PHI>(~0u); else v.template group<S_PHI, 1>(~0u);      // (the 21 terminal slots exist at the last stage only)
    {
        double a6[6] = {C.dpp[0], C.dpp[1], C.dpp[2], 0, 0, 0};
        v.pose(S_PHI, a6, 0, C.phi - (C.phiend + 0.005));
    }
    if (term) {
        PGP a = pg + P_ASET + 45 * C.n;
        BMPC_UNROLL
        for (int rr = 0; rr < 15; rr++) {
            double an[3] = {a[rr], a[rr + 15], a[rr + 30]};
            double bn = pg[P_BSET + rr * 4 + C.n];
            if (!(an[0] == 0 && an[1] == 0 && an[2] == 0 && bn + pg[P_SLACKS0 + 5] > 0)) {
                double a1 = dot3(an, C.bp1), a2 = dot3(an, C.bp2);
                double bnew = bn - dot3(an, C.pend);
                double a6[6] = {0, 0, 0, 0, 0, 0};
                BMPC_UNROLL
                for (int c = 0; c < 3; c++) {
                    double tt = 0;
                    BMPC_UNROLL
                    for (int a_ = 0; a_ < 3; a_++) tt += (a1 * C.bp1[a_] + a2 * C.bp2[a_]) * C.Dep[a_][c];
                    a6[c] = tt;
                }
                v.pose(S_TSET + rr, a6, 3, a1 * C.tz[0] + a2 * C.tz[1] - bnew - C.sl[5]);
            } else v.skip(S_TSET + rr);
        }
        BMPC_UNROLL
        for (int m = 0; m < 3; m++) {
            double au[6], al[6];
            BMPC_UNROLL
            for (int c = 0; c < 6; c++) { au[c] = C.gsn[m][c]; al[c] = -C.gsn[m][c]; }
            v.pose(S_TROTU + m, au, 3, C.projn[m] - C.ubn[m] - C.sl[5]);
            v.pose(S_TROTL + m, al, 3, -(C.projn[m] - C.lbn[m] + C.sl[5]));
        }
    } else {
        BMPC_UNROLL
        for (int i = 0; i < 21; i++) v.skip(S_TSET + i);
    }
  }
    // collision points (ocp :323-330)
    if constexpr ((ROLES & WR_PT0) != 0 || (ROLES & WR_PT1) != 0) {
        double sl[6];
        BMPC_UNROLL
        for (int i = 0; i < 6; i++) sl[i] = pg[P_SLACKS0 + i] + y[Z_D + i];        // (= C.sl, seg_ctx_eval)
        if constexpr ((ROLES & WR_PT0) != 0) walk_points<V, 0, 3>(pg, K, sl, v);
        if constexpr ((ROLES & WR_PT1) != 0) walk_points<V, 3, 6>(pg, K, sl, v);
    }
}

}  // namespace bmpc
