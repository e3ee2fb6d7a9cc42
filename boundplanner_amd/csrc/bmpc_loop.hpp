// bmpc_loop.hpp -- device-resident closed loop: the per-rollout logic of one receding-horizon step
// either side of the NLP solve, one thread per rollout (BASELINE.json configs[4]: thousands of
// rollouts advanced in lock step, nothing but the rollout log crosses PCIe).
//
// Restates (SURVEY.md section 8 rows a9, a11, a12 (sliding window), a13):
//   BoundMPC.step before the solve           /root/reference/bound_planner/BoundMPC/BoundMPC.py:388-589
//   acceptance + compute_return_data         BoundMPC.py:604-676, 678-1040
//   reference_function / error_function      BoundMPC/bound_mpc_functions.py:49-390 (numpy branches)
//   compute_initial_rot_errors, integrate_rotation_reference   bound_mpc_functions.py:16-46
//   compute_orientation_projection_vectors   BoundMPC.py:338-386
//   jac_SO3_inv_left/right, rodrigues_matrix utils/optimization_functions.py:35-104
//   ReferencePath.update / set_point         ReferencePath/ReferencePath.py:160-207
//   integrate_joint (first hat interval)     utils/util_functions.py:55-65, jerk_trajectory_casadi.py:78-175
//   MPCNode.step state advance               BoundMPC/MPCNode.py:106-160
//   obstacle-free collision sets             BoundPlanner/ConvexSetFinder.py:400-421 (+ util_functions.py:121-135)
// The plan-time construction (ReferencePath.__init__, BoundMPC.update) stays on the host
// (boundplanner_amd/reference_path.py, bound_mpc.py) and is serialised into the state vector below by
// boundplanner_amd/device_loop.py.  Scenes with obstacles need the host collision-set finder and are
// refused by the device loop.
//
// Written against the platform macros of bmpc_platform_hip.hpp so that tests/emu/emu_loop.cpp can run
// the identical source on the CPU against the reference's closed-loop trace (test infrastructure only).
#pragma once
#include "bmpc_device.hpp"

#ifndef BMPC_UNROLL
#define BMPC_UNROLL _Pragma("unroll")
#endif

namespace bmpc {

constexpr int LP_NL = 11;        // via points + nr_segs-1 padded copies, at most (8 via points)
constexpr int LP_S = 4;          // nr_segs
constexpr int LP_ROWS = 15;      // max_set_size
constexpr int LP_NMAX = 64;      // horizon bound of the solver
constexpr int LP_LOGW = 24;      // doubles per (step, rollout) log row
// Optional per-(step, selected rollout) record with the content of the reference's trace message (boundmpcmsg/msg/MPCData.msg:1-64):
//   header [16]: iters, status, viol, error_count, n (valid stages), sector, phi_max, split_idxs[5] (before the step), next selector, 0, 0, 0
//   stage i < N [LP_REC_STAGE each; zero beyond n]: p 6, v 6, q 7, dq 7, ddq 7, dddq 7, phi, dphi, e_p 3, de_p 3, e_r 3, de_r 3,
//       e_r_orth1, e_r_par, e_r_orth2 (coordinates along br1, dp_normed, br2), p_ref 6, segment
//   sets [600]: the a_set / b_set / a_set_joints / b_set_joints blocks of the step's parameter vector (casadi_ocp_formulation.py:407-415)
constexpr int LP_REC_HDR = 16, LP_REC_STAGE = 64, LP_REC_SETS = 600;
BMPC_HD int lp_rec_doubles(int N) { return LP_REC_HDR + LP_REC_STAGE * N + LP_REC_SETS; }

// ---- state vector of one rollout (doubles; integers are stored exactly) -------------------------
#define LP_FIELDS(X)                                                                                  \
    X(q, 7) X(dq, 7) X(ddq, 7) X(jerk, 7) X(qf, 7) X(v, 6) X(p_lie, 6)                                \
    X(split, 5) X(sw, 1) X(error_count, 1) X(has_prev, 1) X(slacks0, 6) X(pr_ref, 3) X(iw_ref, 3)     \
    X(phi_current, 1) X(dphi_current, 1) X(phi_max, 1) X(weights, 11)                                 \
    X(dtau, 12) X(dtau_par, 12) X(dtau_o1, 12) X(dtau_o2, 12) /* [seg][c] */                          \
    X(jac_l, 9) X(jac_r, 9) /* row-major */ X(v1, 12) X(v2, 12) X(v3, 12) /* [c][seg] */              \
    X(patch, 1) X(patch_delta, 3) X(accept, 1) X(dead, 1) X(steps, 1)                                 \
    X(rp_sector, 1) X(rp_num_sectors, 1) X(rp_phi_bias, 1) X(rp_phi_max, 1)                           \
    X(rp_p, 33) X(rp_r_tau, 33) X(rp_dr, 33) X(rp_drn, 33) X(rp_iw, 33) X(rp_dp, 33) X(rp_phi, 12)    \
    X(rp_bp1, 33) X(rp_bp2, 33) X(rp_br1, 33) X(rp_br2, 33) X(rp_erb, 66) X(rp_a, 495) X(rp_b, 165)   \
    X(rp_pd, 24) X(rp_r_taud, 12) X(rp_dpd, 24) X(rp_dpdn, 12) /* [c][seg] */ X(rp_phi_switch, 5)

enum LoopField {
#define X(n, c) LF_##n,
    LP_FIELDS(X)
#undef X
        LF_COUNT
};
constexpr int LS_CNT[LF_COUNT] = {
#define X(n, c) c,
    LP_FIELDS(X)
#undef X
};
constexpr int ls_off(int f) {
    int o = 0;
    for (int i = 0; i < f; i++) o += LS_CNT[i];
    return o;
}
#define X(n, c) constexpr int LS_##n = ls_off(LF_##n);
LP_FIELDS(X)
#undef X
constexpr int LS_SIZE = (ls_off(LF_COUNT) + 7) / 8 * 8;

// robot limits and collision-sphere radii (URDF <limit>, RobotModel.py:37-54, BoundMPC.py:171-191): RobotConst

// ---- SO(3) helpers (conventions of scipy.spatial.transform.Rotation, which the reference calls) ----
BMPC_INL void lp_quat_to_mat(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w, xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    R[0] = x2 - y2 - z2 + w2; R[1] = 2 * (xy - zw);       R[2] = 2 * (xz + yw);
    R[3] = 2 * (xy + zw);     R[4] = -x2 + y2 - z2 + w2;  R[5] = 2 * (yz - xw);
    R[6] = 2 * (xz - yw);     R[7] = 2 * (yz + xw);       R[8] = -x2 - y2 + z2 + w2;
}

BMPC_INL void lp_rotvec_to_mat(const double* v, double* R) {
    const double a = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double sc;
    if (a <= 1e-3) { const double a2 = a * a; sc = 0.5 - a2 / 48 + a2 * a2 / 3840; }
    else sc = sin(a / 2) / a;
    const double q[4] = {sc * v[0], sc * v[1], sc * v[2], cos(a / 2)};
    lp_quat_to_mat(q, R);
}

BMPC_INL void lp_mat_to_rotvec(const double* M, double* v) {
    const double tr = M[0] + M[4] + M[8];
    const double dec[4] = {M[0], M[4], M[8], tr};
    int ch = 0;
    for (int i = 1; i < 4; i++) if (dec[i] > dec[ch]) ch = i;
    double q[4];
    if (ch != 3) {
        const int i = ch, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1 - tr + 2 * M[3 * i + i];
        q[j] = M[3 * j + i] + M[3 * i + j];
        q[k] = M[3 * k + i] + M[3 * i + k];
        q[3] = M[3 * k + j] - M[3 * j + k];
    } else {
        q[0] = M[7] - M[5]; q[1] = M[2] - M[6]; q[2] = M[3] - M[1]; q[3] = 1 + tr;
    }
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= nq;
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
    const double a = 2 * atan2(sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), q[3]);
    double sc;
    if (a <= 1e-3) { const double a2 = a * a; sc = 2 + a2 / 12 + 7 * a2 * a2 / 2880; }
    else sc = a / sin(a / 2);
    for (int i = 0; i < 3; i++) v[i] = sc * q[i];
}

// extrinsic z-y-x Euler angles (as_euler("zyx")): M = Rx(e2) Ry(e1) Rz(e0)
BMPC_INL void lp_euler_zyx(const double* M, double* e) {
    double s = M[2];
    s = s > 1.0 ? 1.0 : (s < -1.0 ? -1.0 : s);
    e[1] = asin(s);
    e[0] = atan2(-M[1], M[0]);
    e[2] = atan2(-M[5], M[8]);
    // a half turn comes out as +pi or -pi depending on the SIGN OF A ZERO in M (structural zeros of the padded segments'
    // bases): made deterministic -- +pi -- here and in so3.compute_initial_rot_errors, where the reference's value is a coin flip
    const double PI = 3.14159265358979323846;
    if (fabs(fabs(e[0]) - PI) < 1e-12) e[0] = PI;
    if (fabs(fabs(e[2]) - PI) < 1e-12) e[2] = PI;
}

BMPC_INL void lp_mat3T(const double* A, double* T) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * j + i];
}

// jac_SO3_inv_right (sign +1) / left (sign -1); angle = |axis| + 1e-6 (optimization_functions.py:37,54)
BMPC_INL void lp_jac_inv(const double* ax, double sign, double* J) {
    const double a = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-6;
    const double W[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
    double W2[9];
    mat3mul(W, W, W2);
    const double coef = 1.0 / (a * a) - (1.0 + cos(a)) / (2.0 * a * sin(a));
    for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? 1.0 : 0.0) + sign * 0.5 * W[i] + coef * W2[i];
}

// integrate_rotation_reference (bound_mpc_functions.py:16-27)
BMPC_INL void lp_integrate_rot_ref(const double* pr_ref, const double* omega, double phi0, double phi1, double* out) {
    double r0[9], r1[9];
    lp_rotvec_to_mat(pr_ref, r0);
    const double n = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    if (n > 1e-4) {
        const double w[3] = {omega[0] / n, omega[1] / n, omega[2] / n}, ph = (phi1 - phi0) * n;
        const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        double W2[9], Rr[9];
        mat3mul(W, W, W2);
        const double sp = sin(ph), cp = 1.0 - cos(ph);
        for (int i = 0; i < 9; i++) Rr[i] = ((i % 4 == 0) ? 1.0 : 0.0) + sp * W[i] + cp * W2[i];
        mat3mul(Rr, r0, r1);
    } else {
        for (int i = 0; i < 9; i++) r1[i] = r0[i];
    }
    lp_mat_to_rotvec(r1, out);
}

// compute_initial_rot_errors (bound_mpc_functions.py:30-46): e0, e_par, e_orth1, e_orth2
BMPC_INL void lp_initial_rot_errors(const double* pr, const double* pr_ref, const double* dpn, const double* br1,
                                    const double* br2, double* e0, double* epar, double* eo1, double* eo2) {
    double tc[9], td[9], tdT[9], m[9];
    lp_rotvec_to_mat(pr, tc);
    lp_rotvec_to_mat(pr_ref, td);
    lp_mat3T(td, tdT);
    mat3mul(tc, tdT, m);
    lp_mat_to_rotvec(m, e0);
    double r01[9], r01T[9], Rd[9], t1[9], d01[9], eul[3];
    for (int i = 0; i < 3; i++) { r01[3 * i] = br2[i]; r01[3 * i + 1] = dpn[i]; r01[3 * i + 2] = br1[i]; }
    lp_mat3T(r01, r01T);
    lp_rotvec_to_mat(e0, Rd);
    mat3mul(r01T, Rd, t1);
    mat3mul(t1, r01, d01);
    lp_euler_zyx(d01, eul);
    for (int i = 0; i < 3; i++) { epar[i] = eul[1] * dpn[i]; eo1[i] = eul[0] * br1[i]; eo2[i] = eul[2] * br2[i]; }
}

// X = G^{-1} Bt for 3x3 G, by elimination with partial pivoting (numpy.linalg.solve)
BMPC_INL void lp_solve3(const double* G, const double* Bt, double* X) {
    double A[3][6];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { A[i][j] = G[3 * i + j]; A[i][3 + j] = Bt[3 * i + j]; }
    for (int c = 0; c < 3; c++) {
        int piv = c;
        for (int r = c + 1; r < 3; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 6; j++) { double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        for (int r = c + 1; r < 3; r++) {
            const double f = A[r][c] / A[c][c];
            for (int j = c; j < 6; j++) A[r][j] -= f * A[c][j];
        }
    }
    for (int j = 0; j < 3; j++)
        for (int r = 2; r >= 0; r--) {
            double s = A[r][3 + j];
            for (int c = r + 1; c < 3; c++) s -= A[r][c] * X[3 * c + j];
            X[3 * r + j] = s / A[r][r];
        }
}

// ReferencePath.set_point (ReferencePath.py:160-185)
BMPC_INL void lp_set_point(double* S, int idx) {
    const int j = (int)S[LS_rp_sector] + idx;
    const double* dp = S + LS_rp_dp + 3 * j;
    const double nd = sqrt(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
    for (int c = 0; c < 3; c++) {
        S[LS_rp_pd + LP_S * c + idx] = S[LS_rp_p + 3 * j + c];
        S[LS_rp_pd + LP_S * (3 + c) + idx] = S[LS_rp_iw + 3 * j + c];
        S[LS_rp_r_taud + LP_S * c + idx] = S[LS_rp_r_tau + 3 * j + c];
        S[LS_rp_dpd + LP_S * c + idx] = dp[c] / nd;
        S[LS_rp_dpd + LP_S * (3 + c) + idx] = S[LS_rp_dr + 3 * j + c];
        S[LS_rp_dpdn + LP_S * c + idx] = S[LS_rp_drn + 3 * j + c];
    }
    double cs = 0;
    for (int i = 0; i <= j + 1; i++) cs += S[LS_rp_phi + i];
    S[LS_rp_phi_switch + idx + 1] = cs + S[LS_rp_phi_bias];
}

// ReferencePath.update (ReferencePath.py:187-207)
BMPC_INL void lp_path_update(double* S, bool sw) {
    if ((int)S[LS_rp_sector] >= (int)S[LS_rp_num_sectors] || !sw) return;
    S[LS_rp_sector] += 1;
    for (int c = 0; c < 6; c++)
        for (int i = 0; i < LP_S - 1; i++) {
            S[LS_rp_pd + LP_S * c + i] = S[LS_rp_pd + LP_S * c + i + 1];
            S[LS_rp_dpd + LP_S * c + i] = S[LS_rp_dpd + LP_S * c + i + 1];
        }
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < LP_S - 1; i++) {
            S[LS_rp_r_taud + LP_S * c + i] = S[LS_rp_r_taud + LP_S * c + i + 1];
            S[LS_rp_dpdn + LP_S * c + i] = S[LS_rp_dpdn + LP_S * c + i + 1];
        }
    for (int i = 0; i < LP_S - 1; i++) S[LS_rp_phi_switch + i] = S[LS_rp_phi_switch + i + 1];
    S[LS_rp_phi_switch + LP_S - 1] = S[LS_rp_phi_switch + LP_S] + S[LS_rp_phi_bias];
    lp_set_point(S, LP_S - 1);
}

// ---- per-step collision sets with obstacles (ConvexSetFinder.find_set_collision_avoidance, ---------
// ConvexSetFinder.py:309-375, with compute_set_projs_line :491-510): the scene's obstacle polytopes are shared
// by all rollouts.  The closest pair between the segment [p(q0), p(qf)] of a collision point and a polytope is the
// algorithm of the host restatement (boundplanner_amd/collision_sets.py: golden section over the segment
// parameter, each distance an exact projection by Hildreth's dual coordinate ascent), so that both sides agree
// to rounding; one thread per (rollout, collision point, obstacle), then the greedy nearest-first selection of
// separating halfspaces per (rollout, collision point) inside loop_prepare.
constexpr int LP_MAXOBS = 16;    // obstacle polytopes per scene
constexpr int LP_NV = 32;        // vertices per obstacle
constexpr int LP_CRES = 8;       // doubles per closest-pair result: x(3), y(3), distance, pad

struct LoopScene {
    int n_obs;
    const double* A;      // [n_obs][15][3], rows beyond nrows are zero
    const double* b;      // [n_obs][15]
    const double* AAt;    // [n_obs][15][15]
    const int* nrows;     // [n_obs]
    const double* V;      // [n_obs][LP_NV][3]
    const int* nv;        // [n_obs]
    const double* box;    // [n_obs][6]: lo(3), hi(3) of obstacles that are axis-aligned boxes (is_box[o] != 0)
    const int* is_box;    // [n_obs]
};

// host side (upload / CPU harness): is {A x <= b} (nr rows) an axis-aligned box?  If so lo/hi are its bounds.
inline bool loop_detect_box(const double* A, const double* b, int nr, double* lo, double* hi) {
    if (nr != 6) return false;
    bool have[6] = {false, false, false, false, false, false};
    for (int r = 0; r < 6; r++) {
        int ax = -1;
        for (int c = 0; c < 3; c++) {
            const double a = A[3 * r + c];
            if (a == 0.0) continue;
            if ((a != 1.0 && a != -1.0) || ax >= 0) return false;
            ax = c;
        }
        if (ax < 0) return false;
        if (A[3 * r + ax] > 0) { if (have[ax]) return false; have[ax] = true; hi[ax] = b[r]; }
        else { if (have[3 + ax]) return false; have[3 + ax] = true; lo[ax] = -b[r]; }
    }
    for (int i = 0; i < 6; i++) if (!have[i]) return false;
    return true;
}

// Euclidean projection of y onto {x: A x <= b - 0.001} (collision_sets._project_polytope).  Loops run over the fixed
// LP_ROWS with an early exit at nr so that, unrolled, Ay / lam stay in registers (static indices)
BMPC_INL void lp_project_polytope(const double* A, const double* b, const double* AAt, int nr, const double* y, double* x) {
    double Ay[LP_ROWS], lam[LP_ROWS];
    bool inside = true;
    BMPC_UNROLL
    for (int i = 0; i < LP_ROWS; i++) {
        Ay[i] = 0.0; lam[i] = 0.0;
        if (i < nr) {
            Ay[i] = A[3 * i] * y[0] + A[3 * i + 1] * y[1] + A[3 * i + 2] * y[2];
            if (Ay[i] - (b[i] - 0.001) > 1e-12) inside = false;
        }
    }
    x[0] = y[0]; x[1] = y[1]; x[2] = y[2];
    if (inside) return;
    for (int sweep = 0; sweep < 1200; sweep++) {
        double max_change = 0.0;
        BMPC_UNROLL
        for (int i = 0; i < LP_ROWS; i++) {
            if (i < nr) {
                double r = Ay[i];
                BMPC_UNROLL
                for (int j = 0; j < LP_ROWS; j++)
                    if (j < nr) r -= AAt[LP_ROWS * i + j] * lam[j];
                r -= (b[i] - 0.001);
                const double dg = fmax(AAt[LP_ROWS * i + i], 1e-16);
                const double nw = fmax(0.0, lam[i] + r / dg);
                max_change = fmax(max_change, fabs(nw - lam[i]));
                lam[i] = nw;
            }
        }
        if (max_change < 1e-13) break;
    }
    BMPC_UNROLL
    for (int i = 0; i < LP_ROWS; i++)
        if (i < nr)
            for (int c = 0; c < 3; c++) x[c] -= A[3 * i + c] * lam[i];
}

// distance from the segment point p0 + phi d to the polytope {A x <= b - 0.001}; box != null: the polytope is the
// axis-aligned box [lo, hi] and its exact projection is a clamp (what Hildreth's iteration converges to)
BMPC_INL double lp_seg_dist(const double* A, const double* b, const double* AAt, int nr, const double* box, const double* p0,
                            const double* d, double phi, double* x) {
    const double y[3] = {p0[0] + phi * d[0], p0[1] + phi * d[1], p0[2] + phi * d[2]};
    if (box) {
        for (int c = 0; c < 3; c++) x[c] = fmin(fmax(y[c], box[c] + 0.001), box[3 + c] - 0.001);
    } else {
        lp_project_polytope(A, b, AAt, nr, y, x);
    }
    return sqrt((y[0] - x[0]) * (y[0] - x[0]) + (y[1] - x[1]) * (y[1] - x[1]) + (y[2] - x[2]) * (y[2] - x[2]));
}

// closest pair segment <-> polytope (collision_sets.closest_pair_segment_polytope); out: x, y = p0 + phi d, distance
BMPC_DEV void loop_closest_pair(const double* A, const double* b, const double* AAt, int nr, const double* box, const double* p0,
                                const double* p1, double* out) {
    const double d[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    double x[3], phi = 0.0;
    if (sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) >= 1e-12) {
        double lo = 0.0, hi = 1.0;
        const double gr = (sqrt(5.0) - 1.0) / 2.0;
        double c = hi - gr * (hi - lo), e = lo + gr * (hi - lo);
        double fc = lp_seg_dist(A, b, AAt, nr, box, p0, d, c, x), fe = lp_seg_dist(A, b, AAt, nr, box, p0, d, e, x);
        for (int it = 0; it < 80; it++) {
            if (fc < fe) {
                hi = e; e = c; fe = fc;
                c = hi - gr * (hi - lo);
                fc = lp_seg_dist(A, b, AAt, nr, box, p0, d, c, x);
            } else {
                lo = c; c = e; fc = fe;
                e = lo + gr * (hi - lo);
                fe = lp_seg_dist(A, b, AAt, nr, box, p0, d, e, x);
            }
        }
        const double pm = 0.5 * (lo + hi);
        const double f0 = lp_seg_dist(A, b, AAt, nr, box, p0, d, 0.0, x), f1 = lp_seg_dist(A, b, AAt, nr, box, p0, d, 1.0, x),
                     fm = lp_seg_dist(A, b, AAt, nr, box, p0, d, pm, x);
        // min over (distance, phi) pairs in the order (0, 1, pm): ties go to the smaller phi
        double best = f0; phi = 0.0;
        if (f1 < best) { best = f1; phi = 1.0; }
        if (fm < best || (fm == best && pm < phi)) { best = fm; phi = pm; }
    }
    const double dist = lp_seg_dist(A, b, AAt, nr, box, p0, d, phi, x);
    for (int c = 0; c < 3; c++) { out[c] = x[c]; out[3 + c] = p0[c] + phi * d[c]; }
    out[6] = dist; out[7] = phi;
}

// one (rollout, collision point, obstacle) of the closest-pair pass
BMPC_DEV void loop_collision_pair(const RobotConst* rc, const LoopScene& sc, const double* S, int pt, int ob, double* out) {
    Kin k0, kf;
    kin_eval(rc, S + LS_q, k0);
    kin_eval(rc, S + LS_qf, kf);
    loop_closest_pair(sc.A + 45 * ob, sc.b + LP_ROWS * ob, sc.AAt + LP_ROWS * LP_ROWS * ob, sc.nrows[ob],
                      sc.is_box[ob] ? sc.box + 6 * ob : nullptr, k0.pc[pt], kf.pc[pt], out);
}

// greedy nearest-first separating halfspaces of one collision point (ConvexSetFinder.py:330-375); res: the
// closest-pair results of this (rollout, point) for all obstacles; rows 6.. of a[15][3], b[15] are appended.
// Returns the number of rows, or -1 when they do not fit max_set_size (the host raises there).
BMPC_INL int lp_collision_rows(const LoopScene& sc, const double* res, const double* p0, const double* p1, double a[][3], double* b) {
    int n = 6;
    bool remain[LP_MAXOBS];
    for (int i = 0; i < sc.n_obs; i++) remain[i] = true;
    for (;;) {
        int idx = -1;
        for (int i = 0; i < sc.n_obs; i++)
            if (remain[i] && (idx < 0 || res[LP_CRES * i + 6] < res[LP_CRES * idx + 6])) idx = i;
        if (idx < 0) break;
        const double* cp = res + LP_CRES * idx;
        double av[3] = {cp[0] - cp[3], cp[1] - cp[4], cp[2] - cp[5]};
        double na = sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
        if (na < 1e-6) {          // the segment touches the obstacle
            for (int c = 0; c < 3; c++) av[c] = cp[c] - p0[c];
            na = sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
            if (na < 1e-6) {
                for (int c = 0; c < 3; c++) av[c] = p1[c] - p0[c];
                na = sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
            }
        }
        for (int c = 0; c < 3; c++) av[c] /= na;
        const double bh = av[0] * cp[0] + av[1] * cp[1] + av[2] * cp[2] - 0.001;
        remain[idx] = false;
        for (int i = 0; i < sc.n_obs; i++) {
            if (!remain[i]) continue;
            double mn = 1e300;
            for (int v = 0; v < sc.nv[i]; v++) {
                const double* vv = sc.V + 3 * (LP_NV * i + v);
                mn = fmin(mn, vv[0] * av[0] + vv[1] * av[1] + vv[2] * av[2] - bh);
            }
            if (mn >= -1e-4) remain[i] = false;     // entirely behind the new halfspace
        }
        if (n >= LP_ROWS) return -1;
        a[n][0] = av[0]; a[n][1] = av[1]; a[n][2] = av[2]; b[n] = bh;
        n++;
    }
    return n;
}

// ---- before the solve: BoundMPC.step up to the solver call ---------------------------------------
// S: state; prev: previous solution row (read when has_prev); p/lbx/ubx: rows of the solver arguments
// (lbx/ubx hold the constant limits already, only the stage-0 pins are written here).
BMPC_DEV void loop_prepare(const RobotConst* rc, int N, double* S, const double* prev, double* p, double* lbx, double* ubx,
                           const LoopScene* sc = nullptr, const double* colres = nullptr) {
    // MPCNode.step: p_lie = fk(q) (MPCNode.py:118)
    Kin k;
    kin_eval(rc, S + LS_q, k);
    double p0[6];
    for (int c = 0; c < 3; c++) p0[c] = k.pee[c];
    lp_mat_to_rotvec(k.Ree, p0 + 3);
    for (int c = 0; c < 6; c++) S[LS_p_lie + c] = p0[c];

    lp_path_update(S, S[LS_sw] != 0.0);
    S[LS_sw] = 0.0;
    const int sec = (int)S[LS_rp_sector];

    // warm start: previous solution unshifted, with the omega-reversal patch (BoundMPC.py:418-428)
    S[LS_patch] = 0.0;
    if (S[LS_has_prev] != 0.0) {
        double d[3], n2 = 0;
        for (int c = 0; c < 3; c++) { d[c] = p0[3 + c] - prev[28 * N + (3 + c) * N]; n2 += d[c] * d[c]; }
        if (sqrt(n2) > 1.5) {
            S[LS_patch] = 1.0;
            for (int c = 0; c < 3; c++) S[LS_patch_delta + c] = d[c];
        }
    }

    // initial orientation errors per segment (BoundMPC.py:436-462)
    for (int i = 0; i < LP_S; i++) {
        double prs[3], dpn[3];
        for (int c = 0; c < 3; c++) {
            prs[c] = (i == 0) ? S[LS_pr_ref + c] : S[LS_rp_r_taud + LP_S * c + i];
            dpn[c] = S[LS_rp_dpdn + LP_S * c + i];
        }
        lp_initial_rot_errors(p0 + 3, prs, dpn, S + LS_rp_br1 + 3 * (sec + i), S + LS_rp_br2 + 3 * (sec + i),
                              S + LS_dtau + 3 * i, S + LS_dtau_par + 3 * i, S + LS_dtau_o1 + 3 * i, S + LS_dtau_o2 + 3 * i);
    }
    // projection vectors (BoundMPC.py:338-386); one jac_dtau_l/r from segment 0 serves all segments
    lp_jac_inv(S + LS_dtau, +1.0, S + LS_jac_r);
    lp_jac_inv(S + LS_dtau, -1.0, S + LS_jac_l);
    double rinit0[9];
    lp_rotvec_to_mat(S + LS_dtau, rinit0);
    for (int i = 0; i < LP_S; i++) {
        double Ro[9], RoT[9], rest1[9], Rp[9], RpT[9], rest2[9], rv[3], j1[9], j2[9], dpn[3];
        lp_rotvec_to_mat(S + LS_dtau_o1 + 3 * i, Ro);
        lp_mat3T(Ro, RoT);
        mat3mul(rinit0, RoT, rest1);
        lp_rotvec_to_mat(S + LS_dtau_par + 3 * i, Rp);
        lp_mat3T(Rp, RpT);
        mat3mul(rest1, RpT, rest2);
        lp_mat_to_rotvec(rest1, rv);
        lp_jac_inv(rv, +1.0, j1);
        lp_mat_to_rotvec(rest2, rv);
        lp_jac_inv(rv, +1.0, j2);
        for (int c = 0; c < 3; c++) dpn[c] = S[LS_rp_dpdn + LP_S * c + i];
        double g[3], h[3], kk[3];
        mat3vec(S + LS_jac_r, S + LS_rp_br1 + 3 * (sec + i), g);
        mat3vec(j1, dpn, h);
        mat3vec(j2, S + LS_rp_br2 + 3 * (sec + i), kk);
        double Bt[9], G[9], X[9];      // Bt = Bm^T with Bm = [g h k] columns
        for (int c = 0; c < 3; c++) { Bt[c] = g[c]; Bt[3 + c] = h[c]; Bt[6 + c] = kk[c]; }
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) G[3 * a + b] = Bt[3 * a] * Bt[3 * b] + Bt[3 * a + 1] * Bt[3 * b + 1] + Bt[3 * a + 2] * Bt[3 * b + 2];
        lp_solve3(G, Bt, X);
        for (int c = 0; c < 3; c++) {
            S[LS_v1 + LP_S * c + i] = X[c]; S[LS_v2 + LP_S * c + i] = X[3 + c]; S[LS_v3 + LP_S * c + i] = X[6 + c];
        }
    }

    // Q10: w_phi rescale only when phi_max < 1; phi target clamped to phi + 5 (BoundMPC.py:465-478)
    const double phi_max = S[LS_phi_max], phi_cur = S[LS_phi_current];
    double w4 = S[LS_weights + 4];
    if (phi_max < 1 && phi_max > 0.001) w4 *= fmin(1.0 / ((phi_max - phi_cur) * (phi_max - phi_cur)), 2.0);
    const double phi_clamped = fmin(phi_cur + 5.0, phi_max);

    // ---- the 875 parameters in the order of BoundMPC.py:507-542 ----
    for (int i = 0; i < 5; i++) p[P_SPLIT + i] = S[LS_split + i];
    for (int i = 0; i < 6; i++) p[P_SLACKS0 + i] = S[LS_slacks0 + i];
    for (int i = 0; i < 3; i++) p[P_IWREF + i] = S[LS_iw_ref + i];
    for (int i = 0; i < 12; i++) {
        p[P_DTAU + i] = S[LS_dtau + i]; p[P_DTAU_PAR + i] = S[LS_dtau_par + i];
        p[P_DTAU_O1 + i] = S[LS_dtau_o1 + i]; p[P_DTAU_O2 + i] = S[LS_dtau_o2 + i];
        p[P_DPN + i] = S[LS_rp_dpdn + i];
        p[P_V1 + i] = S[LS_v1 + i]; p[P_V2 + i] = S[LS_v2 + i]; p[P_V3 + i] = S[LS_v3 + i];
    }
    p[P_XPHID] = phi_clamped; p[P_XPHID + 1] = 0.0; p[P_XPHID + 2] = 0.0;
    for (int i = 0; i < 5; i++) p[P_PHISW + i] = S[LS_rp_phi_switch + i];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) { p[P_JACR + 3 * a + b] = S[LS_jac_r + 3 * b + a]; p[P_JACL + 3 * a + b] = S[LS_jac_l + 3 * b + a]; }
    for (int i = 0; i < 24; i++) { p[P_PREF + i] = S[LS_rp_pd + i]; p[P_DPREF + i] = S[LS_rp_dpd + i]; }
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < LP_S; i++) {
            p[P_BP1 + LP_S * c + i] = S[LS_rp_bp1 + 3 * (sec + i) + c]; p[P_BP2 + LP_S * c + i] = S[LS_rp_bp2 + 3 * (sec + i) + c];
            p[P_BR1 + LP_S * c + i] = S[LS_rp_br1 + 3 * (sec + i) + c]; p[P_BR2 + LP_S * c + i] = S[LS_rp_br2 + 3 * (sec + i) + c];
        }
    for (int c = 0; c < 6; c++)
        for (int i = 0; i < LP_S; i++) p[P_ERB + LP_S * c + i] = S[LS_rp_erb + 6 * (sec + i) + c];
    for (int i = 0; i < 11; i++) p[P_W + i] = (i == 4) ? w4 : S[LS_weights + i];
    p[P_PHIMAX] = phi_clamped;
    for (int i = 0; i < 7; i++) p[P_V3 + 12 + i] = 0.0;      // qd (unused desired joint configuration, BoundMPC.py:66)
    for (int i = 0; i < LP_S; i++)
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < LP_ROWS; r++) p[P_ASET + 45 * i + LP_ROWS * c + r] = S[LS_rp_a + 45 * (sec + i) + 3 * r + c];
    for (int r = 0; r < LP_ROWS; r++)
        for (int i = 0; i < LP_S; i++) p[P_BSET + LP_S * r + i] = S[LS_rp_b + LP_ROWS * (sec + i) + r];
    // collision sets: 0.7 m box around each collision point at q0 (ConvexSetFinder.py:400-421), separating halfspaces
    // of the scene's obstacles along the segment to the point's position at qf, everything shrunk by the joint size,
    // padded with (A = 0, b = 10) (BoundMPC.py:480-497, util_functions.py:121-135)
    const bool obst = sc && sc->n_obs > 0;
    Kin kf;
    if (obst) kin_eval(rc, S + LS_qf, kf);
    for (int j = 0; j < 6; j++) {
        double a[LP_ROWS][3], b[LP_ROWS];
        for (int r = 0; r < 6; r++) {
            for (int c = 0; c < 3; c++) a[r][c] = ((r >> 1) == c) ? ((r & 1) ? -1.0 : 1.0) : 0.0;
            b[r] = ((r & 1) ? -k.pc[j][r >> 1] : k.pc[j][r >> 1]) + 0.7;
        }
        int n = 6;
        if (obst) {
            n = lp_collision_rows(*sc, colres + LP_CRES * sc->n_obs * j, k.pc[j], kf.pc[j], a, b);
            if (n < 0) { S[LS_dead] = 2.0; n = LP_ROWS; }       // more rows than max_set_size: the host raises
        }
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < LP_ROWS; r++) p[P_ASETJ + 45 * j + LP_ROWS * c + r] = (r < n) ? a[r][c] : 0.0;
        for (int r = 0; r < LP_ROWS; r++) p[P_BSETJ + 6 * r + j] = (r < n) ? b[r] - rc->colsize[j] : 10.0;
    }

    // stage-0 pins (BoundMPC.py:544-580, Q7): x[0:-1:N] = value on the joint-major arrays
    for (int j = 0; j < 7; j++) {
        lbx[j * N] = ubx[j * N] = S[LS_q + j];
        lbx[7 * N + j * N] = ubx[7 * N + j * N] = S[LS_dq + j];
        lbx[14 * N + j * N] = ubx[14 * N + j * N] = S[LS_ddq + j];
        lbx[21 * N + j * N] = ubx[21 * N + j * N] = S[LS_jerk + j];
    }
    for (int j = 0; j < 6; j++) {
        lbx[28 * N + j * N] = ubx[28 * N + j * N] = p0[j];
        lbx[34 * N + j * N] = ubx[34 * N + j * N] = S[LS_v + j];
    }
}

// constant part of the bounds (BoundMPC.py:171-191, 544-589); infinities as +-1e20
BMPC_INL void loop_bound_const(const RobotConst* rc, int N, int i, double* lo, double* hi) {
    double l = 0.0, h = 1e20;
    if (i < 7 * N) { h = rc->q_hi[i / N]; l = rc->q_lo[i / N]; }
    else if (i < 14 * N) { h = rc->dq_max[(i - 7 * N) / N]; l = -h; }
    else if (i < 21 * N) { h = rc->ddq_max; l = -h; }
    else if (i < 28 * N) { h = rc->u_max; l = -h; }
    else if (i < 40 * N) { h = 1e20; l = -1e20; }
    *lo = l; *hi = h;
}

// element i of the start vector: cold start (BoundMPC.py:412-416) or the patched previous solution
BMPC_INL double loop_x0_elem(int N, const double* S, const double* prev, int i) {
    if (S[LS_has_prev] == 0.0) {
        if (i < 7 * N) return S[LS_q + i / N];
        if (i >= 28 * N && i < 34 * N) return S[LS_p_lie + (i - 28 * N) / N];
        return 0.0;
    }
    if (S[LS_patch] != 0.0 && i >= 31 * N && i < 34 * N) {
        const int c = (i - 31 * N) / N, col = (i - 31 * N) - c * N;
        const int src = (col < N - 1) ? col + 1 : N - 1;
        return prev[31 * N + c * N + src] + S[LS_patch_delta + c];
    }
    return prev[i];
}

BMPC_INL int lp_segment(int idx, const double* split, int n_rows) {
    int i = 0;
    for (int j = 0; j < n_rows - 2; j++) if (idx > (int)split[j + 1]) i = j + 1;
    return i;
}

// ---- after the solve: acceptance, compute_return_data, joint integration ---------------------------
// x: the solver's solution row; prev: the stored previous solution row (read only here; the caller copies
// x over it afterwards where S[LS_accept] is set); log: LP_LOGW doubles or null.
BMPC_DEV void loop_finish(const RobotConst* rc, int N, double dt, double* S, const double* x, const double* prev,
                          int status, double viol, int iters, double* log, double* rec = nullptr, const double* par = nullptr) {
    const int n_w = 44 * N + 6;
    S[LS_steps] += 1.0;
    for (int i = 0; i < 6; i++) S[LS_slacks0 + i] += x[n_w - 6 + i];          // Q1
    const bool success = (status == 0) || (viol < 1e-4);                         // Q8
    const double* w = x;
    S[LS_accept] = 0.0;
    if (!success) {
        S[LS_error_count] += 1.0;
        if (S[LS_has_prev] != 0.0) w = prev;
        else S[LS_error_count] = 0.0;
    } else {
        S[LS_error_count] = 0.0;
        S[LS_accept] = 1.0;
        S[LS_has_prev] = 1.0;
    }
    const int ec = (int)S[LS_error_count], n = N - ec;
    if (log) {
        for (int i = 0; i < LP_LOGW; i++) log[i] = 0.0;
        log[0] = (double)iters; log[1] = (double)status; log[2] = viol; log[3] = (double)ec;
    }
    if (rec) {
        const int nr = lp_rec_doubles(N);
        for (int i = 0; i < nr; i++) rec[i] = 0.0;
        rec[0] = (double)iters; rec[1] = (double)status; rec[2] = viol; rec[3] = (double)ec; rec[4] = (double)(n < 0 ? 0 : n);
        rec[5] = S[LS_rp_sector]; rec[6] = S[LS_phi_max];
        for (int i = 0; i < 5; i++) rec[7 + i] = S[LS_split + i];
        if (par) for (int i = 0; i < LP_REC_SETS; i++) rec[LP_REC_HDR + LP_REC_STAGE * N + i] = par[275 + i];
    }
    if (n < 2) { S[LS_dead] = 1.0; if (log) log[4] = 1.0; return; }   // the reference would index past its arrays here

    const int sec = (int)S[LS_rp_sector];
    double split_prev[5], iw_ref_0[3];
    for (int i = 0; i < 5; i++) split_prev[i] = S[LS_split + i];
    for (int c = 0; c < 3; c++) iw_ref_0[c] = S[LS_iw_ref + c];
    const int nxt = ((int)split_prev[1] == N) ? 1 : (((int)split_prev[2] == N) ? 2 : 3);
    if (rec) rec[12] = (double)nxt;

    double opt_phi[LP_NMAX], ers[LP_NMAX][3], ersn[LP_NMAX][3];
    signed char segs[LP_NMAX];
    double opt_dphi1 = 0.0;
    for (int i = 0; i < n; i++) {
        double pp[6], vv[6];
        for (int c = 0; c < 6; c++) { pp[c] = w[28 * N + c * N + ec + i]; vv[c] = w[34 * N + c * N + ec + i]; }
        const int s = lp_segment(i, split_prev, LP_S);
        const double phi_start = S[LS_rp_phi_switch + lp_segment(i, split_prev, LP_S + 1)];
        double pc[6], pn[6], dpd[6], dpn_[6];
        for (int c = 0; c < 6; c++) {
            pc[c] = S[LS_rp_pd + LP_S * c + s]; pn[c] = S[LS_rp_pd + LP_S * c + s + 1];
            dpd[c] = S[LS_rp_dpd + LP_S * c + s]; dpn_[c] = S[LS_rp_dpd + LP_S * c + s + 1];
        }
        double phi_l = 0, phi_next = 0, dphi = 0;
        for (int c = 0; c < 3; c++) { phi_l += (pp[c] - pc[c]) * dpd[c]; phi_next += (pp[c] - pn[c]) * dpn_[c]; dphi += vv[c] * dpd[c]; }
        double pdr[3], pdrn[3], iw0[3], t1[3], t2[3], dl[3], a1[3], a2[3];
        for (int c = 0; c < 3; c++) {
            pdr[c] = dpd[3 + c] * phi_l + pc[3 + c];
            pdrn[c] = dpn_[3 + c] * phi_next + pn[3 + c];
            iw0[c] = (i <= (int)split_prev[1]) ? iw_ref_0[c] : pc[3 + c];
            t1[c] = pp[3 + c] - S[LS_p_lie + 3 + c];
        }
        mat3vec(S + LS_jac_l, t1, dl);
        for (int c = 0; c < 3; c++) { t1[c] = pdr[c] - iw0[c]; t2[c] = pdrn[c] - iw0[c]; }
        mat3vec(S + LS_jac_r, t1, a1);
        mat3vec(S + LS_jac_r, t2, a2);
        // d = e_r - e_init, dn = e_rn - e_initn (evaluated as in the reference: sum first, then subtract)
        double d[3], dn[3];
        for (int c = 0; c < 3; c++) {
            const double ei = S[LS_dtau + 3 * s + c], ein = S[LS_dtau + 3 * nxt + c];
            d[c] = (ei + dl[c] - a1[c]) - ei;
            dn[c] = (ein + dl[c] - a2[c]) - ein;
        }
        const double* br1 = S + LS_rp_br1 + 3 * (sec + s);
        const double* br2 = S + LS_rp_br2 + 3 * (sec + s);
        const double* br1n = S + LS_rp_br1 + 3 * (sec + s + 1);
        const double* br2n = S + LS_rp_br2 + 3 * (sec + s + 1);
        double dv[6] = {0, 0, 0, 0, 0, 0}, dpnv[3], dpnn[3];
        for (int c = 0; c < 3; c++) {
            dpnv[c] = S[LS_rp_dpdn + LP_S * c + s]; dpnn[c] = S[LS_rp_dpdn + LP_S * c + s + 1];
            dv[0] += d[c] * S[LS_v1 + LP_S * c + s]; dv[1] += d[c] * S[LS_v2 + LP_S * c + s]; dv[2] += d[c] * S[LS_v3 + LP_S * c + s];
            dv[3] += dn[c] * S[LS_v1 + LP_S * c + s + 1]; dv[4] += dn[c] * S[LS_v2 + LP_S * c + s + 1]; dv[5] += dn[c] * S[LS_v3 + LP_S * c + s + 1];
        }
        double e[6] = {0, 0, 0, 0, 0, 0};     // orth1, par, orth2, then the next-segment variants
        for (int c = 0; c < 3; c++) {
            e[0] += (S[LS_dtau_o1 + 3 * s + c] + dv[0] * br1[c]) * br1[c];
            e[1] += (S[LS_dtau_par + 3 * s + c] + dv[1] * dpnv[c]) * dpnv[c];
            e[2] += (S[LS_dtau_o2 + 3 * s + c] + dv[2] * br2[c]) * br2[c];
            e[3] += (S[LS_dtau_o1 + 3 * (s + 1) + c] + dv[3] * br1n[c]) * br1n[c];
            e[4] += (S[LS_dtau_par + 3 * (s + 1) + c] + dv[4] * dpnn[c]) * dpnn[c];
            e[5] += (S[LS_dtau_o2 + 3 * (s + 1) + c] + dv[5] * br2n[c]) * br2n[c];
        }
        if (rec) {
            double* q_ = rec + LP_REC_HDR + LP_REC_STAGE * i;
            for (int c = 0; c < 6; c++) { q_[c] = pp[c]; q_[6 + c] = vv[c]; }
            for (int j = 0; j < 7; j++) {
                q_[12 + j] = w[j * N + ec + i]; q_[19 + j] = w[7 * N + j * N + ec + i];
                q_[26 + j] = w[14 * N + j * N + ec + i]; q_[33 + j] = w[21 * N + j * N + ec + i];
            }
            q_[40] = phi_l + phi_start; q_[41] = dphi;
            double lv[3], rv[3], wv[3] = {vv[3], vv[4], vv[5]}, rd[3] = {dpd[3] * dphi, dpd[4] * dphi, dpd[5] * dphi};
            mat3vec(S + LS_jac_l, wv, lv);
            mat3vec(S + LS_jac_r, rd, rv);
            for (int c = 0; c < 3; c++) {
                q_[42 + c] = pp[c] - (pc[c] + dpd[c] * phi_l);             // e_p
                q_[45 + c] = vv[c] - dpd[c] * dphi;                          // de_p
                q_[48 + c] = S[LS_dtau + 3 * s + c] + dl[c] - a1[c];         // e_r
                q_[51 + c] = lv[c] - rv[c];                                  // de_r
                q_[54 + c] = e[c];                                           // e_r_orth1, e_r_par, e_r_orth2
                q_[57 + c] = pc[c] + dpd[c] * phi_l; q_[60 + c] = pdr[c];    // p_ref
            }
            q_[63] = (double)s;
        }
        opt_phi[i] = phi_l + phi_start;
        if (i == 1) opt_dphi1 = dphi;
        for (int c = 0; c < 3; c++) { ers[i][c] = e[c]; ersn[i][c] = e[3 + c]; }
        segs[i] = (signed char)s;
    }

    // rotation reference integrated to the path parameter of stage 1 (BoundMPC.py:894-914)
    {
        const int j = ((int)S[LS_split + 1] == 1) ? 1 : 0;
        double om[3], pr[3];
        for (int c = 0; c < 3; c++) om[c] = S[LS_rp_dpd + LP_S * (3 + c) + j];
        lp_integrate_rot_ref(S + LS_rp_r_tau + 3 * (sec + j), om, S[LS_rp_phi_switch + j], opt_phi[1], pr);
        if (rec) for (int c = 0; c < 3; c++) rec[LP_REC_HDR + 60 + c] = pr[c];      // ref_data["p"][0][3:] = pr_ref (BoundMPC.py:1023)
        for (int c = 0; c < 3; c++) {
            S[LS_pr_ref + c] = pr[c];
            S[LS_iw_ref + c] = S[LS_rp_pd + LP_S * (3 + c) + j] + (opt_phi[1] - S[LS_rp_phi_switch + j]) * om[c];
        }
    }

    // split indices: countdown of a set switch, or detection of a new one (BoundMPC.py:916-1021)
    const double IN_SET = 0.005, ROT_M = 5 * 3.141592653589793 / 180, PHI_M = 0.03;
    bool sw = S[LS_sw] != 0.0;
    for (int i = 1; i < LP_S - 1; i++) {
        if ((int)S[LS_split + i] < N) {
            S[LS_split + i] -= 1.0;
            if ((int)S[LS_split + i] == 0) { sw = true; S[LS_split + i] = N; }
        } else if (ec == 0) {
            const double* a0 = S + LS_rp_a + 45 * (sec + i - 1);
            const double* b0 = S + LS_rp_b + LP_ROWS * (sec + i - 1);
            const double* a1 = S + LS_rp_a + 45 * (sec + i);
            const double* b1 = S + LS_rp_b + LP_ROWS * (sec + i);
            int last_out = -1;
            unsigned long long m_all = 0ull, m_set1 = 0ull;
            for (int kx = 0; kx < n; kx++) {
                const double px = w[28 * N + kx], py = w[29 * N + kx], pz = w[30 * N + kx], ps = w[n_w - 2 * N + kx];
                double d0 = -1e300, d1 = -1e300;
                for (int r = 0; r < LP_ROWS; r++) {
                    d0 = fmax(d0, a0[3 * r] * px + a0[3 * r + 1] * py + a0[3 * r + 2] * pz - b0[r]);
                    d1 = fmax(d1, a1[3 * r] * px + a1[3 * r + 1] * py + a1[3 * r + 2] * pz - b1[r]);
                }
                const bool in0 = d0 < IN_SET + ps, in1 = d1 < IN_SET + ps;
                if (!in1) last_out = kx;
                const int s = segs[kx];
                const double* up = S + LS_rp_erb + 6 * (sec + s);
                const double* upn = S + LS_rp_erb + 6 * (sec + s + 1);
                bool rot = true;
                for (int c = 0; c < 3; c++)
                    rot = rot && (ers[kx][c] < up[c]) && (ers[kx][c] > up[3 + c]) && (ersn[kx][c] < upn[c] + ROT_M) && (ersn[kx][c] > upn[3 + c] - ROT_M);
                const bool dsw = opt_phi[kx] > S[LS_rp_phi_switch + i] - PHI_M;
                if (dsw && in0 && rot) m_all |= 1ull << kx;
                if (in1) m_set1 |= 1ull << kx;
            }
            // only the trailing run of in-set stages counts for the next set
            int idx_new = -1;
            for (int kx = (last_out < 0 ? 0 : last_out); kx < n; kx++)
                if (((m_all >> kx) & 1ull) && ((m_set1 >> kx) & 1ull)) { idx_new = kx; break; }
            const bool not_at_end = sec + (i - 1) < (int)S[LS_rp_num_sectors];
            if (idx_new >= 0 && not_at_end) {
                if ((int)S[LS_split + i] == N) {
                    S[LS_split + i] = idx_new - 1;
                    // move the via point onto the switching position (BoundMPC.py:989-1011)
                    double dp[3], pv[3], corr = 0;
                    for (int c = 0; c < 3; c++) { dp[c] = S[LS_rp_dpd + LP_S * c + i]; pv[c] = S[LS_rp_pd + LP_S * c + i]; }
                    for (int c = 0; c < 3; c++) corr += (w[28 * N + c * N + idx_new] - pv[c]) * dp[c];
                    for (int c = 0; c < 3; c++) {
                        const double nv = pv[c] + corr * dp[c];
                        S[LS_rp_pd + LP_S * c + i] = nv;
                        S[LS_rp_p + 3 * (sec + i) + c] = nv;
                    }
                    S[LS_rp_phi + sec + i + 1] -= corr;
                    for (int q = i + 1; q < LP_S + 1; q++) S[LS_rp_phi_switch + q] -= corr;
                    double cs = 0;
                    for (int q = 0; q <= (int)S[LS_rp_num_sectors] + 1; q++) cs += S[LS_rp_phi + q];
                    S[LS_rp_phi_max] = cs + S[LS_rp_phi_bias];
                    S[LS_phi_max] = S[LS_rp_phi_max];
                }
                if ((int)S[LS_split + i] == 0) sw = true;
            }
        }
    }
    if (sw) {
        for (int i = 1; i < LP_S; i++) S[LS_split + i] = S[LS_split + i + 1];
        S[LS_split + LP_S] = N;
    }
    S[LS_sw] = sw ? 1.0 : 0.0;
    for (int i = 1; i < LP_S; i++)
        if (S[LS_split + i] <= S[LS_split + i - 1]) S[LS_split + i] = fmin((double)N, S[LS_split + i - 1] + 1.0);
    S[LS_phi_current] = opt_phi[1];
    S[LS_dphi_current] = opt_dphi1;

    // integrate_joint over one sampling interval (only the first two jerk hats reach t = dt) with the
    // closed forms of jerk_trajectory_casadi.py:78-175, then MPCNode.step's state advance
    {
        const double h = dt, t = dt;
        Kin k0;
        double J[6][7], G[6][7], vold[6], qn[7], dqn[7], ddqn[7];
        kin_eval(rc, S + LS_q, k0);
        kin_jac(k0, S + LS_dq, J, G, vold);       // v = J(q) dq at the OLD state (util_functions.py:61-62)
        for (int j = 0; j < 7; j++) {
            const double j0 = w[21 * N + j * N + ec], j1 = w[21 * N + j * N + ec + 1];
            double acc = 0, vel = 0, ang = 0;
            acc += -j0 * t * (t - 2 * h) / h / 2;
            vel += -j0 * t * t * (t - 3 * h) / h / 6;
            ang += -j0 * t * t * t * (t - 4 * h) / h / 24;
            acc += j1 * (t * t) / h / 2;
            vel += -j1 * ((0.0 - t) * (0.0 - t) * (0.0 - t)) / h / 6;
            ang += j1 * ((0.0 - t) * (0.0 - t) * (0.0 - t) * (0.0 - t)) / h / 24;
            const double q = S[LS_q + j], dq = S[LS_dq + j], ddq = S[LS_ddq + j];
            qn[j] = ddq * (t * t) / 2 + dq * t + q + ang;
            dqn[j] = ddq * t + dq + vel;
            ddqn[j] = ddq + acc;
        }
        for (int j = 0; j < 7; j++) {
            S[LS_q + j] = qn[j]; S[LS_dq + j] = dqn[j]; S[LS_ddq + j] = ddqn[j];
            S[LS_qf + j] = w[j * N + N - 1];
            S[LS_jerk + j] = w[21 * N + j * N + ec + 1];
        }
        for (int c = 0; c < 6; c++) S[LS_v + c] = vold[c];
        Kin k1;
        kin_eval(rc, qn, k1);
        for (int c = 0; c < 3; c++) S[LS_p_lie + c] = k1.pee[c];
        lp_mat_to_rotvec(k1.Ree, S + LS_p_lie + 3);
    }
    if (rec) rec[6] = S[LS_phi_max];          // after the via-point adaptation of this step
    if (log) {
        log[5] = S[LS_phi_current]; log[6] = S[LS_phi_max]; log[7] = S[LS_split + 1]; log[8] = (double)sec; log[9] = sw ? 1.0 : 0.0;
        for (int c = 0; c < 6; c++) log[10 + c] = S[LS_p_lie + c];
        for (int j = 0; j < 7; j++) log[16 + j] = S[LS_q + j];
    }
}

// host-side layout lookup for the packer (boundplanner_amd/device_loop.py): offset / count of a state field
inline int loop_field_lookup(const char* name, int* off, int* cnt) {
    static const char* const names[LF_COUNT] = {
#define X(n, c) #n,
        LP_FIELDS(X)
#undef X
    };
    for (int f = 0; f < LF_COUNT; f++) {
        const char *a = names[f], *b = name;
        while (*a && *a == *b) { a++; b++; }
        if (!*a && !*b) { *off = ls_off(f); *cnt = LS_CNT[f]; return 0; }
    }
    return 1;
}

}  // namespace bmpc
