/* ORACLE (test infrastructure).  iiwa14 kinematics restated from the URDF constants
 * (/root/reference/bound_planner/RobotModel/iiwa.urdf <joint> origins, revolute about local z)
 * with the frame choices of RobotModel.py:26-35 (end_effector_link; collision points =
 * origins of joint_3..joint_7 and frame link4_col_link).  Pinned to the reference's
 * serialized CasADi tapes (fk_pos.ca, fk_pos_col_*.ca, hom_trans.ca, jacobian.ca) by
 * tests/test_oracle_kinematics.py. */
#include <math.h>
#include <string.h>

#include "bmpc_internal.h"

#define PI_2 1.5707963267948966
#define PI_1 3.141592653589793

/* joint_1..joint_7 <origin xyz rpy> (iiwa.urdf:25,40,55,70,85,107,122) */
static double JXYZ[7][3] = {{0, 0, 0.1525}, {0, 0, 0.2075}, {0, 0.2325, 0}, {0, 0, 0.1875},
                                  {0, 0.2125, 0}, {0, 0, 0.1875}, {0, 0.0796, 0}};
static double JRPY[7][3] = {{0, 0, 0},        {PI_2, 0, PI_1}, {PI_2, 0, PI_1}, {PI_2, 0, 0},
                                  {-PI_2, PI_1, 0}, {PI_2, 0, 0},    {-PI_2, PI_1, 0}};
/* joint_ee (iiwa.urdf:137): rpy is literally -1.575 (not -pi/2) */
static double EE_XYZ[3] = {0, 0, 0.21};
static double EE_RPY[3] = {0, -1.575, -1.575};
/* link4_col (iiwa.urdf:91) */
static double L4C_XYZ[3] = {0, 0.3, 0};

/* another 7-joint arm with the same frame conventions (the reference's USE_IIWA = False branch, RobotModel.py:10-48: the
 * Kinova Gen3 of gen3_arm.urdf): process-wide, set before solving; NULL restores the iiwa14 */
void bmpc_oracle_set_robot(const double* joint_xyz21, const double* joint_rpy21, const double* ee_xyz, const double* ee_rpy,
                           const double* link4_col_xyz) {
    static const double X0[7][3] = {{0, 0, 0.1525}, {0, 0, 0.2075}, {0, 0.2325, 0}, {0, 0, 0.1875}, {0, 0.2125, 0}, {0, 0, 0.1875}, {0, 0.0796, 0}};
    static const double R0[7][3] = {{0, 0, 0}, {PI_2, 0, PI_1}, {PI_2, 0, PI_1}, {PI_2, 0, 0}, {-PI_2, PI_1, 0}, {PI_2, 0, 0}, {-PI_2, PI_1, 0}};
    static const double E0[3] = {0, 0, 0.21}, ER0[3] = {0, -1.575, -1.575}, L0[3] = {0, 0.3, 0};
    for (int i = 0; i < 7; i++)
        for (int a = 0; a < 3; a++) {
            JXYZ[i][a] = joint_xyz21 ? joint_xyz21[3 * i + a] : X0[i][a];
            JRPY[i][a] = joint_rpy21 ? joint_rpy21[3 * i + a] : R0[i][a];
        }
    for (int a = 0; a < 3; a++) {
        EE_XYZ[a] = joint_xyz21 ? ee_xyz[a] : E0[a];
        EE_RPY[a] = joint_xyz21 ? ee_rpy[a] : ER0[a];
        L4C_XYZ[a] = joint_xyz21 ? link4_col_xyz[a] : L0[a];
    }
}

const int BMPC_COL_NJ[6] = {2, 3, 4, 5, 6, 4};

static void rpy_to_R(const double rpy[3], double R[9]) {
    double cr = cos(rpy[0]), sr = sin(rpy[0]);
    double cp = cos(rpy[1]), sp = sin(rpy[1]);
    double cy = cos(rpy[2]), sy = sin(rpy[2]);
    /* R = Rz(yaw) Ry(pitch) Rx(roll) */
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

static void mat3_vec(const double A[9], const double v[3], double r[3]) {
    for (int i = 0; i < 3; i++) r[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

void bmpc_cross(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

void bmpc_kin_eval(const double q[7], bmpc_kin* k) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double t[3] = {0, 0, 0};
    double Rf[9], Rn[9], tmp[3];
    for (int i = 0; i < 7; i++) {
        /* fixed part of joint i */
        mat3_vec(R, JXYZ[i], tmp);
        for (int a = 0; a < 3; a++) t[a] += tmp[a];
        rpy_to_R(JRPY[i], Rf);
        mat3_mul(R, Rf, Rn);
        for (int a = 0; a < 3; a++) {
            k->o[i][a] = t[a];
            k->z[i][a] = Rn[3 * a + 2];
        }
        /* rotation about local z by q_i */
        double c = cos(q[i]), s = sin(q[i]);
        double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
        mat3_mul(Rn, Rz, R);
        if (i == 3) { /* link_4 frame -> link4_col_link */
            mat3_vec(R, L4C_XYZ, tmp);
            for (int a = 0; a < 3; a++) k->pc[5][a] = t[a] + tmp[a];
        }
    }
    mat3_vec(R, EE_XYZ, tmp);
    for (int a = 0; a < 3; a++) k->pee[a] = t[a] + tmp[a];
    rpy_to_R(EE_RPY, Rf);
    mat3_mul(R, Rf, k->Ree);
    for (int c = 0; c < 5; c++)
        for (int a = 0; a < 3; a++) k->pc[c][a] = k->o[c + 2][a];
}

/* Jp[:, i] = z_i x (pt - o_i) for the first nj joints, 0 otherwise */
void bmpc_kin_point_jac(const bmpc_kin* k, const double pt[3], int nj, double Jp[3][7]) {
    for (int i = 0; i < 7; i++) {
        double r[3], c[3] = {0, 0, 0};
        if (i < nj) {
            for (int a = 0; a < 3; a++) r[a] = pt[a] - k->o[i][a];
            bmpc_cross(k->z[i], r, c);
        }
        for (int a = 0; a < 3; a++) Jp[a][i] = c[a];
    }
}

/* geometric Jacobian of the end effector, LOCAL_WORLD_ALIGNED (RobotModel.py:213-231) */
void bmpc_kin_jac(const bmpc_kin* k, double J[6][7]) {
    double Jp[3][7];
    bmpc_kin_point_jac(k, k->pee, 7, Jp);
    for (int i = 0; i < 7; i++)
        for (int a = 0; a < 3; a++) {
            J[a][i] = Jp[a][i];
            J[3 + a][i] = k->z[i][a];
        }
}

/* G = d(J(q) dq)/dq.  d2p/dq_i dq_j = z_min x c_max with c_j = J_lin[:, j];
 * dz_j/dq_i = z_i x z_j for i < j. */
void bmpc_kin_dvdq(const bmpc_kin* k, const double J[6][7], const double dq[7], double G[6][7]) {
    double suf_c[8][3], pre_z[8][3], suf_z[8][3];
    memset(suf_c, 0, sizeof suf_c);
    memset(pre_z, 0, sizeof pre_z);
    memset(suf_z, 0, sizeof suf_z);
    for (int j = 6; j >= 0; j--)
        for (int a = 0; a < 3; a++) {
            suf_c[j][a] = suf_c[j + 1][a] + J[a][j] * dq[j];   /* sum_{j>=i} c_j dq_j */
            suf_z[j][a] = suf_z[j + 1][a] + k->z[j][a] * dq[j]; /* sum_{j>=i} z_j dq_j */
        }
    for (int j = 0; j < 7; j++)
        for (int a = 0; a < 3; a++) pre_z[j + 1][a] = pre_z[j][a] + k->z[j][a] * dq[j]; /* sum_{j<i+1} */
    for (int i = 0; i < 7; i++) {
        double c_i[3] = {J[0][i], J[1][i], J[2][i]};
        double t1[3], t2[3], t3[3];
        bmpc_cross(k->z[i], suf_c[i], t1);   /* z_i x sum_{j>=i} c_j dq_j */
        bmpc_cross(pre_z[i], c_i, t2);       /* (sum_{j<i} z_j dq_j) x c_i */
        bmpc_cross(k->z[i], suf_z[i + 1], t3); /* z_i x sum_{j>i} z_j dq_j */
        for (int a = 0; a < 3; a++) {
            G[a][i] = t1[a] + t2[a];
            G[3 + a][i] = t3[a];
        }
    }
}

void bmpc_oracle_fk(const double* q, const double* dq, double* ee_pos, double* ee_rot,
                    double* col_pts, double* jac, double* dvdq) {
    bmpc_kin k;
    double J[6][7], G[6][7];
    bmpc_kin_eval(q, &k);
    bmpc_kin_jac(&k, J);
    if (ee_pos) memcpy(ee_pos, k.pee, sizeof k.pee);
    if (ee_rot) memcpy(ee_rot, k.Ree, sizeof k.Ree);
    if (col_pts) memcpy(col_pts, k.pc, sizeof k.pc);
    if (jac) memcpy(jac, J, sizeof J);
    if (dvdq && dq) {
        bmpc_kin_dvdq(&k, J, dq, G);
        memcpy(dvdq, G, sizeof G);
    }
}
