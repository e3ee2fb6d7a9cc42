// Host-side construction of the iiwa14 chain constants (iiwa.urdf <joint> origins;
// RobotModel.py:26-35 frames).  Fixed rotations are R = Rz(yaw) Ry(pitch) Rx(roll).
#pragma once
#include <cmath>

#include "bmpc_device.hpp"

namespace bmpc {

inline void rpy_to_R(const double* rpy, double* R) {
    double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]);
    double cp = std::cos(rpy[1]), sp = std::sin(rpy[1]);
    double cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

}  // namespace bmpc
#include "../../include/boundmpc.h"
namespace bmpc {

// RobotModel/iiwa.urdf: <joint> origins (lines 25, 40, 55, 70, 85, 107, 122), joint_ee (137: rpy literally -1.575),
// link4_col (91), <limit>s; RobotModel.py:37 sphere radii; BoundMPC.py:182-186 acceleration / jerk limits
inline void robot_iiwa14(bmpc_robot& r) {
    const double PI_2 = 1.5707963267948966, PI_1 = 3.141592653589793;
    const double xyz[7][3] = {{0, 0, 0.1525}, {0, 0, 0.2075}, {0, 0.2325, 0}, {0, 0, 0.1875},
                              {0, 0.2125, 0}, {0, 0, 0.1875}, {0, 0.0796, 0}};
    const double rpy[7][3] = {{0, 0, 0},        {PI_2, 0, PI_1}, {PI_2, 0, PI_1}, {PI_2, 0, 0},
                              {-PI_2, PI_1, 0}, {PI_2, 0, 0},    {-PI_2, PI_1, 0}};
    const double qlim[7] = {2.9670597283903604, 2.0943951023931953, 2.9670597283903604, 2.0943951023931953,
                            2.9670597283903604, 2.0943951023931953, 3.0543261909900763};
    const double sizes[7] = {0.09, 0.12, 0.09, 0.10, 0.07, 0.09, 0.075};
    for (int i = 0; i < 7; i++) {
        for (int a = 0; a < 3; a++) { r.joint_xyz[i][a] = xyz[i][a]; r.joint_rpy[i][a] = rpy[i][a]; }
        r.q_lower[i] = -qlim[i]; r.q_upper[i] = qlim[i]; r.dq_max[i] = 10.0; r.col_joint_sizes[i] = sizes[i];
    }
    r.ee_xyz[0] = 0; r.ee_xyz[1] = 0; r.ee_xyz[2] = 0.21;
    r.ee_rpy[0] = 0; r.ee_rpy[1] = -1.575; r.ee_rpy[2] = -1.575;
    r.link4_col_xyz[0] = 0; r.link4_col_xyz[1] = 0.3; r.link4_col_xyz[2] = 0;
    r.ddq_max = 5.0; r.u_max = 35.0;
}

// RobotModel/gen3_arm.urdf (Kinova Gen3, USE_IIWA = False): joint origins (lines 28, 42, 56, 70, 84, 105, 119), end_effector
// (127), link4_col (92), <limit>s (joints 1, 3, 5, 7: +-10 in the file, made unlimited by RobotModel.py:46-48); sphere radii
// RobotModel.py:40
inline void robot_gen3(bmpc_robot& r) {
    const double xyz[7][3] = {{0, 0, 0.15643}, {0, 0.005375, -0.12838}, {0, -0.21038, -0.006375}, {0, 0.006375, -0.21038},
                              {0, -0.20843, -0.006375}, {0, 0.00017505, -0.10593}, {0, -0.10593, -0.00017505}};
    const double rpy[7][3] = {{3.1416, 2.7629E-18, -4.9305E-36}, {1.5708, 2.1343E-17, -1.1102E-16}, {-1.5708, 1.2326E-32, -2.9122E-16},
                              {1.5708, -6.6954E-17, -1.6653E-16}, {-1.5708, 2.2204E-16, -6.373E-17}, {1.5708, 9.2076E-28, -8.2157E-15},
                              {-1.5708, -5.5511E-17, 9.6396E-17}};
    const double qlim[7] = {1e20, 2.24, 1e20, 2.57, 1e20, 2.09, 1e20};
    const double vlim[7] = {1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218};
    const double sizes[7] = {0.09, 0.09, 0.06, 0.06, 0.06, 0.06, 0.075};
    for (int i = 0; i < 7; i++) {
        for (int a = 0; a < 3; a++) { r.joint_xyz[i][a] = xyz[i][a]; r.joint_rpy[i][a] = rpy[i][a]; }
        r.q_lower[i] = -qlim[i]; r.q_upper[i] = qlim[i]; r.dq_max[i] = vlim[i]; r.col_joint_sizes[i] = sizes[i];
    }
    r.ee_xyz[0] = 0; r.ee_xyz[1] = 0; r.ee_xyz[2] = -0.20;
    r.ee_rpy[0] = 0; r.ee_rpy[1] = 1.570796326794895; r.ee_rpy[2] = 1.570796326794895;
    r.link4_col_xyz[0] = 0; r.link4_col_xyz[1] = -0.1; r.link4_col_xyz[2] = 0;
    r.ddq_max = 5.0; r.u_max = 35.0;
}

inline void fill_robot_const(RobotConst& rc, const bmpc_robot& r) {
    for (int i = 0; i < 7; i++) {
        for (int a = 0; a < 3; a++) rc.jxyz[i][a] = r.joint_xyz[i][a];
        rpy_to_R(r.joint_rpy[i], rc.jrot[i]);
        rc.q_lo[i] = r.q_lower[i]; rc.q_hi[i] = r.q_upper[i]; rc.dq_max[i] = r.dq_max[i];
    }
    for (int a = 0; a < 3; a++) { rc.ee_xyz[a] = r.ee_xyz[a]; rc.l4c_xyz[a] = r.link4_col_xyz[a]; }
    rpy_to_R(r.ee_rpy, rc.ee_rot);
    rc.ddq_max = r.ddq_max; rc.u_max = r.u_max;
    for (int c = 0; c < 6; c++) rc.colsize[c] = r.col_joint_sizes[c];
}
inline void fill_robot_const(RobotConst& rc) {      // the default robot
    bmpc_robot r;
    robot_iiwa14(r);
    fill_robot_const(rc, r);
}

}  // namespace bmpc
