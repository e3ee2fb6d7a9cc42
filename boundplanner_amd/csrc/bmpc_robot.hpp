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

inline void fill_robot_const(RobotConst& rc) {
    const double PI_2 = 1.5707963267948966, PI_1 = 3.141592653589793;
    const double xyz[7][3] = {{0, 0, 0.1525}, {0, 0, 0.2075}, {0, 0.2325, 0}, {0, 0, 0.1875},
                              {0, 0.2125, 0}, {0, 0, 0.1875}, {0, 0.0796, 0}};
    const double rpy[7][3] = {{0, 0, 0},        {PI_2, 0, PI_1}, {PI_2, 0, PI_1}, {PI_2, 0, 0},
                              {-PI_2, PI_1, 0}, {PI_2, 0, 0},    {-PI_2, PI_1, 0}};
    for (int i = 0; i < 7; i++) {
        for (int a = 0; a < 3; a++) rc.jxyz[i][a] = xyz[i][a];
        rpy_to_R(rpy[i], rc.jrot[i]);
    }
    const double ee_rpy[3] = {0, -1.575, -1.575};   // iiwa.urdf:137 (literally -1.575)
    rc.ee_xyz[0] = 0; rc.ee_xyz[1] = 0; rc.ee_xyz[2] = 0.21;
    rpy_to_R(ee_rpy, rc.ee_rot);
    rc.l4c_xyz[0] = 0; rc.l4c_xyz[1] = 0.3; rc.l4c_xyz[2] = 0;
}

}  // namespace bmpc
