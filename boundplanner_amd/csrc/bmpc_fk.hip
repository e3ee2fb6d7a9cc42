// Batched forward kinematics (include/boundmpc.h bmpc_fk): one thread per configuration.
#include "bmpc_platform_hip.hpp"

#include "bmpc_device.hpp"

using namespace bmpc;

__global__ void bmpc_fk_kernel(int B, const RobotConst* rc, const double* q, const double* dq, double* ee_pos,
                               double* ee_rot, double* col_pts, double* jac, double* dvdq) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double qq[7], dd[7];
    for (int j = 0; j < 7; j++) { qq[j] = q[(size_t)b * 7 + j]; dd[j] = dq ? dq[(size_t)b * 7 + j] : 0.0; }
    Kin k;
    double J[6][7], G[6][7], v[6];
    kin_eval(rc, qq, k);
    kin_jac(k, dd, J, G, v);
    if (ee_pos) for (int a = 0; a < 3; a++) ee_pos[(size_t)b * 3 + a] = k.pee[a];
    if (ee_rot) for (int a = 0; a < 9; a++) ee_rot[(size_t)b * 9 + a] = k.Ree[a];
    if (col_pts) for (int c = 0; c < 6; c++) for (int a = 0; a < 3; a++) col_pts[(size_t)b * 18 + 3 * c + a] = k.pc[c][a];
    if (jac) for (int a = 0; a < 6; a++) for (int j = 0; j < 7; j++) jac[(size_t)b * 42 + 7 * a + j] = J[a][j];
    if (dvdq) for (int a = 0; a < 6; a++) for (int j = 0; j < 7; j++) dvdq[(size_t)b * 42 + 7 * a + j] = G[a][j];
}

extern "C" hipError_t bmpc_launch_fk(int B, const RobotConst* rc, const double* q, const double* dq, double* ee_pos,
                                     double* ee_rot, double* col_pts, double* jac, double* dvdq, hipStream_t st) {
    hipLaunchKernelGGL(bmpc_fk_kernel, dim3((B + 63) / 64), dim3(64), 0, st, B, rc, q, dq, ee_pos, ee_rot, col_pts, jac, dvdq);
    return hipGetLastError();
}

// Diagnostic: occupies the stream for `ms` milliseconds (at most 10 s, then it ends by itself -- every wave reaches the exit) so
// that the watchdog of the solve entry points can be tested without a kernel that really hangs.
__global__ void bmpc_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();                    // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" hipError_t bmpc_launch_spin(int ms, hipStream_t st) {
    if (ms > 10000) ms = 10000;
    if (ms < 0) ms = 0;
    hipLaunchKernelGGL(bmpc_spin_kernel, dim3(1), dim3(64), 0, st, (long long)ms * 100000LL);
    return hipGetLastError();
}
