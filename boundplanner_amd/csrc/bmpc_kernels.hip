// Device code of libboundmpc_hip.so, compiled once per BMPC_NT (threads = lanes cooperating on ONE
// instance: 64 = one wavefront per instance, 128/256 = 2/4 wavefronts per instance sharing the
// same LDS image).  Exposes plain launch functions to the host-side C ABI in bmpc_capi.hip.
#include "bmpc_platform_hip.hpp"

#ifndef BMPC_NT
#define BMPC_NT 64
#endif

#include "bmpc_solver.hpp"

using namespace bmpc;

#define BMPC_CAT2(a, b) a##b
#define BMPC_CAT(a, b) BMPC_CAT2(a, b)

__global__ __launch_bounds__(BMPC_NT) void BMPC_CAT(bmpc_solve_kernel_nt, BMPC_NT)(KernelArgs A) {
    __shared__ double lds[LDS_DOUBLES];
    LDSD* ldsb = (LDSD*)lds;
    const int lane = BMPC_LANE();
    double* wsb = A.ws + (size_t)BMPC_BLOCK() * ws_doubles(A.o.N);
    // one wavefront per instance; resident workgroups pull the next instance from a device-scope
    // counter (iteration counts vary 7..100, so a static deal leaves most CUs idle at the tail)
    for (;;) {
        if (lane == 0) (ldsb + O_misc)[63] = (double)BMPC_ATOMIC_INC(A.counter);
        BMPC_SYNC();
        int b = (int)(ldsb + O_misc)[63];
        BMPC_SYNC();
        if (b >= A.B) break;
        solve_instance(A, ldsb, wsb, b, lane);
    }
}

#if BMPC_NT == 64
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
#endif


extern "C" hipError_t BMPC_CAT(bmpc_launch_solve_nt, BMPC_NT)(const KernelArgs* A, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL(BMPC_CAT(bmpc_solve_kernel_nt, BMPC_NT), dim3(nblocks), dim3(BMPC_NT), 0, st, *A);
    return hipGetLastError();
}

#if BMPC_NT == 64
extern "C" hipError_t bmpc_launch_fk(int B, const RobotConst* rc, const double* q, const double* dq, double* ee_pos,
                                     double* ee_rot, double* col_pts, double* jac, double* dvdq, hipStream_t st) {
    hipLaunchKernelGGL(bmpc_fk_kernel, dim3((B + 63) / 64), dim3(64), 0, st, B, rc, q, dq, ee_pos, ee_rot, col_pts, jac, dvdq);
    return hipGetLastError();
}
#endif
