// CPU build of the device-resident closed-loop step logic (boundplanner_amd/csrc/bmpc_loop.hpp) --
// TEST INFRASTRUCTURE ONLY: lets tests/test_device_loop.py replay the reference's closed-loop trace
// through the identical source the HIP kernels compile, without a GPU.  Never shipped, never used by
// the product path.
#include <cmath>
#include <cstring>
#include <vector>
#define BMPC_DEV inline
#define BMPC_INL inline
#define BMPC_HD inline
#define BMPC_NOINL
typedef double LDSD;
#define BMPC_SYNC() do {} while (0)
#define BMPC_LANE() 0
#define BMPC_NT 64
#define BMPC_BLOCK() 0
#define BMPC_NBLOCKS() 1
#define BMPC_AS1
#define BMPC_SCHED_FENCE() do {} while (0)
#define BMPC_SINCOS(x, s, c) do { (s) = std::sin(x); (c) = std::cos(x); } while (0)
using std::fmax;
using std::fmin;
#include "../../boundplanner_amd/csrc/bmpc_loop.hpp"
#include "../../boundplanner_amd/csrc/bmpc_robot.hpp"

using namespace bmpc;

extern "C" int emu_loop_state_doubles(void) { return LS_SIZE; }
extern "C" int emu_loop_logw(void) { return LP_LOGW; }
extern "C" int emu_loop_field(const char* name, int* off, int* cnt) { return loop_field_lookup(name, off, cnt); }

// one rollout: state S, previous solution row prev (n_w), outputs x0/lbx/ubx (n_w) and p (875)
extern "C" void emu_loop_prepare(int N, double* S, const double* prev, double* x0, double* lbx, double* ubx, double* p) {
    RobotConst rc;
    fill_robot_const(rc);
    const int n_w = 44 * N + 6;
    for (int i = 0; i < n_w; i++) loop_bound_const(&rc, N, i, lbx + i, ubx + i);
    loop_prepare(&rc, N, S, prev, p, lbx, ubx);
    for (int i = 0; i < n_w; i++) x0[i] = loop_x0_elem(N, S, prev, i);
}

// the same with scene obstacles: A [n_obs][15][3], b [n_obs][15], nrows, V [n_obs][32][3], nv (layout of bmpc_loop_set_obstacles)
extern "C" void emu_loop_prepare_obs(int N, double* S, const double* prev, double* x0, double* lbx, double* ubx, double* p, int n_obs,
                                     const double* A, const double* b, const int* nrows, const double* V, const int* nv) {
    RobotConst rc;
    fill_robot_const(rc);
    const int n_w = 44 * N + 6;
    std::vector<double> AAt((size_t)n_obs * LP_ROWS * LP_ROWS, 0.0), colres((size_t)6 * n_obs * LP_CRES, 0.0);
    for (int o = 0; o < n_obs; o++)
        for (int r = 0; r < nrows[o]; r++)
            for (int q = 0; q < nrows[o]; q++) {
                double s = 0;
                for (int c = 0; c < 3; c++) s += A[45 * o + 3 * r + c] * A[45 * o + 3 * q + c];
                AAt[(size_t)LP_ROWS * LP_ROWS * o + LP_ROWS * r + q] = s;
            }
    std::vector<double> box((size_t)n_obs * 6, 0.0);
    std::vector<int> is_box(n_obs, 0);
    for (int o = 0; o < n_obs; o++) is_box[o] = loop_detect_box(A + 45 * o, b + LP_ROWS * o, nrows[o], box.data() + 6 * o, box.data() + 6 * o + 3) ? 1 : 0;
    LoopScene sc{n_obs, A, b, AAt.data(), nrows, V, nv, box.data(), is_box.data()};
    for (int pt = 0; pt < 6; pt++)
        for (int ob = 0; ob < n_obs; ob++) loop_collision_pair(&rc, sc, S, pt, ob, colres.data() + (size_t)(pt * n_obs + ob) * LP_CRES);
    for (int i = 0; i < n_w; i++) loop_bound_const(&rc, N, i, lbx + i, ubx + i);
    loop_prepare(&rc, N, S, prev, p, lbx, ubx, &sc, colres.data());
    for (int i = 0; i < n_w; i++) x0[i] = loop_x0_elem(N, S, prev, i);
}

extern "C" void emu_loop_finish(int N, double dt, double* S, const double* x, double* prev, int status, double viol,
                                int iters, double* log, double* rec, const double* par) {
    RobotConst rc;
    fill_robot_const(rc);
    loop_finish(&rc, N, dt, S, x, prev, status, viol, iters, log, rec, par);
    if (S[LS_accept] != 0.0) std::memcpy(prev, x, sizeof(double) * (44 * N + 6));
}
extern "C" int emu_loop_record_doubles(int N) { return lp_rec_doubles(N); }

extern "C" void emu_so3(const double* v, const double* M, double* R_of_v, double* v_of_M, double* eul_of_M) {
    lp_rotvec_to_mat(v, R_of_v);
    lp_mat_to_rotvec(M, v_of_M);
    lp_euler_zyx(M, eul_of_M);
}
