// CPU thread emulation of ONE workgroup of the HIP solver -- TEST INFRASTRUCTURE ONLY.
// Compiles the identical device source (boundplanner_amd/csrc/bmpc_device.hpp, bmpc_solver.hpp)
// with 64 std::threads standing in for the 64 lanes and a barrier for __syncthreads(), so the
// kernel logic can be debugged and compared with the oracle without a GPU.  Never shipped,
// never used by the product path or by the -m gpu tests.
#include <barrier>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static std::barrier<>* g_bar = nullptr;
static thread_local int t_lane = 0;
#define BMPC_DEV
#define BMPC_INL inline
#define BMPC_KBODY inline
#define BMPC_PIN(x) do {} while (0)
#define BMPC_UNIFORM(x) (x)
#define BMPC_OPAQUE_I(x) do {} while (0)
#define BMPC_TOUCH_LINE(g, l) do {} while (0)
#define BMPC_HD inline
#define BMPC_NOINL
typedef double LDSD;
#define BMPC_AS1
#define BMPC_SYNC() g_bar->arrive_and_wait()
#define BMPC_LANE() t_lane
#ifndef BMPC_NT
#define BMPC_NT 64
#endif
#define BMPC_BLOCK() 0
#define BMPC_NBLOCKS() 1
#define BMPC_ATOMIC_INC(ptr) __atomic_fetch_add((ptr), 1, __ATOMIC_RELAXED)
using std::fmax;
using std::fmin;

#include "../../boundplanner_amd/csrc/bmpc_solver.hpp"
#include "../../boundplanner_amd/csrc/bmpc_robot.hpp"

extern "C" int emu_solve(int N, double dt, double tol, int max_iter, int hess, double hess_switch, double mu_init,
                         double kappa_mu, double theta_mu, double kappa_eps, const double* x0, const double* lbx,
                         const double* ubx, const double* p, double* x, double* f, int* iters, int* status,
                         double* viol) {
    using namespace bmpc;
    RobotConst rc;
    fill_robot_const(rc);
    KernelArgs A;
    A.B = 1;
    A.o = SolverOpts{N, dt, tol, max_iter, hess, hess_switch, mu_init, kappa_mu, theta_mu, kappa_eps};
    A.rc = &rc;
    A.x0 = x0; A.lbx = lbx; A.ubx = ubx; A.p = p;
    A.x = x; A.f = f; A.viol = viol; A.g = nullptr; A.iters = iters; A.status = status;
    std::vector<double> ws(ws_doubles(N), 0.0), lds(LDS_DOUBLES + 64, 0.0);
    A.ws = ws.data();
    int counter = 0;
    A.counter = &counter;
    A.prof = nullptr;
    std::barrier<> bar(BMPC_NT);
    g_bar = &bar;
    std::vector<std::thread> th;
    for (int l = 0; l < BMPC_NT; l++)
        th.emplace_back([&, l] { t_lane = l; solve_instance(A, lds.data(), A.ws, 0, l); });
    for (auto& t : th) t.join();
    return 0;
}
