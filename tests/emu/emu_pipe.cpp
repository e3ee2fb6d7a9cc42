// CPU thread emulation of the batch-synchronous HIP pipeline -- TEST INFRASTRUCTURE ONLY.
// Compiles the identical device source (bmpc_stage.hpp, bmpc_pair_kernels.hpp,
// bmpc_ric_kernel.hpp) with 64 std::threads standing in for the 64 lanes of a wavefront and a
// barrier for __syncthreads(); workgroups run one after the other.  Lets the kernel logic be
// debugged against the oracle without a GPU.  Never shipped, never used by the product path.
#include <barrier>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static std::barrier<>* g_bar = nullptr;
static thread_local int t_lane = 0;
#define BMPC_EMU_TRACE 1
#define BMPC_DEV inline
#define BMPC_INL inline
#define BMPC_KBODY inline
#define BMPC_PIN(x) do {} while (0)
#define BMPC_UNIFORM(x) (x)
#define BMPC_OPAQUE_I(x) do {} while (0)
#define BMPC_TOUCH_LINE(g, l) do {} while (0)
#define BMPC_HD inline
#define BMPC_NOINL
typedef double LDSD;
typedef double bmpc_v2d __attribute__((vector_size(16)));
typedef bmpc_v2d LDSV2;
#define BMPC_RSQRT(x) (1.0 / std::sqrt(x))
#define BMPC_RCP(x) (1.0 / (x))
#define BMPC_MUL24(a, b) ((a) * (b))
#define BMPC_SCHED_FENCE() do {} while (0)
#define BMPC_SINCOS(x, s, c) do { (s) = std::sin(x); (c) = std::cos(x); } while (0)
#define BMPC_LDS_ADD(ptr, v) (*(ptr) += (v))
#define BMPC_AS1
template <int NCH, int NT> static inline void bmpc_async_copy(const double* gsrc, double* lds_dst, int lane) {
    const int wave = lane >> 6, wl = lane & 63;
    for (int i = 0; i < (NCH + NT / 64 - 1) / (NT / 64); i++) {
        const int c = i * (NT / 64) + wave;
        if (c < NCH) { lds_dst[128 * c + 2 * wl] = gsrc[128 * c + 2 * wl]; lds_dst[128 * c + 2 * wl + 1] = gsrc[128 * c + 2 * wl + 1]; }
    }
}
#define BMPC_ASYNC_WAIT() do {} while (0)
#define BMPC_SYNC() g_bar->arrive_and_wait()
#define BMPC_FENCE_SYNC() g_bar->arrive_and_wait()
#define BMPC_LANE() t_lane
#define BMPC_NT 64
#define BMPC_BLOCK() 0
#define BMPC_NBLOCKS() 1
#define BMPC_ATOMIC_INC(ptr) __atomic_fetch_add((ptr), 1, __ATOMIC_RELAXED)
using std::fmax;
using std::fmin;

#include "../../boundplanner_amd/csrc/bmpc_pair_kernels.hpp"
#include "../../boundplanner_amd/csrc/bmpc_ric_kernel.hpp"
#include "../../boundplanner_amd/csrc/bmpc_robot.hpp"

using namespace bmpc;
#ifndef EMU_TRIAL_NW
#define EMU_TRIAL_NW 1
#endif
#ifndef EMU_RIC_NT
#define EMU_RIC_NT 128
#endif

template <class F> static void launch(int nblocks, F body, int nt = 64) {
    if (nblocks <= 0) return;
    std::barrier<> bar(nt);
    g_bar = &bar;
    std::vector<std::thread> th;
    for (int l = 0; l < nt; l++)
        th.emplace_back([&, l] {
            t_lane = l;
            // workgroups run one after the other on the same LDS buffer: nobody starts the next one's LDS writes
            // while a slower lane still reads this one's
            for (int blk = 0; blk < nblocks; blk++) { body(blk, l); bar.arrive_and_wait(); }
        });
    for (auto& t : th) t.join();
}

extern "C" int emu_pipe_solve(int N, double dt, double tol, int max_iter, int hess, double hess_switch, double mu_init,
                              double kappa_mu, double theta_mu, double kappa_eps, int B, const double* x0,
                              const double* lbx, const double* ubx, const double* p, double* x, double* g, double* f,
                              int* iters, int* status, double* viol, int verbose, double* lam_g, double* lam_x, int slots,
                              double mu_floor_k, double dw0, double inertia_err, int inertia, int stall_n, int gn_backoff, int slack_reset, double ls_alpha_mem, int trial_repeats) {
    RobotConst rc;
    fill_robot_const(rc);
    PipeArgs A;   // emulation: BMPC_AS1 is empty, host and device views coincide
    A.B = B; A.N = N; A.natt = 0; A.pad0_ = 0;
    A.lam_g = nullptr; A.lam_x = nullptr; A.cont = nullptr;
    A.o = SolverOpts{N, dt, tol, max_iter, hess, hess_switch, mu_init, kappa_mu, theta_mu, kappa_eps, mu_floor_k, dw0, inertia_err, ls_alpha_mem, inertia, stall_n, gn_backoff, slack_reset, trial_repeats};
    A.rc = &rc;
    A.x0 = x0; A.lbx = lbx; A.ubx = ubx; A.p = p;
    A.x = x; A.f = f; A.viol = viol; A.g = g; A.iters = iters; A.status = status;
    const int cap = (slots > 0 && slots < B) ? slots : B;       // pool of `cap` slots: B > cap streams through it
    const char* lay = getenv("BMPC_LAYOUT");
    const int slot_major = lay ? atoi(lay) : 1;
    std::vector<double> work(pipe_workspace_doubles(cap, N, slot_major));
    pipe_carve(A, work.data(), cap, N, slot_major);
    std::vector<InstState> st(cap);
    std::vector<int> l_eval(cap), l_step(cap), l_trial(cap), l_evn(cap), l_trn(cap), l_done(cap), l_admit(cap), l_curv(cap), srcv(cap), cnt(NCNT, 0), tbl(3 * HREC);
    build_scatter_table(tbl.data());
    A.st = st.data();
    A.L.eval = l_eval.data(); A.L.step = l_step.data(); A.L.trial = l_trial.data();
    A.L.eval_next = l_evn.data(); A.L.trial_next = l_trn.data(); A.L.cnt = cnt.data();
    A.L.done = l_done.data(); A.L.admit = l_admit.data(); A.L.curv = l_curv.data(); A.src = srcv.data();
    A.tbl = tbl.data();
    std::vector<double> lds(std::max<size_t>(std::max<size_t>(pair_lds_doubles(N, true), trial_lds_doubles(N, 4)), RIC_LDS_DOUBLES) + 64);
    const int n0 = cap;
    const int nb_inst = (cap + 63) / 64, nw = waves_for(N, cap);
    cnt[0] = n0; cnt[6] = n0; cnt[9] = n0;
    launch(nb_inst, [&](int blk, int l) { k_init_inst_body(A, blk * 64 + l); });
    launch(nw, [&](int blk, int l) { k_init_body(A, blk, l, lds.data()); });
    launch(nb_inst, [&](int blk, int l) { k_init_fin_body(A, blk * 64 + l); });
    k_pool_reset_body(A, false);
    auto retire = [&]() {        // bmpc_pipe_launch_retire
        launch(waves_for(N, cnt[8]), [&](int blk, int l) { k_out_body(A, blk, l, lds.data()); });
        launch((cnt[8] + 63) / 64, [&](int blk, int l) { k_fin_body(A, blk * 64 + l); });
        launch((cnt[8] + 63) / 64, [&](int blk, int l) { k_admit_body(A, blk * 64 + l); });
        launch(waves_for(N, cnt[9]), [&](int blk, int l) { k_init_body(A, blk, l, lds.data()); });
        launch((cnt[9] + 63) / 64, [&](int blk, int l) { k_init_fin_body(A, blk * 64 + l); });
        k_pool_reset_body(A, true);
    };
    int steps = 0;
    for (; steps < 12 * (max_iter + 2) * ((B + cap - 1) / cap + 1); steps++) {
        retire();
        if (cnt[7] >= B) break;
        if (verbose) printf("step %d: n_eval %d n_trial %d finished %d retired %d next row %d\n", steps, cnt[0], cnt[2], cnt[5], cnt[7], cnt[6]);
        launch(waves_for(N, cnt[0]), [&](int blk, int l) { k_points_body(A, blk, l, lds.data()); });
        launch(waves_for(N, cnt[0]), [&](int blk, int l) { k_pose_body(A, blk, l, lds.data()); });
        if (getenv("BMPC_EMU_EVAL_SPLIT") && atoi(getenv("BMPC_EMU_EVAL_SPLIT"))) {      // the tail regime's two-wavefront k_eval on the GPU
            launch(waves_for(N, cnt[0]), [&](int blk, int l) { k_eval_body<2>(A, blk, l, lds.data()); });
            launch(waves_for(N, cnt[0]), [&](int blk, int l) { k_eval_body<1>(A, blk, l, lds.data()); });
        } else
        launch(waves_for(N, cnt[0]), [&](int blk, int l) { k_eval_body<0>(A, blk, l, lds.data()); });
        launch(waves_for(N, cnt[10]), [&](int blk, int l) { k_curv_body(A, blk, l, lds.data()); });
        if (getenv("BMPC_EMU_RIC_SPEC") && atoi(getenv("BMPC_EMU_RIC_SPEC"))) {      // the speculative pair (the deep tail's kernels on the GPU)
            A.natt = 3;
            launch(cnt[0] * A.natt, [&](int blk, int l) { k_ric_att_body<EMU_RIC_NT, false>(*reinterpret_cast<const PipeArgsH*>(&A), blk, l, lds.data()); }, EMU_RIC_NT);
            launch(cnt[0], [&](int blk, int l) { k_ric_body<EMU_RIC_NT, false, true>(*reinterpret_cast<const PipeArgsH*>(&A), blk, l, lds.data()); }, EMU_RIC_NT);
        } else
        launch(cnt[0], [&](int blk, int l) { k_ric_body<EMU_RIC_NT>(*reinterpret_cast<const PipeArgsH*>(&A), blk, l, lds.data()); }, EMU_RIC_NT);
        launch(cnt[1], [&](int blk, int l) { k_fwd_body(A, blk, l, lds.data()); });
        launch(waves_for(N, cnt[1]), [&](int blk, int l) { k_step_body(A, blk, l, lds.data()); });
        if (getenv("BMPC_EMU_TRIAL_SPEC") && atoi(getenv("BMPC_EMU_TRIAL_SPEC")) && slot_major)      // the tail regime's line search on the GPU
            launch(waves_for(N, cnt[2]), [&](int blk, int l) { k_trial_spec_body(A, blk, l, lds.data()); }, 64 * TRIAL_SPEC);
        else
        launch(waves_for(N, cnt[2]), [&](int blk, int l) { k_trial_body_t<EMU_TRIAL_NW>(A, blk, l, lds.data()); }, 64 * EMU_TRIAL_NW);
        k_rotate_body(A);
        std::swap(A.L.eval, A.L.eval_next);
        std::swap(A.L.trial, A.L.trial_next);
    }
    if (lam_g && lam_x) {
        A.lam_g = lam_g; A.lam_x = lam_x;
        launch(nw, [&](int blk, int l) { k_mult_body(A, blk, l, lds.data()); });
        launch(nb_inst, [&](int blk, int l) { k_mult_sweep_body(A, blk * 64 + l); });
    }
    return steps;
}
