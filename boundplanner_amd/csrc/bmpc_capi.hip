// libboundmpc_hip.so: C ABI (include/boundmpc.h) + kernel launches for gfx950.
#include "bmpc_platform_hip.hpp"

#define BMPC_NT 64
#include "bmpc_pipeline.hpp"
#include "bmpc_robot.hpp"

#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/boundmpc.h"

using namespace bmpc;

extern "C" hipError_t bmpc_launch_fk(int B, const RobotConst* rc, const double* q, const double* dq, double* ee_pos,
                                     double* ee_rot, double* col_pts, double* jac, double* dvdq, hipStream_t st);

extern "C" hipError_t bmpc_launch_spin(int ms, hipStream_t st);

extern "C" hipError_t bmpc_pipe_launch_init(const PipeArgsH* A, int n0, hipStream_t st);
extern "C" hipError_t bmpc_pipe_launch_retire_out(const PipeArgsH* A, int n_max, hipStream_t st);
extern "C" hipError_t bmpc_pipe_launch_retire_admit(const PipeArgsH* A, int n_max, int refill, hipStream_t st);
// closed loop: called between the two halves of a retirement with the list of slots whose instances have just retired
// (device pointers: list, its length); enqueues the caller's post-processing / next-problem kernels on the stream
typedef int (*bmpc_retire_hook)(void* ctx, const int* d_done, const int* d_n_done, int n_max, void* stream);
extern "C" hipError_t bmpc_pipe_launch_step(PipeArgsH* A, int n_act, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int* was_lat);
extern "C" hipError_t bmpc_pipe_launch_pick(PipeArgsH* A0, PipeArgsH* A1, const int* prio, int n_max, hipStream_t st);
extern "C" hipError_t bmpc_pipe_launch_mult(const PipeArgsH* A, hipStream_t st);
extern "C" void bmpc_pipe_build_table(int* tbl);
extern "C" size_t bmpc_pipe_state_bytes(void);

struct bmpc_handle {
    bmpc_opts o;
    int n_w, n_g, n_cu, nblocks_max;
    RobotConst* d_rc = nullptr;
    bmpc_robot robot;              // host copy of the robot table behind d_rc
    double* d_prof = nullptr;   // diagnostic builds only
    // workspace (grown on demand to the largest batch seen)
    int pipe_cap = 0;
    // workspace layout (pipe_carve): slot-major; BMPC_LAYOUT=0 in the environment selects the field-major layout of round 1 (A/B runs)
    int slot_major = [] { const char* e = getenv("BMPC_LAYOUT"); return e ? atoi(e) : 1; }();
    double* d_pipe = nullptr;      // one slab: SoA iterate/row arrays, stage records, gains, partials
    void* d_pipe_st = nullptr;     // InstState[cap]
    int* d_pipe_lists = nullptr;   // 8 lists + the slot -> row map of cap ints each + NCNT counters; then the same block again for
                                   // the fast lane of the closed loop without lock step (pipe_solve)
    int* d_pipe_tbl = nullptr;     // scatter table of the stage record
    int* h_cnt = nullptr;          // pinned host copy of the counters (2 x NCNT: bulk lane, fast lane)
    // closed loop without lock step, two lanes: streams of the fast lane / of the bulk lane (null: the caller's stream), fork / join events
    hipStream_t st_fast = nullptr, st_bulk = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join_f = nullptr, ev_join_b = nullptr;
    int lane_cfg[4] = {0, 0, 0, 0};         // [0] 1 = streams exist; what they were created for: [1] reserved CUs [2] fast lane on all CUs
    double lane_stats[8] = {0};    // last hooked solve: [0] bursts [1] fast super-steps [2] bulk super-steps [3] sum of the fast lane's instance counts at its round ends [6] fast-lane rounds
    int last_steps = 0;
    PipeArgsH last_args;           // arguments of the most recent pipeline solve (its final iterate stays in the workspace)
    bool last_valid = false;
    double *d_lam_g = nullptr, *d_lam_x = nullptr;   // staging of the multipliers for the host-pointer entry
    int cap_lam = 0;
    // asynchronous solves: one in flight per handle, driven by a worker thread on the handle's stream
    std::thread worker;
    int worker_rc = 0;
    std::atomic<int> n_active{0};  // unfinished instances of the solve in flight (updated at every readback)
    // staging for the host-pointer entry
    double *d_x0 = nullptr, *d_lbx = nullptr, *d_ubx = nullptr, *d_p = nullptr, *d_x = nullptr, *d_g = nullptr,
           *d_f = nullptr, *d_viol = nullptr;
    int *d_iters = nullptr, *d_status = nullptr;
    int cap = 0;
    bool cap_g = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_wait = nullptr;
    // bmpc_debug_time_ric: HIP events around every launch of the Riccati kernel (the dominant kernel: bench.py's roofline leg)
    bool time_ric = false;
    hipEvent_t ric_ev[16] = {nullptr};      // 8 pairs: a burst has at most 8 super-steps and ends with a wait for the stream
    int ric_pending = 0, ric_is_lat[8] = {0}, ric_nact[8] = {0};
    int ric_full_n = 0;                     // grid size that counts as "the whole batch" (the first burst of a solve)
    double ric_full[3] = {0, 0, 0};         // launches of bmpc_k_ric over the whole batch: summed duration [ms], launches, instance-iterations
    double ric_ms[2] = {0, 0};              // [0] bmpc_k_ric, [1] bmpc_k_ric_lat: summed launch durations of the last solve
    long ric_launches[2] = {0, 0}, ric_sweeps[2] = {0, 0};
    bool wedged = false;           // a wait ran into the watchdog: the stream may still be busy, the handle refuses further work
    float last_ms = 0.f;
    std::atomic<bool> busy{false}; // a solve is running on this handle (a handle serves one host thread at a time)
    std::atomic<int> n_loops{0};   // device loops borrowing this handle (bmpc_loop_create / bmpc_loop_destroy)
    bool destroy_pending = false;  // bmpc_destroy called while loops were alive: the last loop frees the handle
    std::string err;
};

#define HIPCHK(h, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

static int pipe_ensure(bmpc_handle* h, int B);

// Wait for everything enqueued on `st` so far.  With bmpc_opts.watchdog_ms > 0 the wait polls an event and gives up after that
// long: a kernel that does not return then costs the caller an error code -- rc 5, bmpc_last_error() --
// instead of a host thread stuck in hipStreamSynchronize for ever.  The handle is unusable afterwards (its stream may never
// drain): destroy it, or end the process when it does not come back.
static int wait_stream(bmpc_handle* h, hipStream_t st) {
    if (h->o.watchdog_ms <= 0) { HIPCHK(h, hipStreamSynchronize(st)); return 0; }
    HIPCHK(h, hipEventRecord(h->ev_wait, st));
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; spins++) {
        const hipError_t q = hipEventQuery(h->ev_wait);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { h->err = std::string("hipEventQuery: ") + hipGetErrorString(q); return 2; }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > h->o.watchdog_ms) {
            h->wedged = true;
            h->err = "watchdog: the GPU did not finish the enqueued work within " + std::to_string(h->o.watchdog_ms) +
                     " ms (bmpc_opts.watchdog_ms); the handle is unusable, destroy it";
            return 5;
        }
        if (ms < 2.0) std::this_thread::yield();            // a burst of super-steps takes a few ms: stay responsive
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
#define WEDGED_FAIL(h) if ((h)->wedged) { (h)->err = "the handle ran into its watchdog earlier and is unusable"; return 5; }

// one solve at a time per handle: a second host thread entering gets an error instead of a corrupted workspace
struct BusyGuard {
    bmpc_handle* h; bool ok;
    explicit BusyGuard(bmpc_handle* h_) : h(h_) { bool f = false; ok = h->busy.compare_exchange_strong(f, true); }
    ~BusyGuard() { if (ok) h->busy.store(false); }
};
#define BUSY_OR_FAIL(h, what)                                                                                   \
    BusyGuard busy_guard_(h);                                                                                   \
    if (!busy_guard_.ok) return 4;       /* (h->err belongs to the thread that owns the handle: not touched) */

extern "C" void bmpc_default_opts(bmpc_opts* o, int N) {
    o->N = N; o->nr_segs = 4; o->dt = 0.1; o->tol = 1e-5; o->max_iter = 100; o->device = 0;
    o->hess = 2; o->hess_switch = 1.0; o->mu_init = 0.1; o->kappa_mu = 0.1; o->theta_mu = 2.0; o->kappa_eps = 1000.0;
    o->mu_floor_k = 1e4; o->inertia = 2; o->dw0 = 1e-4; o->inertia_err = 1e-2; o->stall_n = 8; o->gn_backoff = 2; o->slack_reset = 1; o->ls_alpha_mem = 0.0; o->trial_repeats = 9;
    o->max_batch = 0; o->pool_slots = 0; o->watchdog_ms = 30000;
}

extern "C" int bmpc_create(const bmpc_opts* o, bmpc_handle** out) {
    if (!o || !out) return 1;
    *out = nullptr;
    if (o->N < 3 || o->N > 64 || o->nr_segs != 4 || !(o->dt > 0)) return 1;
    if (o->inertia < 0 || o->inertia > 2 || !(o->dw0 > 0) || o->mu_floor_k < 0) return 1;
    if (o->pool_slots < 0 || (o->pool_slots > 0 && o->pool_slots < 64)) return 1;
    bmpc_handle* h = new bmpc_handle();
    h->o = *o;
    h->o.trial_repeats = o->trial_repeats < 0 ? 0 : (o->trial_repeats > 9 ? 9 : o->trial_repeats);     // a line search has at most ten trials
    h->n_w = 44 * o->N + 6;
    h->n_g = 147 * (o->N - 1) + 21;
    *out = h;   // returned even on a HIP failure so that bmpc_last_error() can be read
    int ndev = 0;
    HIPCHK(h, hipGetDeviceCount(&ndev));
    if (ndev <= 0) { h->err = "no HIP device"; return 2; }
    HIPCHK(h, hipSetDevice(o->device));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, o->device));
    h->n_cu = prop.multiProcessorCount;
    h->nblocks_max = h->n_cu * 3;      // rows of the diagnostic cycle counters
    RobotConst rc;
    robot_iiwa14(h->robot);
    fill_robot_const(rc, h->robot);
    HIPCHK(h, hipMalloc((void**)&h->d_rc, sizeof(RobotConst)));
    HIPCHK(h, hipMemcpy(h->d_rc, &rc, sizeof(RobotConst), hipMemcpyHostToDevice));
    HIPCHK(h, hipMalloc((void**)&h->d_prof, (size_t)h->nblocks_max * 16 * sizeof(double)));
    HIPCHK(h, hipMemset(h->d_prof, 0, (size_t)h->nblocks_max * 16 * sizeof(double)));
    {
        std::vector<int> tbl(3 * HREC);
        bmpc_pipe_build_table(tbl.data());
        HIPCHK(h, hipMalloc((void**)&h->d_pipe_tbl, tbl.size() * sizeof(int)));
        HIPCHK(h, hipMemcpy(h->d_pipe_tbl, tbl.data(), tbl.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipHostMalloc((void**)&h->h_cnt, 2 * NCNT * sizeof(int)));
    }
    HIPCHK(h, hipStreamCreate(&h->stream));
    HIPCHK(h, hipEventCreate(&h->ev0));
    HIPCHK(h, hipEventCreate(&h->ev1));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_wait, hipEventDisableTiming));
    if (o->max_batch > 0) return pipe_ensure(h, o->max_batch);   // workspace up front
    return 0;
}

// device loops (bmpc_loop.hip) register with the handle they borrow, so that destroying the handle first is safe
extern "C" void bmpc_handle_retain(bmpc_handle* h) { if (h) h->n_loops.fetch_add(1); }
extern "C" void bmpc_handle_release(bmpc_handle* h) {
    if (!h) return;
    if (h->n_loops.fetch_sub(1) == 1 && h->destroy_pending) { h->destroy_pending = false; bmpc_destroy(h); }
}

extern "C" void bmpc_destroy(bmpc_handle* h) {
    if (!h) return;
    if (h->n_loops.load() > 0) { h->destroy_pending = true; return; }     // deferred until the last loop is gone
    if (h->worker.joinable()) h->worker.join();
    if (h->wedged) {
        // The handle ran into its watchdog: its stream may never drain, and kernels still queued on it may write the workspace.
        // Nothing is waited for and nothing is freed -- device buffers, events and the stream are leaked on purpose (a
        // hipStreamSynchronize / hipFree here is the unbounded wait the watchdog exists to prevent).  The process should end
        // with an error and let a fresh one take over (include/boundmpc.h, bmpc_opts.watchdog_ms).
        delete h;
        return;
    }
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t e : h->ric_ev) if (e) (void)hipEventDestroy(e);
    double* bufs[] = {h->d_x0, h->d_lbx, h->d_ubx, h->d_p, h->d_x, h->d_g, h->d_f, h->d_viol};
    for (double* b : bufs) if (b) (void)hipFree(b);
    if (h->d_iters) (void)hipFree(h->d_iters);
    if (h->d_status) (void)hipFree(h->d_status);
    if (h->d_rc) (void)hipFree(h->d_rc);
    if (h->d_prof) (void)hipFree(h->d_prof);
    if (h->d_pipe) (void)hipFree(h->d_pipe);
    if (h->d_pipe_st) (void)hipFree(h->d_pipe_st);
    if (h->d_pipe_lists) (void)hipFree(h->d_pipe_lists);
    if (h->d_pipe_tbl) (void)hipFree(h->d_pipe_tbl);
    if (h->d_lam_g) (void)hipFree(h->d_lam_g);
    if (h->d_lam_x) (void)hipFree(h->d_lam_x);
    if (h->h_cnt) (void)hipHostFree(h->h_cnt);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_wait) (void)hipEventDestroy(h->ev_wait);
    for (hipEvent_t e : {h->ev_fork, h->ev_join_f, h->ev_join_b}) if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : {h->st_fast, h->st_bulk}) if (s) (void)hipStreamDestroy(s);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" const char* bmpc_last_error(const bmpc_handle* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int bmpc_dims(const bmpc_handle* h, int* n_w, int* n_g, int* n_p) {
    if (!h) return 1;
    if (n_w) *n_w = h->n_w;
    if (n_g) *n_g = h->n_g;
    if (n_p) *n_p = NPAR;
    return 0;
}

extern "C" void bmpc_robot_iiwa14(bmpc_robot* r) { if (r) robot_iiwa14(*r); }
extern "C" void bmpc_robot_gen3(bmpc_robot* r) { if (r) robot_gen3(*r); }

extern "C" int bmpc_set_robot(bmpc_handle* h, const bmpc_robot* r) {
    if (!h || !r) return 1;
    WEDGED_FAIL(h);
    int rc_ = bmpc_wait(h);
    if (rc_) return rc_;
    BUSY_OR_FAIL(h, "bmpc_set_robot");
    for (int i = 0; i < 7; i++)
        if (!(r->q_lower[i] <= r->q_upper[i]) || !(r->dq_max[i] > 0) || !(r->col_joint_sizes[i] >= 0)) { h->err = "bmpc_set_robot: inconsistent limits"; return 1; }
    if (!(r->ddq_max > 0) || !(r->u_max > 0)) { h->err = "bmpc_set_robot: inconsistent limits"; return 1; }
    HIPCHK(h, hipSetDevice(h->o.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    RobotConst rc;
    fill_robot_const(rc, *r);
    HIPCHK(h, hipMemcpy(h->d_rc, &rc, sizeof(RobotConst), hipMemcpyHostToDevice));
    h->robot = *r;
    h->last_valid = false;
    return 0;
}
extern "C" int bmpc_get_robot(const bmpc_handle* h, bmpc_robot* r) {
    if (!h || !r) return 1;
    *r = h->robot;
    return 0;
}

extern "C" void* bmpc_stream(bmpc_handle* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int bmpc_get_opts(const bmpc_handle* h, bmpc_opts* o) {
    if (!h || !o) return 1;
    *o = h->o;
    return 0;
}

extern "C" int bmpc_gbounds(const bmpc_handle* h, double* lbg, double* ubg) {
    if (!h || !lbg || !ubg) return 1;
    const double INF = 1e20;
    const int N = h->o.N;
    int r = 0;
    auto put = [&](int n, double lo, double hi) { for (int i = 0; i < n; i++) { lbg[r] = lo; ubg[r] = hi; r++; } };
    put(35 * (N - 1), 0, 0);
    for (int k = 1; k < N; k++) {
        put(15, -INF, 0); put(3, -INF, 0); put(3, 0, INF); put(90, -INF, 0); put(1, -INF, 0);
        if (k == N - 1) { put(15, -INF, 0); put(3, -INF, 0); put(3, 0, INF); }
    }
    return r == h->n_g ? 0 : 1;
}

// ------------------------------------------------------------------------------------------
// workspace + launch sequence
// ------------------------------------------------------------------------------------------
// workspace for `B` slots; a handle created with pool_slots > 0 never holds more than that many (larger batches stream
// through the pool, bmpc_opts.pool_slots)
static int pipe_ensure(bmpc_handle* h, int B) {
    if (h->o.pool_slots > 0 && B > h->o.pool_slots) B = h->o.pool_slots;
    if (B <= h->pipe_cap) return 0;
    int cap = B > h->o.max_batch ? B : h->o.max_batch;
    if (h->o.pool_slots > 0 && cap > h->o.pool_slots) cap = h->o.pool_slots;
    if (h->d_pipe) { (void)hipFree(h->d_pipe); h->d_pipe = nullptr; }
    if (h->d_pipe_st) { (void)hipFree(h->d_pipe_st); h->d_pipe_st = nullptr; }
    if (h->d_pipe_lists) { (void)hipFree(h->d_pipe_lists); h->d_pipe_lists = nullptr; }
    h->pipe_cap = 0;
    const size_t n = pipe_workspace_doubles(cap, h->o.N, h->slot_major);
    HIPCHK(h, hipMalloc((void**)&h->d_pipe, n * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->d_pipe_st, (size_t)cap * bmpc_pipe_state_bytes()));
    HIPCHK(h, hipMalloc((void**)&h->d_pipe_lists, 2 * (9 * (size_t)cap + NCNT) * sizeof(int)));
    h->pipe_cap = cap;
    return 0;
}

// one super-step; with bmpc_debug_time_ric the Riccati launch is bracketed by an event pair (collected by ric_collect after the
// next wait for the stream)
static hipError_t step_timed(bmpc_handle* h, PipeArgsH* A, int n_act, hipStream_t st) {
    if (!h->time_ric || h->ric_pending >= 8) return bmpc_pipe_launch_step(A, n_act, st, nullptr, nullptr, nullptr);
    const int i = h->ric_pending++;
    h->ric_nact[i] = n_act;
    return bmpc_pipe_launch_step(A, n_act, st, h->ric_ev[2 * i], h->ric_ev[2 * i + 1], &h->ric_is_lat[i]);
}
static void ric_collect(bmpc_handle* h) {        // the stream is idle: every recorded pair has completed
    for (int i = 0; i < h->ric_pending; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ric_ev[2 * i], h->ric_ev[2 * i + 1]) == hipSuccess) {
            h->ric_ms[h->ric_is_lat[i] ? 1 : 0] += ms; h->ric_launches[h->ric_is_lat[i] ? 1 : 0] += 1;
            if (!h->ric_is_lat[i] && h->ric_nact[i] >= h->ric_full_n) { h->ric_full[0] += ms; h->ric_full[1] += 1; h->ric_full[2] += h->ric_nact[i]; }
        }
    }
    h->ric_pending = 0;
}

// Two-lane closed loop (pipe_solve with a retire hook and priority flags): streams of the lanes.  reserved_cus > 0: the bulk lane's
// stream is masked off that many CUs and the fast lane's stream runs on them alone (fast_all: on every CU) -- the fast lane's
// small grids never queue behind the bulk kernels' workgroups; reserved_cus == 0: the bulk lane stays on the caller's stream, the
// fast lane gets a stream of the highest priority.
static int lanes_ensure(bmpc_handle* h, int reserved_cus, int fast_all) {
    if (h->lane_cfg[0] && h->lane_cfg[1] == reserved_cus && h->lane_cfg[2] == fast_all) return 0;
    for (hipStream_t* s : {&h->st_fast, &h->st_bulk}) if (*s) { (void)hipStreamSynchronize(*s); (void)hipStreamDestroy(*s); *s = nullptr; }
    if (!h->ev_fork) {
        HIPCHK(h, hipEventCreate(&h->ev_fork));
        HIPCHK(h, hipEventCreate(&h->ev_join_f));
        HIPCHK(h, hipEventCreate(&h->ev_join_b));
    }
    if (reserved_cus > 0 && reserved_cus < h->n_cu) {
        const int words = (h->n_cu + 31) / 32;
        std::vector<uint32_t> fast(words, 0u), bulk(words, 0u);
        for (int c = 0; c < h->n_cu; c++) {
            if (c < reserved_cus) fast[c / 32] |= 1u << (c % 32); else bulk[c / 32] |= 1u << (c % 32);
            if (fast_all) fast[c / 32] |= 1u << (c % 32);
        }
        HIPCHK(h, hipExtStreamCreateWithCUMask(&h->st_fast, (uint32_t)words, fast.data()));
        HIPCHK(h, hipExtStreamCreateWithCUMask(&h->st_bulk, (uint32_t)words, bulk.data()));
    } else {
        int lo = 0, hi = 0;
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(h, hipStreamCreateWithPriority(&h->st_fast, hipStreamNonBlocking, hi));
    }
    h->lane_cfg[0] = 1; h->lane_cfg[1] = reserved_cus; h->lane_cfg[2] = fast_all;
    return 0;
}
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

static int pipe_solve(bmpc_handle* h, int B, const double* d_x0, const double* d_lbx, const double* d_ubx,
                      const double* d_p, double* d_x, double* d_g, double* d_f, int* d_iters, int* d_status,
                      double* d_viol, hipStream_t st, bmpc_retire_hook hook = nullptr, void* hook_ctx = nullptr,
                      const int* d_cont = nullptr, const int* d_prio = nullptr, int prio_max = 0) {
    WEDGED_FAIL(h);
    int rc = pipe_ensure(h, B);
    if (rc) return rc;
    const int N = h->o.N, cap = h->pipe_cap;
    PipeArgsH A;
    A.B = B; A.N = N; A.natt = 0; A.pad0_ = 0;
    A.o = SolverOpts{N, h->o.dt, h->o.tol, h->o.max_iter, h->o.hess, h->o.hess_switch,
                     h->o.mu_init, h->o.kappa_mu, h->o.theta_mu, h->o.kappa_eps,
                     h->o.mu_floor_k, h->o.dw0, h->o.inertia_err, h->o.ls_alpha_mem, h->o.inertia, h->o.stall_n, h->o.gn_backoff, h->o.slack_reset, h->o.trial_repeats};
    A.rc = h->d_rc;
    A.x0 = d_x0; A.lbx = d_lbx; A.ubx = d_ubx; A.p = d_p;
    A.x = d_x; A.f = d_f; A.viol = d_viol; A.g = d_g; A.iters = d_iters; A.status = d_status;
    pipe_carve(A, h->d_pipe, cap, N, h->slot_major);
    A.st = (InstState*)h->d_pipe_st;
    int* L = h->d_pipe_lists;
    A.L.eval = L; A.L.step = L + cap; A.L.trial = L + 2 * (size_t)cap; A.L.eval_next = L + 3 * (size_t)cap;
    A.L.trial_next = L + 4 * (size_t)cap; A.L.done = L + 5 * (size_t)cap; A.L.admit = L + 6 * (size_t)cap;
    A.src = L + 7 * (size_t)cap; A.L.curv = L + 8 * (size_t)cap; A.L.cnt = L + 9 * (size_t)cap;
    A.tbl = h->d_pipe_tbl;
    A.prof = h->d_prof;
    A.lam_g = nullptr; A.lam_x = nullptr;
    A.cont = d_cont;
    h->last_valid = false;
    if (hook && B > cap) { h->err = "closed-loop solve: more rollouts than workspace slots"; return 1; }
    auto retire_lane = [&](PipeArgsH& AA, int n_max, int refill, hipStream_t s_) -> int {
        HIPCHK(h, bmpc_pipe_launch_retire_out(&AA, n_max, s_));
        if (hook) { if (int r = hook(hook_ctx, AA.L.done, AA.L.cnt + 8, n_max, (void*)s_)) { h->err = "retire hook failed"; return r; } }
        HIPCHK(h, bmpc_pipe_launch_retire_admit(&AA, n_max, refill, s_));
        return 0;
    };
    auto retire = [&](int n_max, int refill) -> int { return retire_lane(A, n_max, refill, st); };
    // Closed loop without lock step, TWO LANES (round 4).  With one lane every live rollout iterates at the cadence of a full
    // super-step (2.5 ms at 4096 x N=30), and the run lasts as long as the slowest rollout's iterations (13 k of them on
    // configs[4]) times that.  Here the rollouts that lag behind (d_prio, at most prio_max: set by the caller's hook from the
    // steps they have left) iterate in a lane of their own -- lists and counters of its own over the SAME slots and state --
    // on a second stream, several super-steps of a few dozen instances per bulk super-step.  Per-instance arithmetic does not
    // depend on the lane, so the log stays bitwise that of the lock-step loop.
    const bool two = hook && d_prio && prio_max > 0;
    PipeArgsH A1 = A;
    hipStream_t sb = st, sf = st;
    const int lane_burst = env_int("BMPC_FAST_BURST", 4);  // super-steps of the bulk lane per burst
    const int lane_k = env_int("BMPC_FAST_K", 3);          // super-steps of the fast lane per round
    if (two) {
        if (int r = lanes_ensure(h, env_int("BMPC_FAST_CUS", 0), env_int("BMPC_FAST_ALL", 0))) return r;
        sf = h->st_fast; if (h->st_bulk) sb = h->st_bulk;
        int* L1 = L + 9 * (size_t)cap + NCNT;
        A1.L.eval = L1; A1.L.step = L1 + cap; A1.L.trial = L1 + 2 * (size_t)cap; A1.L.eval_next = L1 + 3 * (size_t)cap;
        A1.L.trial_next = L1 + 4 * (size_t)cap; A1.L.done = L1 + 5 * (size_t)cap; A1.L.admit = L1 + 6 * (size_t)cap;
        A1.L.curv = L1 + 8 * (size_t)cap; A1.L.cnt = L1 + 9 * (size_t)cap;
        HIPCHK(h, hipMemsetAsync(A1.L.cnt, 0, NCNT * sizeof(int), st));
    }
    if (hook) for (double& v : h->lane_stats) v = 0;
    // The workspace is a pool of `cap` slots.  B <= cap: every instance has its slot (slot = row).  B > cap (a handle
    // created with pool_slots): the rows stream through the pool -- a slot whose instance has finished is retired
    // (outputs written) and takes the next row at the start of the following super-step, so the kernels keep working on
    // ~cap instances until the input runs out and only ONE straggler tail is paid for the whole call.
    const int n0 = B < cap ? B : cap;
    const bool streaming = B > cap || hook != nullptr;      // slots are refilled (closed loop: with the same row's next problem)
    int cnt0[NCNT] = {0};
    cnt0[0] = n0; cnt0[6] = n0; cnt0[9] = n0;
    HIPCHK(h, hipMemcpyAsync(A.L.cnt, cnt0, sizeof cnt0, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipEventRecord(h->ev0, st));
    HIPCHK(h, bmpc_pipe_launch_init(&A, n0, st));
    // Every instance advances one stage of its own state machine per super-step; finished instances leave the work
    // lists.  The host only needs upper bounds of the list lengths to size the grids, and the retired count to stop:
    // read back every few super-steps.
    int n_act = n0, steps = 0, retired = 0, next_row = n0;
    h->n_active.store(B);
    h->ric_pending = 0; h->ric_ms[0] = h->ric_ms[1] = 0; h->ric_launches[0] = h->ric_launches[1] = 0;
    h->ric_full_n = n0; h->ric_full[0] = h->ric_full[1] = h->ric_full[2] = 0;
    const long max_steps = hook ? (1L << 40) : 12L * (h->o.max_iter + 2) * ((B + cap - 1) / cap + 1);
    while (retired < B && steps < max_steps) {
        const int burst = steps < 8 ? 8 : 4;
        if (hook) {
            // closed loop without lock step: a burst of super-steps, then the counters come back and exactly the rollouts whose
            // solve finished in this burst (cnt[8], the done list) are retired -- outputs, the caller's hook (post-processing, next
            // problem), re-admission -- with grids sized by that count; nothing is launched when nobody finished.  (Round 3 ran
            // the whole retirement sequence before every super-step with grids sized for all rollouts: ~1 ms of empty launches
            // per super-step.)  A finished rollout waits at most one burst for its next problem.
            if (two && steps >= 8) {
                // deal the live instances out between the lanes; the bulk lane then runs its burst while the fast lane goes through
                // rounds of its own -- a few super-steps, counters back, retirement of ITS finished rollouts (hook and re-admission on
                // the fast lane's stream: they touch those rollouts only) -- until the bulk burst has ended
                const int n1 = n_act < prio_max ? n_act : prio_max;
                HIPCHK(h, bmpc_pipe_launch_pick(&A, &A1, d_prio, n_act, st));
                HIPCHK(h, hipEventRecord(h->ev_fork, st));
                HIPCHK(h, hipStreamWaitEvent(sf, h->ev_fork, 0));
                if (sb != st) HIPCHK(h, hipStreamWaitEvent(sb, h->ev_fork, 0));
                for (int i = 0; i < lane_burst; i++, steps++) HIPCHK(h, step_timed(h, &A, n_act, sb));
                HIPCHK(h, hipEventRecord(h->ev_join_b, sb));
                h->lane_stats[0] += 1; h->lane_stats[2] += lane_burst;
                for (int round = 0;; round++) {
                    for (int i = 0; i < lane_k; i++) HIPCHK(h, bmpc_pipe_launch_step(&A1, n1, sf, nullptr, nullptr, nullptr));
                    HIPCHK(h, hipMemcpyAsync(h->h_cnt + NCNT, A1.L.cnt, NCNT * sizeof(int), hipMemcpyDeviceToHost, sf));
                    if (int r = wait_stream(h, sf)) return r;
                    const int live1 = h->h_cnt[NCNT + 0] + h->h_cnt[NCNT + 2], nd1 = h->h_cnt[NCNT + 8];
                    h->lane_stats[1] += lane_k; h->lane_stats[3] += live1 + nd1; h->lane_stats[6] += 1;
                    const hipError_t q = hipEventQuery(h->ev_join_b);
                    if (q != hipErrorNotReady) { HIPCHK(h, q); break; }          // the bulk burst is over: join (what the fast lane finished last is retired below)
                    if (live1 + nd1 == 0) break;                                  // nobody in the fast lane
                    if (nd1 > 0) { if (int r = retire_lane(A1, nd1, 1, sf)) return r; }
                }
                HIPCHK(h, hipEventRecord(h->ev_join_f, sf));
                HIPCHK(h, hipStreamWaitEvent(st, h->ev_join_f, 0));
                if (sb != st) HIPCHK(h, hipStreamWaitEvent(st, h->ev_join_b, 0));
            } else {
                for (int i = 0; i < burst; i++, steps++) HIPCHK(h, step_timed(h, &A, n_act, st));
            }
            auto read_counters = [&]() -> int {
                HIPCHK(h, hipMemcpyAsync(h->h_cnt, A.L.cnt, NCNT * sizeof(int), hipMemcpyDeviceToHost, st));
                if (two) HIPCHK(h, hipMemcpyAsync(h->h_cnt + NCNT, A1.L.cnt, NCNT * sizeof(int), hipMemcpyDeviceToHost, st));
                else for (int i = 0; i < NCNT; i++) h->h_cnt[NCNT + i] = 0;
                return wait_stream(h, st);
            };
            if (int r = read_counters()) return r;
            ric_collect(h);
            const int n_done = h->h_cnt[8], n_done1 = h->h_cnt[NCNT + 8];
            if (n_done > 0) { if (int r = retire(n_done, 1)) return r; }
            if (n_done1 > 0) { if (int r = retire_lane(A1, n_done1, 1, st)) return r; }
            retired = h->h_cnt[7] + h->h_cnt[NCNT + 7];      // rows whose rollout has ended (counted by k_admit: one burst behind)
            if (n_done + n_done1 > 0 && retired + n_done + n_done1 >= B) {      // possibly the last ones: their retirement decides whether anybody goes on
                if (int r = read_counters()) return r;
                retired = h->h_cnt[7] + h->h_cnt[NCNT + 7];
            }
            n_act = B - retired;
            h->n_active.store(B - retired);
            continue;
        }
        // while input rows are left (as far as the host knows: next_row only grows), finished instances make room before
        // every super-step; afterwards they are retired once per burst
        const bool rows_left = streaming && next_row < B;
        for (int i = 0; i < burst; i++, steps++) {
            if (rows_left && i > 0) { if (int r = retire(cap, 1)) return r; }
            HIPCHK(h, step_timed(h, &A, rows_left ? cap : n_act, st));
        }
        if (int r = retire(rows_left ? cap : n_act, rows_left ? 1 : 0)) return r;
        HIPCHK(h, hipMemcpyAsync(h->h_cnt, A.L.cnt, NCNT * sizeof(int), hipMemcpyDeviceToHost, st));
        if (int r = wait_stream(h, st)) return r;
        ric_collect(h);
        retired = h->h_cnt[7];
        next_row = h->h_cnt[6] < B ? h->h_cnt[6] : B;
        n_act = next_row - retired;
        if (streaming && next_row < B) n_act = cap < B ? cap : B;
        h->n_active.store(B - retired);
    }
    h->last_steps = steps;
    h->ric_sweeps[0] = h->h_cnt[11] + (two ? h->h_cnt[NCNT + 11] : 0); h->ric_sweeps[1] = h->h_cnt[12] + (two ? h->h_cnt[NCNT + 12] : 0);
    HIPCHK(h, hipEventRecord(h->ev1, st));
    // the outputs are complete and the per-handle workspace is free when the call returns (the loop above synchronised)
    if (int r = wait_stream(h, st)) return r;
    if (retired < B) { h->err = "pipeline did not drain (internal error)"; return 3; }
    h->last_args = A; h->last_valid = !streaming;      // multipliers need every instance's final iterate in its slot
    h->last_args.cont = nullptr;
    return 0;
}

static int launch(bmpc_handle* h, int B, const double* d_x0, const double* d_lbx, const double* d_ubx,
                  const double* d_p, double* d_x, double* d_g, double* d_f, int* d_iters, int* d_status,
                  double* d_viol, hipStream_t st) {
    return pipe_solve(h, B, d_x0, d_lbx, d_ubx, d_p, d_x, d_g, d_f, d_iters, d_status, d_viol, st);
}

extern "C" int bmpc_solve_dev(bmpc_handle* h, int B, const double* d_x0, const double* d_lbx,
                              const double* d_ubx, const double* d_p, double* d_x, double* d_g, double* d_f,
                              int* d_iters, int* d_status, double* d_viol, void* stream) {
    if (!h || B < 0 || !d_x0 || !d_lbx || !d_ubx || !d_p || !d_x || !d_f || !d_iters || !d_status || !d_viol) {
        if (h) h->err = "bmpc_solve_dev: null argument";
        return 1;
    }
    int wrc = bmpc_wait(h);        // an asynchronous solve in flight owns the workspace
    if (wrc) return wrc;
    if (B == 0) return 0;
    BUSY_OR_FAIL(h, "bmpc_solve_dev");
    HIPCHK(h, hipSetDevice(h->o.device));
    return launch(h, B, d_x0, d_lbx, d_ubx, d_p, d_x, d_g, d_f, d_iters, d_status, d_viol, (hipStream_t)stream);
}

// Closed loop without lock step (bmpc_loop_run_async): B rows, each a rollout whose successive problems are produced in
// place by `hook`; a row is solved again while d_cont[row] != 0.  Internal to the library (bmpc_loop.hip).
extern "C" int bmpc_solve_dev_hooked(bmpc_handle* h, int B, const double* d_x0, const double* d_lbx, const double* d_ubx,
                                     const double* d_p, double* d_x, double* d_f, int* d_iters, int* d_status, double* d_viol,
                                     void* stream, bmpc_retire_hook hook, void* hook_ctx, const int* d_cont,
                                     const int* d_prio, int prio_max) {
    if (!h || B <= 0 || !hook || !d_cont) return 1;
    int wrc = bmpc_wait(h);
    if (wrc) return wrc;
    BUSY_OR_FAIL(h, "bmpc_loop_run_async");
    HIPCHK(h, hipSetDevice(h->o.device));
    return pipe_solve(h, B, d_x0, d_lbx, d_ubx, d_p, d_x, nullptr, d_f, d_iters, d_status, d_viol, (hipStream_t)stream, hook, hook_ctx, d_cont, d_prio, prio_max);
}
// lane statistics of the last two-lane hooked solve (tools/closed_loop_device.py): see bmpc_handle::lane_stats
extern "C" int bmpc_debug_lane_stats(bmpc_handle* h, double* out8) {
    if (!h || !out8) return 1;
    for (int i = 0; i < 8; i++) out8[i] = h->lane_stats[i];
    return 0;
}

// Asynchronous form of bmpc_solve_dev: returns at once; the data-dependent launch sequence is driven by
// a worker thread on the handle's own stream.  One solve in flight per handle (a second call waits for
// the first).  Two handles used alternately overlap the straggler tail of one batch (few active
// instances, launch-latency bound) with the bulk of the next.
// unfinished instances of the solve in flight on this handle (0 when idle): lets a caller that keeps
// several batches in flight start the next one when the previous has left its bulk phase
extern "C" int bmpc_active(bmpc_handle* h) { return h ? h->n_active.load() : 0; }

extern "C" int bmpc_wait(bmpc_handle* h) {
    if (!h) return 1;
    if (h->worker.joinable()) h->worker.join();
    int rc = h->worker_rc;
    h->worker_rc = 0;
    return rc;
}

extern "C" int bmpc_solve_dev_async(bmpc_handle* h, int B, const double* d_x0, const double* d_lbx,
                                    const double* d_ubx, const double* d_p, double* d_x, double* d_g, double* d_f,
                                    int* d_iters, int* d_status, double* d_viol) {
    if (!h || B < 0 || !d_x0 || !d_lbx || !d_ubx || !d_p || !d_x || !d_f || !d_iters || !d_status || !d_viol) {
        if (h) h->err = "bmpc_solve_dev_async: null argument";
        return 1;
    }
    int rc = bmpc_wait(h);
    if (rc) return rc;
    if (B == 0) return 0;
    { bool f = false; if (!h->busy.compare_exchange_strong(f, true)) return 4; }
    h->n_active.store(B);
    h->worker = std::thread([=]() {      // the worker owns the handle until it is done (bmpc_wait joins it)
        int r = 0;
        if (hipSetDevice(h->o.device) != hipSuccess) { h->err = "hipSetDevice failed in the worker"; r = 2; }
        if (r == 0) r = launch(h, B, d_x0, d_lbx, d_ubx, d_p, d_x, d_g, d_f, d_iters, d_status, d_viol, h->stream);
        if (r == 0) r = wait_stream(h, h->stream);
        h->worker_rc = r;
        if (r != 0) h->n_active.store(0);        // a failed solve is not "active" for ever; bmpc_wait reports the code
        h->busy.store(false);
    });
    return 0;
}

// Multipliers of the most recent solve on this handle: its final iterate and row multipliers are still
// in the workspace.  d_lam_g [B][n_g], d_lam_x [B][n_w]: device pointers; enqueued on `stream` and waited for.
extern "C" int bmpc_multipliers_dev(bmpc_handle* h, int B, double* d_lam_g, double* d_lam_x, void* stream) {
    if (!h || !d_lam_g || !d_lam_x) { if (h) h->err = "bmpc_multipliers_dev: null argument"; return 1; }
    WEDGED_FAIL(h);
    int rc = bmpc_wait(h);
    if (rc) return rc;
    if (!h->last_valid || h->last_args.B != B) { h->err = "bmpc_multipliers_dev: no finished solve of this batch size on the handle"; return 1; }
    HIPCHK(h, hipSetDevice(h->o.device));
    PipeArgsH A = h->last_args;
    A.lam_g = d_lam_g; A.lam_x = d_lam_x;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(h, bmpc_pipe_launch_mult(&A, st));
    if (int r = wait_stream(h, st)) return r;
    return 0;
}

static int ensure_cap(bmpc_handle* h, int B, bool want_g) {
    if (B <= h->cap && (!want_g || h->cap_g)) return 0;
    int cap = B > h->cap ? B : h->cap;
    if (h->o.max_batch > cap) cap = h->o.max_batch;
    double** bufs[] = {&h->d_x0, &h->d_lbx, &h->d_ubx, &h->d_x};
    for (double** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; HIPCHK(h, hipMalloc((void**)b, (size_t)cap * h->n_w * sizeof(double))); }
    if (h->d_p) (void)hipFree(h->d_p);
    HIPCHK(h, hipMalloc((void**)&h->d_p, (size_t)cap * NPAR * sizeof(double)));
    if (h->d_f) (void)hipFree(h->d_f);
    HIPCHK(h, hipMalloc((void**)&h->d_f, (size_t)cap * sizeof(double)));
    if (h->d_viol) (void)hipFree(h->d_viol);
    HIPCHK(h, hipMalloc((void**)&h->d_viol, (size_t)cap * sizeof(double)));
    if (h->d_iters) (void)hipFree(h->d_iters);
    HIPCHK(h, hipMalloc((void**)&h->d_iters, (size_t)cap * sizeof(int)));
    if (h->d_status) (void)hipFree(h->d_status);
    HIPCHK(h, hipMalloc((void**)&h->d_status, (size_t)cap * sizeof(int)));
    if (h->d_g) { (void)hipFree(h->d_g); h->d_g = nullptr; }
    h->cap_g = false;
    if (want_g) { HIPCHK(h, hipMalloc((void**)&h->d_g, (size_t)cap * h->n_g * sizeof(double))); h->cap_g = true; }
    h->cap = cap;
    return 0;
}

extern "C" int bmpc_solve(bmpc_handle* h, int B, const double* x0, const double* lbx, const double* ubx,
                          const double* p, double* x, double* g, double* lam_g, double* lam_x, double* f,
                          int* iters, int* status, double* viol) {
    if (!h || B < 0 || !x0 || !lbx || !ubx || !p || !x || !f || !iters || !status || !viol) {
        if (h) h->err = "bmpc_solve: null argument";
        return 1;
    }
    if (B == 0) return 0;
    WEDGED_FAIL(h);                // (before ensure_cap below, whose hipFree would wait for a stream that never drains)
    int rc = bmpc_wait(h);         // an asynchronous solve in flight owns the workspace
    if (rc) return rc;
    BUSY_OR_FAIL(h, "bmpc_solve");
    // multipliers need every instance's final iterate in its own workspace slot: refused up front (not after the solve) when the
    // call would stream through a smaller pool
    if ((lam_g || lam_x) && h->o.pool_slots > 0 && B > h->o.pool_slots) {
        h->err = "bmpc_solve: lam_g / lam_x need B <= pool_slots (a streamed call keeps no final iterates)";
        return 1;
    }
    HIPCHK(h, hipSetDevice(h->o.device));
    rc = ensure_cap(h, B, g != nullptr);
    if (rc) return rc;
    size_t nw = (size_t)B * h->n_w * sizeof(double);
    hipStream_t st = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->d_x0, x0, nw, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->d_lbx, lbx, nw, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->d_ubx, ubx, nw, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->d_p, p, (size_t)B * NPAR * sizeof(double), hipMemcpyHostToDevice, st));
    rc = launch(h, B, h->d_x0, h->d_lbx, h->d_ubx, h->d_p, h->d_x, g ? h->d_g : nullptr, h->d_f, h->d_iters,
                h->d_status, h->d_viol, st);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(x, h->d_x, nw, hipMemcpyDeviceToHost, st));
    if (g) HIPCHK(h, hipMemcpyAsync(g, h->d_g, (size_t)B * h->n_g * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(f, h->d_f, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(viol, h->d_viol, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(iters, h->d_iters, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(status, h->d_status, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    if (int r = wait_stream(h, st)) return r;
    HIPCHK(h, hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    if (lam_g || lam_x) {
        if (B > h->cap_lam) {
            if (h->d_lam_g) { (void)hipFree(h->d_lam_g); h->d_lam_g = nullptr; }
            if (h->d_lam_x) { (void)hipFree(h->d_lam_x); h->d_lam_x = nullptr; }
            h->cap_lam = 0;
            HIPCHK(h, hipMalloc((void**)&h->d_lam_g, (size_t)B * h->n_g * sizeof(double)));
            HIPCHK(h, hipMalloc((void**)&h->d_lam_x, (size_t)B * h->n_w * sizeof(double)));
            h->cap_lam = B;
        }
        rc = bmpc_multipliers_dev(h, B, h->d_lam_g, h->d_lam_x, st);
        if (rc) return rc;
        if (lam_g) HIPCHK(h, hipMemcpy(lam_g, h->d_lam_g, (size_t)B * h->n_g * sizeof(double), hipMemcpyDeviceToHost));
        if (lam_x) HIPCHK(h, hipMemcpy(lam_x, h->d_lam_x, (size_t)B * h->n_w * sizeof(double), hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int bmpc_last_kernel_ms(bmpc_handle* h, float* ms) {
    if (!h || !ms) return 1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, h->ev0, h->ev1) == hipSuccess) h->last_ms = t;
    *ms = h->last_ms;
    return 0;
}

extern "C" int bmpc_fk(bmpc_handle* h, int B, const double* q, const double* dq, double* ee_pos,
                       double* ee_rot, double* col_pts, double* jac, double* dvdq) {
    if (!h || B < 0 || !q) { if (h) h->err = "bmpc_fk: null argument"; return 1; }
    if (B == 0) return 0;
    HIPCHK(h, hipSetDevice(h->o.device));
    double *d_q = nullptr, *d_dq = nullptr, *d_out = nullptr;
    const size_t per = 3 + 9 + 18 + 42 + 42;
    auto body = [&]() -> int {
        HIPCHK(h, hipMalloc((void**)&d_q, (size_t)B * 7 * sizeof(double)));
        HIPCHK(h, hipMalloc((void**)&d_dq, (size_t)B * 7 * sizeof(double)));
        HIPCHK(h, hipMalloc((void**)&d_out, (size_t)B * per * sizeof(double)));
        HIPCHK(h, hipMemcpy(d_q, q, (size_t)B * 7 * sizeof(double), hipMemcpyHostToDevice));
        if (dq) HIPCHK(h, hipMemcpy(d_dq, dq, (size_t)B * 7 * sizeof(double), hipMemcpyHostToDevice));
        else HIPCHK(h, hipMemset(d_dq, 0, (size_t)B * 7 * sizeof(double)));
        double* o_pos = d_out; double* o_rot = o_pos + (size_t)B * 3; double* o_col = o_rot + (size_t)B * 9;
        double* o_jac = o_col + (size_t)B * 18; double* o_dv = o_jac + (size_t)B * 42;
        HIPCHK(h, bmpc_launch_fk(B, h->d_rc, d_q, d_dq, o_pos, o_rot, o_col, o_jac, o_dv, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (ee_pos) HIPCHK(h, hipMemcpy(ee_pos, o_pos, (size_t)B * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (ee_rot) HIPCHK(h, hipMemcpy(ee_rot, o_rot, (size_t)B * 9 * sizeof(double), hipMemcpyDeviceToHost));
        if (col_pts) HIPCHK(h, hipMemcpy(col_pts, o_col, (size_t)B * 18 * sizeof(double), hipMemcpyDeviceToHost));
        if (jac) HIPCHK(h, hipMemcpy(jac, o_jac, (size_t)B * 42 * sizeof(double), hipMemcpyDeviceToHost));
        if (dvdq) HIPCHK(h, hipMemcpy(dvdq, o_dv, (size_t)B * 42 * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    };
    const int rc = body();          // temporaries are freed on the error paths too
    if (d_q) (void)hipFree(d_q);
    if (d_dq) (void)hipFree(d_dq);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

// diagnostic: per-phase cycle sums accumulated by a -DBMPC_PROFILE build (zeros otherwise)
extern "C" int bmpc_debug_phase_cycles(bmpc_handle* h, double* out16) {
    if (!h || !out16) return 1;
    std::vector<double> buf((size_t)h->nblocks_max * 16);
    HIPCHK(h, hipMemcpy(buf.data(), h->d_prof, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++) out16[i] = 0;
    for (int b = 0; b < h->nblocks_max; b++) for (int i = 0; i < 16; i++) out16[i] += buf[(size_t)b * 16 + i];
    HIPCHK(h, hipMemset(h->d_prof, 0, buf.size() * sizeof(double)));
    return 0;
}

// diagnostic: per-instance solver state the most recent finished solve left in the workspace, row by row (out [B][12], host):
// {iterations, status, mu, alpha (1e300: the line search found no acceptable step), alpha_dual, fraction-to-boundary alpha, delta_w,
//  exact Hessian wanted next, factorisation retries, rejected trials, KKT error of the previous iterate, stall counter} -- with
// max_iter = k these are the decisions of iteration k - 1 (tests/test_iterate_parity.py compares them with the oracle's).  Needs
// a solve that kept every instance in a slot of its own (B <= slots).
extern "C" int bmpc_debug_inst_state(bmpc_handle* h, int B, double* out) {
    if (!h || !out) { if (h) h->err = "bmpc_debug_inst_state: null argument"; return 1; }
    int rc = bmpc_wait(h);
    if (rc) return rc;
    WEDGED_FAIL(h);
    if (!h->last_valid || h->last_args.B != B || B > h->pipe_cap || (h->o.pool_slots > 0 && B > h->o.pool_slots)) {
        h->err = "bmpc_debug_inst_state: no finished solve of this batch size with a slot per instance on the handle"; return 1;
    }
    HIPCHK(h, hipSetDevice(h->o.device));
    std::vector<InstState> st((size_t)B);
    std::vector<int> src((size_t)B);
    HIPCHK(h, hipMemcpy(st.data(), h->d_pipe_st, st.size() * sizeof(InstState), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(src.data(), h->last_args.src, src.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int s = 0; s < B; s++) {
        const InstState& t = st[(size_t)s];
        const int row = src[(size_t)s];
        if (row < 0 || row >= B) { h->err = "bmpc_debug_inst_state: slot map out of range"; return 1; }
        double* o = out + (size_t)row * 12;
        o[0] = t.it; o[1] = t.status; o[2] = t.mu; o[3] = t.alpha; o[4] = t.ad; o[5] = t.ap; o[6] = t.hreg; o[7] = t.hess_mode;
        o[8] = t.tries; o[9] = t.bt; o[10] = t.err_prev; o[11] = t.stall;
    }
    return 0;
}

// diagnostic / measurement: HIP events around every launch of the Riccati kernel (bmpc_k_ric: the throughput variant, bmpc_k_ric_lat:
// the latency variant of nearly empty super-steps) on the handle's stream, from the next solve on.  bmpc_debug_ric_stats returns, for
// the most recent solve, out[0..2] = {summed launch durations in ms, launches, instance-iterations (workgroups that ran)} of
// bmpc_k_ric and out[3..5] the same for bmpc_k_ric_lat.  bench.py's roofline leg: algorithmic flops of the launches / their duration.
extern "C" int bmpc_debug_time_ric(bmpc_handle* h, int on) {
    if (!h) return 1;
    int rc = bmpc_wait(h);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->o.device));
    if (on && !h->ric_ev[0]) for (auto& e : h->ric_ev) HIPCHK(h, hipEventCreate(&e));
    h->time_ric = on != 0;
    return 0;
}
extern "C" int bmpc_debug_ric_stats(bmpc_handle* h, double* out6) {
    if (!h || !out6) return 1;
    int rc = bmpc_wait(h);
    if (rc) return rc;
    for (int v = 0; v < 2; v++) { out6[3 * v] = h->ric_ms[v]; out6[3 * v + 1] = (double)h->ric_launches[v]; out6[3 * v + 2] = (double)h->ric_sweeps[v]; }
    return 0;
}

// diagnostic: keep the handle's stream busy for `ms` milliseconds (at most 10 s) -- lets a test exercise the watchdog
extern "C" int bmpc_debug_ric_stats_full(bmpc_handle* h, double* out3) {
    if (!h || !out3) return 1;
    for (int i = 0; i < 3; i++) out3[i] = h->ric_full[i];
    return 0;
}

extern "C" int bmpc_debug_spin(bmpc_handle* h, int ms) {
    if (!h) return 1;
    WEDGED_FAIL(h);
    HIPCHK(h, hipSetDevice(h->o.device));
    HIPCHK(h, bmpc_launch_spin(ms, h->stream));
    return 0;
}
