// Device-resident closed loop (include/boundmpc.h, bmpc_loop_*): kernels around the batched solve for
// gfx950.  Per-rollout logic: bmpc_loop.hpp (one thread per rollout -- it is sequential, data-dependent
// bookkeeping on 3-vectors, a few microseconds per step); bulk row traffic (start vector, bound rows,
// warm-start copy) goes through element-wise kernels so that every HBM access is coalesced.
#include "bmpc_platform_hip.hpp"

#define BMPC_NT 64
#include "bmpc_loop.hpp"
#include "bmpc_robot.hpp"

#include <chrono>
#include <string>
#include <vector>

#include "../../include/boundmpc.h"

using namespace bmpc;

extern "C" void bmpc_handle_retain(bmpc_handle* h);
extern "C" void bmpc_handle_release(bmpc_handle* h);

__global__ __launch_bounds__(256) void bmpc_loop_k_bounds(int R, int N, const RobotConst* rc, double* lbx, double* ubx) {
    const int n_w = 44 * N + 6;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)R * n_w) return;
    loop_bound_const(rc, N, (int)(e % n_w), lbx + e, ubx + e);
}

// closest pairs collision-point segment <-> obstacle: one thread per (rollout, collision point), blockIdx.y = obstacle, so
// that the obstacle's rows are wave-uniform (scalar loads)
// Every kernel below runs either over all R rollouts (list == nullptr) or over the rollouts of a device list of length
// *n_list (bmpc_loop_run_async: the rollouts whose solve has just retired).
#define LOOP_ROLLOUT(e, r)                                  \
    int r = (e);                                            \
    if (list) { if ((e) >= *n_list) return; r = list[e]; }  \
    else if ((e) >= R) return;

__global__ __launch_bounds__(64) void bmpc_loop_k_colpairs(int R, const RobotConst* rc, LoopScene sc, const double* S, double* colres,
                                                           const int* list, const int* n_list) {
    const int e6 = blockIdx.x * 64 + threadIdx.x, ob = blockIdx.y;
    const int e = e6 / 6, pt = e6 - 6 * e;
    LOOP_ROLLOUT(e, r)
    const double* s = S + (size_t)r * LS_SIZE;
    if (s[LS_dead] != 0.0) return;
    loop_collision_pair(rc, sc, s, pt, ob, colres + (((size_t)r * 6 + pt) * sc.n_obs + ob) * LP_CRES);
}

__global__ __launch_bounds__(64) void bmpc_loop_k_prepare(int R, int N, const RobotConst* rc, double* S, const double* prev,
                                                          double* p, double* lbx, double* ubx, LoopScene sc, const double* colres,
                                                          const int* list, const int* n_list) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    LOOP_ROLLOUT(e, r)
    const size_t n_w = 44 * N + 6;
    double* s = S + (size_t)r * LS_SIZE;
    if (s[LS_dead] != 0.0) return;
    loop_prepare(rc, N, s, prev + r * n_w, p + (size_t)r * NPAR, lbx + r * n_w, ubx + r * n_w, &sc,
                 colres ? colres + (size_t)r * 6 * sc.n_obs * LP_CRES : nullptr);
}

__global__ __launch_bounds__(256) void bmpc_loop_k_x0(int R, int N, const double* S, const double* prev, double* x0,
                                                      const int* list, const int* n_list) {
    const int n_w = 44 * N + 6;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int e = (int)(idx / n_w), i = (int)(idx - (size_t)e * n_w);
    LOOP_ROLLOUT(e, r)
    x0[(size_t)r * n_w + i] = loop_x0_elem(N, S + (size_t)r * LS_SIZE, prev + (size_t)r * n_w, i);
}

// steps_left (bmpc_loop_run_async): per-rollout countdown; the log row of a rollout's step is (steps taken so far) * R + r,
// cont[r] tells the solver whether the rollout has another problem coming
__global__ __launch_bounds__(64) void bmpc_loop_k_finish(int R, int N, double dt, const RobotConst* rc, double* S, const double* x,
                                                         const double* prev, const int* status, const double* viol,
                                                         const int* iters, double* log, const int* list, const int* n_list,
                                                         int* steps_left, int* cont, int nsteps, const double* par, double* rec,
                                                         const int* rec_slot) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    LOOP_ROLLOUT(e, r)
    const size_t n_w = 44 * N + 6;
    double* s = S + (size_t)r * LS_SIZE;
    double* lg = log ? log + (size_t)r * LP_LOGW : nullptr;
    if (steps_left) {
        const int left = steps_left[r];
        if (log) lg = log + ((size_t)(nsteps - left) * R + r) * LP_LOGW;
        steps_left[r] = left - 1;
        cont[r] = left - 1 > 0;
    }
    if (s[LS_dead] != 0.0) {
        s[LS_accept] = 0.0;
        if (lg) { for (int i = 0; i < LP_LOGW; i++) lg[i] = 0.0; lg[4] = 1.0; }
        return;
    }
    double* rc_ = (rec && rec_slot[r] >= 0) ? rec + (size_t)rec_slot[r] * lp_rec_doubles(N) : nullptr;
    loop_finish(rc, N, dt, s, x + r * n_w, prev + r * n_w, status[r], viol[r], iters[r], lg, rc_, par + (size_t)r * NPAR);
}

// prev_solution <- accepted solution (BoundMPC.py:643)
__global__ __launch_bounds__(256) void bmpc_loop_k_keep(int R, int N, const double* S, const double* x, double* prev,
                                                        const int* list, const int* n_list) {
    const int n_w = 44 * N + 6;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int e = (int)(idx / n_w), i = (int)(idx - (size_t)e * n_w);
    LOOP_ROLLOUT(e, r)
    const size_t o = (size_t)r * n_w + i;
    if (S[(size_t)r * LS_SIZE + LS_accept] != 0.0) prev[o] = x[o];
}

// bmpc_loop_run_async, two lanes: priority flags of the rollouts for the solver (bmpc_capi.hip, pipe_solve).  The rollouts that
// lag behind -- most steps left -- are flagged, as many whole "steps left" classes from the top as fit prio_max; nobody while the
// class of the laggards alone is larger (at the start every rollout has all its steps left).  One workgroup.
__global__ __launch_bounds__(1024) void bmpc_loop_k_prio(int R, const int* steps_left, int* prio, int prio_max) {
    __shared__ int hist[1024];
    __shared__ int s_max, s_thr;
    const int tid = threadIdx.x;
    hist[tid] = 0;
    if (tid == 0) s_max = 0;
    __syncthreads();
    int m = 0;
    for (int r = tid; r < R; r += 1024) m = max(m, steps_left[r]);
    atomicMax(&s_max, m);
    __syncthreads();
    const int base = s_max - 1023;                 // classes below `base` share bin 0 (never reached before the budget is spent unless all fit)
    for (int r = tid; r < R; r += 1024) { const int v = steps_left[r]; if (v > 0) atomicAdd(&hist[max(v - base, 0)], 1); }
    __syncthreads();
    if (tid == 0) {
        int cum = 0, thr = 1024;
        for (int b = 1023; b >= 0; b--) { if (cum + hist[b] > prio_max) break; cum += hist[b]; thr = b; }
        s_thr = thr;
    }
    __syncthreads();
    const int thr = s_thr;
    for (int r = tid; r < R; r += 1024) { const int v = steps_left[r]; prio[r] = (v > 0 && max(v - base, 0) >= thr) ? 1 : 0; }
}

struct bmpc_loop {
    bmpc_handle* h = nullptr;
    int R = 0, N = 0, n_w = 0, dev = 0;
    double dt = 0.1;
    RobotConst* d_rc = nullptr;
    double *d_S = nullptr, *d_prev = nullptr, *d_x0 = nullptr, *d_lbx = nullptr, *d_ubx = nullptr, *d_p = nullptr,
           *d_x = nullptr, *d_f = nullptr, *d_viol = nullptr, *d_log = nullptr;
    int *d_iters = nullptr, *d_status = nullptr;
    LoopScene sc{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // device pointers
    double* d_scene = nullptr;     // A | b | AAt | V
    int* d_scene_i = nullptr;      // nrows | nv
    double* d_colres = nullptr;
    int *d_steps_left = nullptr, *d_cont = nullptr;    // bmpc_loop_run_async
    int* d_prio = nullptr;         // ... its fast lane: [R] 1 = the rollout lags behind (bmpc_loop_k_prio)
    int prio_max = 0;
    int* d_rec_slot = nullptr;     // [R] index of the rollout's record in a step's block, or -1 (bmpc_loop_set_record)
    int n_rec = 0;
    double* d_rec = nullptr;       // [steps][n_rec][lp_rec_doubles(N)] of the last bmpc_loop_run
    size_t rec_cap = 0, rec_steps = 0;
    int async_nsteps = 0;
    double* async_log = nullptr;
    size_t log_cap = 0;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::string err;
};

#define LCHK(L, call)                                                            \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            (L)->err = std::string(#call) + ": " + hipGetErrorString(e_);        \
            return 2;                                                            \
        }                                                                        \
    } while (0)

extern "C" int bmpc_loop_state_doubles(void) { return LS_SIZE; }
extern "C" int bmpc_loop_log_doubles(void) { return LP_LOGW; }
extern "C" int bmpc_loop_record_doubles(int N) { return lp_rec_doubles(N); }

// MPCData records (include/boundmpc.h): the rollouts whose steps bmpc_loop_run records from now on (n = 0: none)
extern "C" int bmpc_loop_set_record(bmpc_loop* L, int n, const int* rollouts) {
    if (!L || n < 0 || (n > 0 && !rollouts)) return 1;
    LCHK(L, hipSetDevice(L->dev));
    std::vector<int> slot((size_t)L->R, -1);
    for (int j = 0; j < n; j++) {
        if (rollouts[j] < 0 || rollouts[j] >= L->R) { L->err = "bmpc_loop_set_record: rollout index out of range"; return 1; }
        if (slot[(size_t)rollouts[j]] >= 0) { L->err = "bmpc_loop_set_record: a rollout is listed twice"; return 1; }
        slot[(size_t)rollouts[j]] = j;
    }
    if (!L->d_rec_slot) LCHK(L, hipMalloc((void**)&L->d_rec_slot, (size_t)L->R * sizeof(int)));
    LCHK(L, hipStreamSynchronize(L->st));
    LCHK(L, hipMemcpy(L->d_rec_slot, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice));
    L->n_rec = n; L->rec_steps = 0;
    return 0;
}
// the records of the last bmpc_loop_run: out [steps][n][bmpc_loop_record_doubles(N)], room for max_steps steps (out == NULL: only
// *steps is returned, to size the buffer); more recorded steps than room: rc 1, nothing copied
extern "C" int bmpc_loop_records(bmpc_loop* L, double* out, int max_steps, int* steps) {
    if (!L || (!out && !steps)) return 1;
    LCHK(L, hipSetDevice(L->dev));
    LCHK(L, hipStreamSynchronize(L->st));
    if (steps) *steps = (int)L->rec_steps;
    if (!out) return 0;
    if (L->rec_steps > (size_t)(max_steps < 0 ? 0 : max_steps)) { L->err = "bmpc_loop_records: more recorded steps than the buffer holds"; return 1; }
    if (L->rec_steps > 0)
        LCHK(L, hipMemcpy(out, L->d_rec, L->rec_steps * L->n_rec * lp_rec_doubles(L->N) * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int bmpc_loop_field(const char* name, int* offset, int* count) { return loop_field_lookup(name, offset, count); }
extern "C" const char* bmpc_loop_last_error(const bmpc_loop* L) { return L ? L->err.c_str() : "null loop"; }

extern "C" void bmpc_loop_destroy(bmpc_loop* L) {
    if (!L) return;
    if (L->st) { (void)hipSetDevice(L->dev); (void)hipStreamSynchronize(L->st); }     // nothing of the loop still in flight
    double* bufs[] = {L->d_S, L->d_prev, L->d_x0, L->d_lbx, L->d_ubx, L->d_p, L->d_x, L->d_f, L->d_viol, L->d_log};
    for (double* b : bufs) if (b) (void)hipFree(b);
    if (L->d_rec) (void)hipFree(L->d_rec);
    if (L->d_rec_slot) (void)hipFree(L->d_rec_slot);
    if (L->d_iters) (void)hipFree(L->d_iters);
    if (L->d_status) (void)hipFree(L->d_status);
    if (L->d_rc) (void)hipFree(L->d_rc);
    if (L->d_scene) (void)hipFree(L->d_scene);
    if (L->d_scene_i) (void)hipFree(L->d_scene_i);
    if (L->d_colres) (void)hipFree(L->d_colres);
    if (L->d_steps_left) (void)hipFree(L->d_steps_left);
    if (L->d_cont) (void)hipFree(L->d_cont);
    if (L->d_prio) (void)hipFree(L->d_prio);
    if (L->e0) (void)hipEventDestroy(L->e0);
    if (L->e1) (void)hipEventDestroy(L->e1);
    if (L->h) bmpc_handle_release(L->h);     // a bmpc_destroy deferred because of this loop runs now
    delete L;
}

extern "C" int bmpc_loop_create(bmpc_handle* h, int R, bmpc_loop** out) {
    if (!h || !out || R <= 0) return 1;
    bmpc_loop* L = new bmpc_loop();
    *out = L;
    L->R = R;
    bmpc_opts o;
    if (bmpc_get_opts(h, &o) != 0) { L->err = "bmpc_get_opts failed"; return 1; }
    if (o.N > LP_NMAX) { L->err = "horizon too long for the device loop"; return 1; }
    L->N = o.N; L->dt = o.dt; L->n_w = 44 * o.N + 6;
    L->dev = o.device;
    LCHK(L, hipSetDevice(o.device));
    RobotConst rc;
    bmpc_robot rob;
    if (bmpc_get_robot(h, &rob) != 0) { L->err = "bmpc_get_robot failed"; return 1; }
    fill_robot_const(rc, rob);                 // the loop works on the robot the handle has at this moment
    LCHK(L, hipMalloc((void**)&L->d_rc, sizeof(RobotConst)));
    LCHK(L, hipMemcpy(L->d_rc, &rc, sizeof(RobotConst), hipMemcpyHostToDevice));
    const size_t nw = (size_t)R * L->n_w * sizeof(double);
    LCHK(L, hipMalloc((void**)&L->d_S, (size_t)R * LS_SIZE * sizeof(double)));
    LCHK(L, hipMemset(L->d_S, 0, (size_t)R * LS_SIZE * sizeof(double)));
    double** rows[] = {&L->d_prev, &L->d_x0, &L->d_lbx, &L->d_ubx, &L->d_x};
    for (double** b : rows) { LCHK(L, hipMalloc((void**)b, nw)); LCHK(L, hipMemset(*b, 0, nw)); }
    LCHK(L, hipMalloc((void**)&L->d_p, (size_t)R * NPAR * sizeof(double)));
    LCHK(L, hipMalloc((void**)&L->d_f, (size_t)R * sizeof(double)));
    LCHK(L, hipMalloc((void**)&L->d_viol, (size_t)R * sizeof(double)));
    LCHK(L, hipMalloc((void**)&L->d_iters, (size_t)R * sizeof(int)));
    LCHK(L, hipMalloc((void**)&L->d_status, (size_t)R * sizeof(int)));
    LCHK(L, hipMemset(L->d_viol, 0, (size_t)R * sizeof(double)));
    LCHK(L, hipMemset(L->d_iters, 0, (size_t)R * sizeof(int)));
    LCHK(L, hipMemset(L->d_status, 0, (size_t)R * sizeof(int)));
    L->h = h;
    bmpc_handle_retain(h);                    // the handle outlives the loop even if the caller destroys it first
    L->st = (hipStream_t)bmpc_stream(h);      // the handle's stream: one stream (one hardware queue) per solver handle
    LCHK(L, hipEventCreate(&L->e0));
    LCHK(L, hipEventCreate(&L->e1));
    const size_t ne = (size_t)R * L->n_w;
    hipLaunchKernelGGL(bmpc_loop_k_bounds, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, L->st, R, L->N, L->d_rc, L->d_lbx, L->d_ubx);
    LCHK(L, hipGetLastError());
    LCHK(L, hipStreamSynchronize(L->st));
    return 0;
}

static bool range_ok(bmpc_loop* L, int first, int count) {
    if (!L || first < 0 || count <= 0 || first + count > L->R) { if (L) L->err = "rollout range out of bounds"; return false; }
    return true;
}

extern "C" int bmpc_loop_upload(bmpc_loop* L, int first, int count, const double* state, const double* prev) {
    if (!range_ok(L, first, count) || !state) return 1;
    LCHK(L, hipSetDevice(L->dev));      // the calling host thread may be new
    LCHK(L, hipMemcpy(L->d_S + (size_t)first * LS_SIZE, state, (size_t)count * LS_SIZE * sizeof(double), hipMemcpyHostToDevice));
    if (prev) LCHK(L, hipMemcpy(L->d_prev + (size_t)first * L->n_w, prev, (size_t)count * L->n_w * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int bmpc_loop_download(bmpc_loop* L, int first, int count, double* state, double* prev) {
    if (!range_ok(L, first, count)) return 1;
    LCHK(L, hipSetDevice(L->dev));      // the calling host thread may be new
    LCHK(L, hipStreamSynchronize(L->st));
    if (state) LCHK(L, hipMemcpy(state, L->d_S + (size_t)first * LS_SIZE, (size_t)count * LS_SIZE * sizeof(double), hipMemcpyDeviceToHost));
    if (prev) LCHK(L, hipMemcpy(prev, L->d_prev + (size_t)first * L->n_w, (size_t)count * L->n_w * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int bmpc_loop_set_obstacles(bmpc_loop* L, int n_obs, const double* A, const double* b, const int* nrows, const double* V,
                                       const int* nv) {
    if (!L || n_obs < 0 || n_obs > LP_MAXOBS || (n_obs > 0 && (!A || !b || !nrows || !V || !nv))) { if (L) L->err = "bmpc_loop_set_obstacles: bad arguments"; return 1; }
    LCHK(L, hipSetDevice(L->dev));
    LCHK(L, hipStreamSynchronize(L->st));
    if (L->d_scene) { (void)hipFree(L->d_scene); L->d_scene = nullptr; }
    if (L->d_scene_i) { (void)hipFree(L->d_scene_i); L->d_scene_i = nullptr; }
    if (L->d_colres) { (void)hipFree(L->d_colres); L->d_colres = nullptr; }
    L->sc = LoopScene{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n_obs == 0) return 0;
    for (int i = 0; i < n_obs; i++)
        if (nrows[i] < 1 || nrows[i] > LP_ROWS || nv[i] < 1 || nv[i] > LP_NV) { L->err = "obstacle with too many rows or vertices"; return 1; }
    const size_t nA = (size_t)n_obs * 45, nb = (size_t)n_obs * LP_ROWS, nAAt = (size_t)n_obs * LP_ROWS * LP_ROWS, nV = (size_t)n_obs * LP_NV * 3;
    const size_t nBox = (size_t)n_obs * 6;
    std::vector<double> h(nA + nb + nAAt + nV + nBox, 0.0);
    std::vector<int> hi(3 * (size_t)n_obs);
    for (int o = 0; o < n_obs; o++) {
        for (int r = 0; r < nrows[o]; r++) {
            for (int c = 0; c < 3; c++) h[45 * o + 3 * r + c] = A[45 * o + 3 * r + c];
            h[nA + LP_ROWS * o + r] = b[LP_ROWS * o + r];
        }
        for (int r = 0; r < nrows[o]; r++)
            for (int q = 0; q < nrows[o]; q++) {
                double sum = 0;
                for (int c = 0; c < 3; c++) sum += A[45 * o + 3 * r + c] * A[45 * o + 3 * q + c];
                h[nA + nb + (size_t)LP_ROWS * LP_ROWS * o + LP_ROWS * r + q] = sum;
            }
        for (int v = 0; v < nv[o]; v++)
            for (int c = 0; c < 3; c++) h[nA + nb + nAAt + 3 * ((size_t)LP_NV * o + v) + c] = V[3 * (LP_NV * o + v) + c];
        hi[o] = nrows[o]; hi[n_obs + o] = nv[o];
        double* bx = h.data() + nA + nb + nAAt + nV + 6 * (size_t)o;
        hi[2 * n_obs + o] = loop_detect_box(A + 45 * o, b + LP_ROWS * o, nrows[o], bx, bx + 3) ? 1 : 0;
    }
    LCHK(L, hipMalloc((void**)&L->d_scene, h.size() * sizeof(double)));
    LCHK(L, hipMemcpy(L->d_scene, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    LCHK(L, hipMalloc((void**)&L->d_scene_i, hi.size() * sizeof(int)));
    LCHK(L, hipMemcpy(L->d_scene_i, hi.data(), hi.size() * sizeof(int), hipMemcpyHostToDevice));
    LCHK(L, hipMalloc((void**)&L->d_colres, (size_t)L->R * 6 * n_obs * LP_CRES * sizeof(double)));
    LCHK(L, hipMemset(L->d_colres, 0, (size_t)L->R * 6 * n_obs * LP_CRES * sizeof(double)));
    L->sc = LoopScene{n_obs, L->d_scene, L->d_scene + nA, L->d_scene + nA + nb, L->d_scene_i, L->d_scene + nA + nb + nAAt, L->d_scene_i + n_obs,
                      L->d_scene + nA + nb + nAAt + nV, L->d_scene_i + 2 * n_obs};
    return 0;
}

// n rollouts at most: all of them (list == nullptr, n == R) or those of a device list
static int launch_prepare(bmpc_loop* L, hipStream_t st, int n, const int* list = nullptr, const int* n_list = nullptr) {
    const size_t ne = (size_t)n * L->n_w;
    if (L->sc.n_obs > 0) {
        hipLaunchKernelGGL(bmpc_loop_k_colpairs, dim3((n * 6 + 63) / 64, L->sc.n_obs), dim3(64), 0, st, L->R, L->d_rc, L->sc, L->d_S,
                           L->d_colres, list, n_list);
    }
    hipLaunchKernelGGL(bmpc_loop_k_prepare, dim3((n + 63) / 64), dim3(64), 0, st, L->R, L->N, L->d_rc, L->d_S, L->d_prev,
                       L->d_p, L->d_lbx, L->d_ubx, L->sc, L->d_colres, list, n_list);
    hipLaunchKernelGGL(bmpc_loop_k_x0, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, L->R, L->N, L->d_S, L->d_prev, L->d_x0,
                       list, n_list);
    LCHK(L, hipGetLastError());
    return 0;
}

static int launch_finish(bmpc_loop* L, hipStream_t st, int n, double* d_log_rows, const int* list = nullptr, const int* n_list = nullptr,
                         int* steps_left = nullptr, int* cont = nullptr, int nsteps = 0, double* d_rec_rows = nullptr) {
    const size_t ne = (size_t)n * L->n_w;
    hipLaunchKernelGGL(bmpc_loop_k_finish, dim3((n + 63) / 64), dim3(64), 0, st, L->R, L->N, L->dt, L->d_rc, L->d_S, L->d_x,
                       L->d_prev, L->d_status, L->d_viol, L->d_iters, d_log_rows, list, n_list, steps_left, cont, nsteps, L->d_p, d_rec_rows, L->d_rec_slot);
    hipLaunchKernelGGL(bmpc_loop_k_keep, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, L->R, L->N, L->d_S, L->d_x, L->d_prev,
                       list, n_list);
    LCHK(L, hipGetLastError());
    return 0;
}

static int ensure_log(bmpc_loop* L, size_t rows) {
    if (rows <= L->log_cap) return 0;
    if (L->d_log) (void)hipFree(L->d_log);
    L->d_log = nullptr; L->log_cap = 0;
    LCHK(L, hipMalloc((void**)&L->d_log, rows * LP_LOGW * sizeof(double)));
    L->log_cap = rows;
    return 0;
}

static int do_solve(bmpc_loop* L) {
    int rc = bmpc_solve_dev(L->h, L->R, L->d_x0, L->d_lbx, L->d_ubx, L->d_p, L->d_x, nullptr, L->d_f, L->d_iters, L->d_status,
                            L->d_viol, (void*)L->st);
    if (rc != 0) { L->err = std::string("bmpc_solve_dev: ") + bmpc_last_error(L->h); return rc; }
    return 0;
}

extern "C" int bmpc_loop_prepare(bmpc_loop* L) {
    if (!L) return 1;
    LCHK(L, hipSetDevice(L->dev));      // the calling host thread may be new
    if (int rc = launch_prepare(L, L->st, L->R)) return rc;
    LCHK(L, hipStreamSynchronize(L->st));
    return 0;
}

extern "C" int bmpc_loop_solve(bmpc_loop* L) { return L ? do_solve(L) : 1; }

extern "C" int bmpc_loop_finish(bmpc_loop* L, double* log) {
    if (!L) return 1;
    LCHK(L, hipSetDevice(L->dev));      // the calling host thread may be new
    if (log) { if (int rc = ensure_log(L, (size_t)L->R)) return rc; }
    if (L->n_rec > 0) {
        const size_t need = (size_t)L->n_rec * lp_rec_doubles(L->N);
        if (need > L->rec_cap) {
            if (L->d_rec) (void)hipFree(L->d_rec);
            L->d_rec = nullptr; L->rec_cap = 0;
            LCHK(L, hipMalloc((void**)&L->d_rec, need * sizeof(double)));
            L->rec_cap = need;
        }
    }
    if (L->n_rec > 0) LCHK(L, hipMemsetAsync(L->d_rec, 0, (size_t)L->n_rec * lp_rec_doubles(L->N) * sizeof(double), L->st));      // (a dead rollout's record stays zero)
    if (int rc = launch_finish(L, L->st, L->R, log ? L->d_log : nullptr, nullptr, nullptr, nullptr, nullptr, 0, L->n_rec > 0 ? L->d_rec : nullptr)) return rc;
    L->rec_steps = L->n_rec > 0 ? 1 : 0;
    LCHK(L, hipStreamSynchronize(L->st));
    if (log) LCHK(L, hipMemcpy(log, L->d_log, (size_t)L->R * LP_LOGW * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int bmpc_loop_run(bmpc_loop* L, int nsteps, double* log, float* ms_total, float* ms_solve) {
    if (!L || nsteps <= 0) return 1;
    LCHK(L, hipSetDevice(L->dev));      // the calling host thread may be new
    if (log) { if (int rc = ensure_log(L, (size_t)nsteps * L->R)) return rc; }
    if (L->n_rec > 0) {
        const size_t need = (size_t)nsteps * L->n_rec * lp_rec_doubles(L->N);
        if (need > L->rec_cap) {
            if (L->d_rec) (void)hipFree(L->d_rec);
            L->d_rec = nullptr; L->rec_cap = 0;
            LCHK(L, hipMalloc((void**)&L->d_rec, need * sizeof(double)));
            L->rec_cap = need;
        }
    }
    double solve_s = 0.0;
    if (L->n_rec > 0) LCHK(L, hipMemsetAsync(L->d_rec, 0, (size_t)nsteps * L->n_rec * lp_rec_doubles(L->N) * sizeof(double), L->st));     // (a dead rollout's record stays zero)
    LCHK(L, hipEventRecord(L->e0, L->st));
    for (int s = 0; s < nsteps; s++) {
        if (int rc = launch_prepare(L, L->st, L->R)) return rc;
        auto t0 = std::chrono::steady_clock::now();
        if (int rc = do_solve(L)) return rc;
        solve_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double* recs = L->n_rec > 0 ? L->d_rec + (size_t)s * L->n_rec * lp_rec_doubles(L->N) : nullptr;
        if (int rc = launch_finish(L, L->st, L->R, log ? L->d_log + (size_t)s * L->R * LP_LOGW : nullptr, nullptr, nullptr, nullptr, nullptr, 0, recs)) return rc;
    }
    L->rec_steps = L->n_rec > 0 ? (size_t)nsteps : 0;
    LCHK(L, hipEventRecord(L->e1, L->st));
    LCHK(L, hipStreamSynchronize(L->st));
    float ms = 0.f;
    LCHK(L, hipEventElapsedTime(&ms, L->e0, L->e1));
    if (ms_total) *ms_total = ms;
    if (ms_solve) *ms_solve = (float)(1e3 * solve_s);
    if (log) LCHK(L, hipMemcpy(log, L->d_log, (size_t)nsteps * L->R * LP_LOGW * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

// ---- closed loop without lock step ---------------------------------------------------------------------------------------
// The rollouts are independent, so nothing but convenience makes them wait for each other: here every rollout is a row of
// ONE hooked solver call (bmpc_capi.hip).  When the solve of a rollout's current step retires, the hook below runs the
// finish / keep / collision-pair / prepare / start-vector kernels for just those rollouts, and the solver re-admits the
// slot with the rollout's next problem at the start of the following super-step.  The straggler tail is then paid once per
// run instead of once per MPC step.  Per-rollout arithmetic does not depend on the schedule: the log equals bmpc_loop_run's.
typedef int (*bmpc_retire_hook)(void* ctx, const int* d_done, const int* d_n_done, int n_max, void* stream);
extern "C" int bmpc_solve_dev_hooked(bmpc_handle* h, int B, const double* d_x0, const double* d_lbx, const double* d_ubx,
                                     const double* d_p, double* d_x, double* d_f, int* d_iters, int* d_status, double* d_viol,
                                     void* stream, bmpc_retire_hook hook, void* hook_ctx, const int* d_cont,
                                     const int* d_prio, int prio_max);

static int loop_retire_hook(void* ctx, const int* d_done, const int* d_n_done, int n_max, void* stream) {
    bmpc_loop* L = (bmpc_loop*)ctx;
    hipStream_t st = (hipStream_t)stream;
    if (n_max > L->R) n_max = L->R;          // the solver's slot count may exceed the rollouts (workspace sized by max_batch)
    if (int rc = launch_finish(L, st, n_max, L->async_log, d_done, d_n_done, L->d_steps_left, L->d_cont, L->async_nsteps)) return rc;
    if (int rc = launch_prepare(L, st, n_max, d_done, d_n_done)) return rc;
    if (L->prio_max > 0) {                   // the steps left have changed: who lags behind now
        hipLaunchKernelGGL(bmpc_loop_k_prio, dim3(1), dim3(1024), 0, st, L->R, (const int*)L->d_steps_left, L->d_prio, L->prio_max);
        LCHK(L, hipGetLastError());
    }
    return 0;
}

extern "C" int bmpc_loop_run_async(bmpc_loop* L, int nsteps, double* log, float* ms_total) {
    if (!L || nsteps <= 0) return 1;
    LCHK(L, hipSetDevice(L->dev));
    if (log) { if (int rc = ensure_log(L, (size_t)nsteps * L->R)) return rc; }
    if (!L->d_steps_left) {
        LCHK(L, hipMalloc((void**)&L->d_steps_left, (size_t)L->R * sizeof(int)));
        LCHK(L, hipMalloc((void**)&L->d_cont, (size_t)L->R * sizeof(int)));
        LCHK(L, hipMalloc((void**)&L->d_prio, (size_t)L->R * sizeof(int)));
    }
    // fast lane for the rollouts that lag behind: at most BMPC_FAST_LANE of them.  Off unless asked for: measured on configs[4]
    // (EXPERIMENTS.md) it does not pay yet -- a super-step of a few dozen instances beside the bulk lane takes 0.85 - 1.1 ms
    // (0.64 ms on an empty GPU), no faster than the bulk lane's own once half of the rollouts have finished
    { const char* e = getenv("BMPC_FAST_LANE"); L->prio_max = e ? atoi(e) : 0; }
    LCHK(L, hipMemsetAsync(L->d_prio, 0, (size_t)L->R * sizeof(int), L->st));
    std::vector<int> left((size_t)L->R, nsteps);
    LCHK(L, hipMemcpyAsync(L->d_steps_left, left.data(), left.size() * sizeof(int), hipMemcpyHostToDevice, L->st));
    LCHK(L, hipMemsetAsync(L->d_cont, 0, (size_t)L->R * sizeof(int), L->st));
    LCHK(L, hipStreamSynchronize(L->st));       // `left` goes out of scope below
    L->async_nsteps = nsteps;
    L->async_log = log ? L->d_log : nullptr;
    LCHK(L, hipEventRecord(L->e0, L->st));
    if (int rc = launch_prepare(L, L->st, L->R)) return rc;         // step 0 of every rollout
    int rc = bmpc_solve_dev_hooked(L->h, L->R, L->d_x0, L->d_lbx, L->d_ubx, L->d_p, L->d_x, L->d_f, L->d_iters, L->d_status, L->d_viol,
                                   (void*)L->st, loop_retire_hook, L, L->d_cont, L->prio_max > 0 ? L->d_prio : nullptr, L->prio_max);
    if (rc != 0) { L->err = std::string("bmpc_solve_dev_hooked: ") + bmpc_last_error(L->h); return rc; }
    LCHK(L, hipEventRecord(L->e1, L->st));
    LCHK(L, hipStreamSynchronize(L->st));
    float ms = 0.f;
    LCHK(L, hipEventElapsedTime(&ms, L->e0, L->e1));
    if (ms_total) *ms_total = ms;
    if (log) LCHK(L, hipMemcpy(log, L->d_log, (size_t)nsteps * L->R * LP_LOGW * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int bmpc_loop_problem(bmpc_loop* L, double* x0, double* lbx, double* ubx, double* p) {
    if (!L) return 1;
    LCHK(L, hipSetDevice(L->dev));      // the calling host thread may be new
    LCHK(L, hipStreamSynchronize(L->st));
    const size_t nw = (size_t)L->R * L->n_w * sizeof(double);
    if (x0) LCHK(L, hipMemcpy(x0, L->d_x0, nw, hipMemcpyDeviceToHost));
    if (lbx) LCHK(L, hipMemcpy(lbx, L->d_lbx, nw, hipMemcpyDeviceToHost));
    if (ubx) LCHK(L, hipMemcpy(ubx, L->d_ubx, nw, hipMemcpyDeviceToHost));
    if (p) LCHK(L, hipMemcpy(p, L->d_p, (size_t)L->R * NPAR * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int bmpc_loop_solution(bmpc_loop* L, double* x, int* iters, int* status, double* viol) {
    if (!L) return 1;
    LCHK(L, hipSetDevice(L->dev));      // the calling host thread may be new
    LCHK(L, hipStreamSynchronize(L->st));
    if (x) LCHK(L, hipMemcpy(x, L->d_x, (size_t)L->R * L->n_w * sizeof(double), hipMemcpyDeviceToHost));
    if (iters) LCHK(L, hipMemcpy(iters, L->d_iters, (size_t)L->R * sizeof(int), hipMemcpyDeviceToHost));
    if (status) LCHK(L, hipMemcpy(status, L->d_status, (size_t)L->R * sizeof(int), hipMemcpyDeviceToHost));
    if (viol) LCHK(L, hipMemcpy(viol, L->d_viol, (size_t)L->R * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int bmpc_loop_set_solution(bmpc_loop* L, const double* x, const int* iters, const int* status, const double* viol) {
    if (!L || !x || !status || !viol) return 1;
    LCHK(L, hipSetDevice(L->dev));      // the calling host thread may be new
    LCHK(L, hipMemcpy(L->d_x, x, (size_t)L->R * L->n_w * sizeof(double), hipMemcpyHostToDevice));
    if (iters) LCHK(L, hipMemcpy(L->d_iters, iters, (size_t)L->R * sizeof(int), hipMemcpyHostToDevice));
    LCHK(L, hipMemcpy(L->d_status, status, (size_t)L->R * sizeof(int), hipMemcpyHostToDevice));
    LCHK(L, hipMemcpy(L->d_viol, viol, (size_t)L->R * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}
