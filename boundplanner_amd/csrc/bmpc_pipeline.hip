// Kernels of the batch-synchronous pipeline engine (gfx950) + their launch functions.
// Device code: bmpc_stage.hpp / bmpc_pair_kernels.hpp / bmpc_ric_kernel.hpp.  Host sequencing:
// bmpc_capi.hip (pipe_solve).
#include "bmpc_platform_hip.hpp"
#include <cstdlib>

#define BMPC_NT 64
#include "bmpc_pair_kernels.hpp"
#include "bmpc_ric_kernel.hpp"

using namespace bmpc;

// waves per SIMD the register allocator must leave room for (latency hiding vs spills)
#ifndef BMPC_PAIR_WPS
#define BMPC_PAIR_WPS 1
#endif
#ifndef BMPC_EVAL_WPS
#define BMPC_EVAL_WPS BMPC_PAIR_WPS
#endif
#ifndef BMPC_TRIAL_NW
#define BMPC_TRIAL_NW 1      // wavefronts per group of pairs in k_trial: 1 = one walks all rows, 4 = one part of the walk each (measured, not kept: EXPERIMENTS.md)
#endif
#ifndef BMPC_TRIAL_WPS
#define BMPC_TRIAL_WPS (BMPC_TRIAL_NW == 4 ? 2 : BMPC_PAIR_WPS)
#endif
#ifndef BMPC_STEP_WPS
#define BMPC_STEP_WPS BMPC_PAIR_WPS
#endif
#ifndef BMPC_POINTS_WPS
#define BMPC_POINTS_WPS BMPC_PAIR_WPS
#endif
#ifndef BMPC_RIC_WPS
#define BMPC_RIC_WPS 1
#endif

// the launch argument is the host view of the argument block; device code reads it through the
// layout-identical view whose pointers are global-address-space qualified
#define DV(H) (*reinterpret_cast<const PipeArgs*>(&(H)))

// thread-per-pair kernels: one wavefront per workgroup, floor(64/(N-1)) instances per wavefront
__global__ __launch_bounds__(64) void bmpc_k_init_inst(PipeArgsH H) { k_init_inst_body(DV(H), blockIdx.x * 64 + threadIdx.x); }
// dynamic LDS of the thread-per-pair kernels: [emitter tile (k_eval, k_curv)] [staged parameter vectors]
extern __shared__ __attribute__((aligned(16))) double bmpc_dyn_lds[];
__global__ __launch_bounds__(64, BMPC_PAIR_WPS) void bmpc_k_init(PipeArgsH H) { k_init_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds); }
__global__ __launch_bounds__(64, BMPC_PAIR_WPS) void bmpc_k_pose(PipeArgsH H) { k_pose_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds); }
__global__ __launch_bounds__(64, BMPC_EVAL_WPS) void bmpc_k_eval(PipeArgsH H) {
    k_eval_body<0>(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds);
}
__global__ __launch_bounds__(64, BMPC_POINTS_WPS) void bmpc_k_points(PipeArgsH H) { k_points_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds); }
__global__ __launch_bounds__(64, BMPC_PAIR_WPS) void bmpc_k_curv(PipeArgsH H) {
    __shared__ double lds[EM_DOUBLES + 8];
    k_curv_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)lds);
}
// Two pairs of kernels of a super-step do not depend on each other -- k_points / k_pose (both read the iterate, write disjoint side
// fields) and k_eval / k_curv (disjoint fields of the stage record) -- and are launched as ONE grid each: the first nw workgroups
// run one body, the rest the other.  Same work in the bulk regime, two launches fewer; in the straggler tail, where a launch costs
// its single-thread latency, the two bodies run side by side (k_points 39 + k_pose 41 us -> 41, k_eval 113 + k_curv ~50 -> 113).
// BMPC_SPLIT_LAUNCHES=1 in the environment keeps the four launches (per-kernel profiles).
__global__ __launch_bounds__(64, BMPC_PAIR_WPS) void bmpc_k_points_pose(PipeArgsH H, int nw) {
    if ((int)blockIdx.x < nw) k_points_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds);
    else k_pose_body(DV(H), (int)blockIdx.x - nw, threadIdx.x, (LDSD*)bmpc_dyn_lds);
}
__global__ __launch_bounds__(64, BMPC_EVAL_WPS) void bmpc_k_eval_curv(PipeArgsH H, int nw) {
    if ((int)blockIdx.x < nw) k_eval_body<0>(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds);
    else k_curv_body(DV(H), (int)blockIdx.x - nw, threadIdx.x, (LDSD*)bmpc_dyn_lds);
}
// k_eval as two wavefronts side by side (up to BMPC_EVAL_SPLIT_WGS groups of pairs: always, by default) -- everything but the chained block /
// the chained block alone (bmpc_pair_kernels.hpp, k_eval_body) -- beside k_curv
__global__ __launch_bounds__(64, BMPC_EVAL_WPS) void bmpc_k_eval_curv_split(PipeArgsH H, int nw) {
    if ((int)blockIdx.x < nw) k_eval_body<1>(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds);
    else if ((int)blockIdx.x < 2 * nw) k_eval_body<2>(DV(H), (int)blockIdx.x - nw, threadIdx.x, (LDSD*)bmpc_dyn_lds);
    else k_curv_body(DV(H), (int)blockIdx.x - 2 * nw, threadIdx.x, (LDSD*)bmpc_dyn_lds);
}
// (the two bodies as kernels of their own: per-kernel profiles, BMPC_SPLIT_LAUNCHES=1)
__global__ __launch_bounds__(64, BMPC_EVAL_WPS) void bmpc_k_eval_main(PipeArgsH H) { k_eval_body<1>(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds); }
__global__ __launch_bounds__(64, BMPC_EVAL_WPS) void bmpc_k_eval_chain(PipeArgsH H) { k_eval_body<2>(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds); }
#ifndef BMPC_EVAL_SPLIT_WGS
// groups of pairs up to which bmpc_k_eval_curv_split replaces bmpc_k_eval_curv (0 = never).  Always: built for the latency of the tail
// regime (a third of k_eval's arithmetic beside the rest), the two lighter bodies also beat the one-wavefront kernel in the bulk regime
// (544 - 566 -> 493 - 497 us at 8192 live, bench 124.3 - 125.0 -> 127.4 - 127.9 k solves/s in an ABAB run on one box)
#define BMPC_EVAL_SPLIT_WGS (1 << 30)
#endif
#ifndef BMPC_RIC_NT
#define BMPC_RIC_NT 128     // lanes cooperating on one instance in the Riccati kernel
#endif
#ifndef BMPC_RIC_SPEC_BELOW
#define BMPC_RIC_SPEC_BELOW 512     // live instances below which bmpc_k_ric_att(_thr) + bmpc_k_ric_sel replace bmpc_k_ric_lat / bmpc_k_ric (2 .. 5 attempts per instance)
#endif
#ifndef BMPC_RIC_WPE
#define BMPC_RIC_WPE 3      // wavefronts per SIMD the throughput variant is compiled for
#endif
// disable_tail_calls: a call site marked `tail` (LLVM marks every call that is handed no pointer into the caller's stack frame)
// makes the callee save and restore the 64 callee-saved VGPRs it uses (the register-usage propagation that lets the sweeps clobber
// them is skipped for functions with such call sites): 128 scratch accesses per lane and sweep
__global__ __launch_bounds__(BMPC_RIC_NT, BMPC_RIC_WPE) __attribute__((disable_tail_calls)) void bmpc_k_ric(PipeArgsH H) {          // throughput variant: 6 workgroups per CU (3 wavefronts / SIMD)
    __shared__ __attribute__((aligned(16))) double lds[RIC_LDS_DOUBLES];
    k_ric_body<BMPC_RIC_NT, true>(ric_kernel_args(), blockIdx.x, threadIdx.x, (LDSD*)lds);
}
__global__ __launch_bounds__(BMPC_RIC_NT, 1) __attribute__((disable_tail_calls)) void bmpc_k_ric_lat(PipeArgsH H) {      // latency variant for the straggler tail
    __shared__ __attribute__((aligned(16))) double lds[RIC_LDS_DOUBLES];
    k_ric_body<BMPC_RIC_NT, false>(ric_kernel_args(), blockIdx.x, threadIdx.x, (LDSD*)lds);
}
// the speculative pair for the deep tail (few instances alive, each on a CU of its own): RIC_NATT factorisation attempts of every
// instance side by side, then the selection + forward start (bmpc_ric_kernel.hpp)
__global__ __launch_bounds__(BMPC_RIC_NT, 1) __attribute__((disable_tail_calls)) void bmpc_k_ric_att(PipeArgsH H) {
    __shared__ __attribute__((aligned(16))) double lds[RIC_LDS_DOUBLES];
    k_ric_att_body<BMPC_RIC_NT, false>(ric_kernel_args(), blockIdx.x, threadIdx.x, (LDSD*)lds);
}
// (the same with the throughput compilation of the sweeps: more attempts than 512 workgroups -- two per CU -- can hold)
__global__ __launch_bounds__(BMPC_RIC_NT, BMPC_RIC_WPE) __attribute__((disable_tail_calls)) void bmpc_k_ric_att_thr(PipeArgsH H) {
    __shared__ __attribute__((aligned(16))) double lds[RIC_LDS_DOUBLES];
    k_ric_att_body<BMPC_RIC_NT, true>(ric_kernel_args(), blockIdx.x, threadIdx.x, (LDSD*)lds);
}
__global__ __launch_bounds__(BMPC_RIC_NT, 1) __attribute__((disable_tail_calls)) void bmpc_k_ric_sel(PipeArgsH H) {
    __shared__ __attribute__((aligned(16))) double lds[RIC_LDS_DOUBLES];
    k_ric_body<BMPC_RIC_NT, false, true>(ric_kernel_args(), blockIdx.x, threadIdx.x, (LDSD*)lds);
}
#ifndef BMPC_RIC_LAT_BELOW
#define BMPC_RIC_LAT_BELOW 512      // fewer active instances than this: the latency variant (every wavefront has a SIMD to itself anyway)
#endif
__global__ __launch_bounds__(64) void bmpc_k_fwd(PipeArgsH H) {
    __shared__ __attribute__((aligned(16))) double lds[FW_LDS_DOUBLES];
    k_fwd_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)lds);
}
__global__ __launch_bounds__(64, BMPC_STEP_WPS) void bmpc_k_step(PipeArgsH H) { k_step_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds); }
__global__ __launch_bounds__(64) void bmpc_k_init_fin(PipeArgsH H) { k_init_fin_body(DV(H), blockIdx.x * 64 + threadIdx.x); }
__global__ __launch_bounds__(64) void bmpc_k_admit(PipeArgsH H) { k_admit_body(DV(H), blockIdx.x * 64 + threadIdx.x); }
__global__ void bmpc_k_pool_reset(PipeArgsH H, int done_too) { if (threadIdx.x == 0 && blockIdx.x == 0) k_pool_reset_body(DV(H), done_too != 0); }
__global__ __launch_bounds__(64 * BMPC_TRIAL_NW, BMPC_TRIAL_WPS) void bmpc_k_trial(PipeArgsH H) {
    k_trial_body_t<BMPC_TRIAL_NW>(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds);
}
// the line search of the tail regime: four step lengths side by side, one wavefront (one SIMD) each (bmpc_pair_kernels.hpp)
__global__ __launch_bounds__(64 * TRIAL_SPEC, 1) void bmpc_k_trial_spec(PipeArgsH H) {
    k_trial_spec_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds);
}
#ifndef BMPC_TRIAL_SPEC_WGS
#define BMPC_TRIAL_SPEC_WGS 256       // groups of pairs up to which bmpc_k_trial_spec replaces bmpc_k_trial: one workgroup (four SIMDs) per CU (0 = never)
#endif
__global__ void bmpc_k_rotate(PipeArgsH H) { if (threadIdx.x == 0 && blockIdx.x == 0) k_rotate_body(DV(H)); }
__global__ __launch_bounds__(64, BMPC_PAIR_WPS) void bmpc_k_out(PipeArgsH H) { k_out_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds); }
__global__ __launch_bounds__(64, BMPC_PAIR_WPS) void bmpc_k_mult(PipeArgsH H) { k_mult_body(DV(H), blockIdx.x, threadIdx.x, (LDSD*)bmpc_dyn_lds); }
__global__ __launch_bounds__(64) void bmpc_k_mult_sweep(PipeArgsH H) { k_mult_sweep_body(DV(H), blockIdx.x * 64 + threadIdx.x); }
__global__ __launch_bounds__(64) void bmpc_k_fin(PipeArgsH H) { k_fin_body(DV(H), blockIdx.x * 64 + threadIdx.x); }

// Closed loop without lock step, two lanes (bmpc_capi.hip, pipe_solve): the live instances -- eval and trial lists of both lanes --
// are dealt out again between the bulk lane H0 and the fast lane H1 by the priority flag of their row (prio[row] != 0: the
// rollout lags behind, its instances iterate at the cadence of a nearly empty GPU).  Destination: the *_next lists (empty between
// super-steps); bmpc_k_rotate of either lane then makes them current.  Per-instance arithmetic does not depend on the lane.
__global__ __launch_bounds__(64) void bmpc_k_pick(PipeArgsH H0, PipeArgsH H1, const int* prio) {
    const PipeArgs& A0 = DV(H0);
    const PipeArgs& A1 = DV(H1);
    const int e = blockIdx.x * 64 + threadIdx.x;
    for (int src = 0; src < 4; src++) {
        const PipeArgs& S = (src & 1) ? A1 : A0;
        const bool trial = src >= 2;
        if (e >= S.L.cnt[trial ? 2 : 0]) continue;
        const int b = (trial ? S.L.trial : S.L.eval)[e];
        const PipeArgs& T = prio[A0.src[b]] ? A1 : A0;
        if (trial) { const int pos = BMPC_ATOMIC_INC(T.L.cnt + 4); T.L.trial_next[pos] = b; }
        else { const int pos = BMPC_ATOMIC_INC(T.L.cnt + 3); T.L.eval_next[pos] = b; }
    }
}

#define LAUNCH(kern, nb, nt)                                            \
    do {                                                                \
        if ((nb) > 0) hipLaunchKernelGGL(kern, dim3(nb), dim3(nt), 0, st, *A); \
    } while (0)
#define LAUNCH_DYN(kern, nb, nt, lds_doubles)                           \
    do {                                                                \
        if ((nb) > 0) hipLaunchKernelGGL(kern, dim3(nb), dim3(nt), (lds_doubles) * sizeof(double), st, *A); \
    } while (0)

// first fill of the pool: n0 = min(B, slots) slots take the first n0 input rows (cnt[0] = cnt[6] = cnt[9] = n0 set by the host)
extern "C" hipError_t bmpc_pipe_launch_init(const PipeArgsH* A, int n0, hipStream_t st) {
    LAUNCH(bmpc_k_init_inst, (n0 + 63) / 64, 64);
    LAUNCH_DYN(bmpc_k_init, waves_for(A->N, n0), 64, pair_lds_doubles(A->N, false));
    LAUNCH(bmpc_k_init_fin, (n0 + 63) / 64, 64);
    hipLaunchKernelGGL(bmpc_k_pool_reset, dim3(1), dim3(64), 0, st, *A, 0);
    return hipGetLastError();
}

// retirement of the instances that finished (at most n_max), first half: outputs in the reference layout
extern "C" hipError_t bmpc_pipe_launch_retire_out(const PipeArgsH* A, int n_max, hipStream_t st) {
    LAUNCH_DYN(bmpc_k_out, waves_for(A->N, n_max), 64, pair_lds_doubles(A->N, false));
    LAUNCH(bmpc_k_fin, (n_max + 63) / 64, 64);
    return hipGetLastError();
}
// second half: while input rows are left (or, closed loop, the row has its next problem ready) the slots are refilled and
// initialised.  A solve of B <= slots instances without a retire hook never refills.
extern "C" hipError_t bmpc_pipe_launch_retire_admit(const PipeArgsH* A, int n_max, int refill, hipStream_t st) {
    LAUNCH(bmpc_k_admit, (n_max + 63) / 64, 64);
    if (refill) {
        LAUNCH_DYN(bmpc_k_init, waves_for(A->N, n_max), 64, pair_lds_doubles(A->N, false));
        LAUNCH(bmpc_k_init_fin, (n_max + 63) / 64, 64);
    }
    hipLaunchKernelGGL(bmpc_k_pool_reset, dim3(1), dim3(64), 0, st, *A, 1);
    return hipGetLastError();
}

// one super-step for at most n_act active instances; swaps the double-buffered lists in *A
// e0 / e1 (optional): events recorded around the Riccati launch (bmpc_debug_time_ric); *was_lat: which variant was launched
extern "C" hipError_t bmpc_pipe_launch_step(PipeArgsH* A, int n_act, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int* was_lat) {
    const int nw = waves_for(A->N, n_act);
    static const int split_launches = [] { const char* e = getenv("BMPC_SPLIT_LAUNCHES"); return e ? atoi(e) : 0; }();
    if (split_launches) {
        LAUNCH_DYN(bmpc_k_points, nw, 64, pair_lds_doubles(A->N, false));
        LAUNCH_DYN(bmpc_k_pose, nw, 64, pair_lds_doubles(A->N, false));
        static const int eval_split_wgs_s = [] { const char* e = getenv("BMPC_EVAL_SPLIT_WGS"); return e ? atoi(e) : BMPC_EVAL_SPLIT_WGS; }();
        if (nw <= eval_split_wgs_s) {
            LAUNCH_DYN(bmpc_k_eval_main, nw, 64, pair_lds_doubles(A->N, true));
            LAUNCH_DYN(bmpc_k_eval_chain, nw, 64, pair_lds_doubles(A->N, true));
        } else
        LAUNCH_DYN(bmpc_k_eval, nw, 64, pair_lds_doubles(A->N, true));
        LAUNCH(bmpc_k_curv, nw, 64);
    } else if (nw > 0) {
        hipLaunchKernelGGL(bmpc_k_points_pose, dim3(2 * nw), dim3(64), pair_lds_doubles(A->N, false) * sizeof(double), st, *A, nw);
        // BMPC_EVAL_SPLIT_WGS (read once; 0 = never)
        static const int eval_split_wgs = [] { const char* e = getenv("BMPC_EVAL_SPLIT_WGS"); return e ? atoi(e) : BMPC_EVAL_SPLIT_WGS; }();
        if (nw <= eval_split_wgs)
            hipLaunchKernelGGL(bmpc_k_eval_curv_split, dim3(A->o.hess == 2 ? 3 * nw : 2 * nw), dim3(64), pair_lds_doubles(A->N, true) * sizeof(double), st, *A, nw);
        else
        hipLaunchKernelGGL(bmpc_k_eval_curv, dim3(A->o.hess == 2 ? 2 * nw : nw), dim3(64), pair_lds_doubles(A->N, true) * sizeof(double), st, *A, nw);
    }
    // BMPC_RIC_LAT_BELOW in the environment (read once): A/B runs and the test that the two variants agree bitwise
    static const int lat_below = [] { const char* e = getenv("BMPC_RIC_LAT_BELOW"); return e ? atoi(e) : BMPC_RIC_LAT_BELOW; }();
    if (e0) (void)hipEventRecord(e0, st);
    // BMPC_RIC_SPEC_BELOW (read once; 0 = never): below it the factorisation attempts of an iteration run side by side
    static const int spec_below = [] { const char* e = getenv("BMPC_RIC_SPEC_BELOW"); return e ? atoi(e) : BMPC_RIC_SPEC_BELOW; }();
    if (n_act < spec_below && n_act > 0) {
        // as many attempts per instance as the chip holds at once: up to 512 workgroups with the latency compilation of the sweeps
        // (two per CU), up to 1536 with the throughput compilation (six per CU)
        static const int natt_env = [] { const char* e = getenv("BMPC_RIC_NATT"); return e ? atoi(e) : 0; }();
        int natt = natt_env > 0 ? natt_env : (n_act * RIC_NATT <= 512 ? RIC_NATT : 1536 / n_act);
        A->natt = natt < 2 ? 2 : (natt > RIC_NATT ? RIC_NATT : natt);
        if (n_act * A->natt <= 512) LAUNCH(bmpc_k_ric_att, n_act * A->natt, BMPC_RIC_NT);
        else LAUNCH(bmpc_k_ric_att_thr, n_act * A->natt, BMPC_RIC_NT);
        LAUNCH(bmpc_k_ric_sel, n_act, BMPC_RIC_NT);
    } else if (n_act < lat_below) LAUNCH(bmpc_k_ric_lat, n_act, BMPC_RIC_NT);
    else LAUNCH(bmpc_k_ric, n_act, BMPC_RIC_NT);
    if (e1) (void)hipEventRecord(e1, st);
    if (was_lat) *was_lat = n_act < lat_below || n_act < spec_below;      // (the launches of the tail regime, whichever kernels ran)
    LAUNCH(bmpc_k_fwd, n_act, 64);
    LAUNCH_DYN(bmpc_k_step, nw, 64, pair_lds_doubles(A->N, false));
    // (BMPC_TRIAL_REPEATS in the environment, read once, overrides bmpc_opts.trial_repeats: A/B runs)
    static const int env_repeats = [] { const char* e = getenv("BMPC_TRIAL_REPEATS"); return e ? atoi(e) : -1; }();
    if (env_repeats >= 0) A->o.trial_repeats = env_repeats;
    // BMPC_TRIAL_SPEC_WGS (read once; 0 = never): up to that many groups of pairs the step lengths of a line search are tried side by side (slot-major layout)
    static const int trial_spec_wgs = [] { const char* e = getenv("BMPC_TRIAL_SPEC_WGS"); return e ? atoi(e) : BMPC_TRIAL_SPEC_WGS; }();
    if (nw <= trial_spec_wgs && A->NP == (size_t)(A->N - 1)) LAUNCH_DYN(bmpc_k_trial_spec, nw, 64 * TRIAL_SPEC, trial_lds_doubles(A->N, TRIAL_SPEC));
    else
    LAUNCH_DYN(bmpc_k_trial, nw, 64 * BMPC_TRIAL_NW, trial_lds_doubles(A->N, BMPC_TRIAL_NW));      // trial points (+ multiplier update) + filter test, backtracking inside
    LAUNCH(bmpc_k_rotate, 1, 64);
    int* t = A->L.eval; A->L.eval = A->L.eval_next; A->L.eval_next = t;
    t = A->L.trial; A->L.trial = A->L.trial_next; A->L.trial_next = t;
    return hipGetLastError();
}

// two-lane closed loop: deal the (at most n_max) live instances out between the lanes; swaps the double-buffered lists of both
extern "C" hipError_t bmpc_pipe_launch_pick(PipeArgsH* A0, PipeArgsH* A1, const int* prio, int n_max, hipStream_t st) {
    if (n_max <= 0) return hipSuccess;
    hipLaunchKernelGGL(bmpc_k_pick, dim3((n_max + 63) / 64), dim3(64), 0, st, *A0, *A1, prio);
    for (PipeArgsH* A : {A0, A1}) {
        hipLaunchKernelGGL(bmpc_k_rotate, dim3(1), dim3(64), 0, st, *A);
        int* t = A->L.eval; A->L.eval = A->L.eval_next; A->L.eval_next = t;
        t = A->L.trial; A->L.trial = A->L.trial_next; A->L.trial_next = t;
    }
    return hipGetLastError();
}

// multipliers of the solve whose final iterate is still in the workspace (A->lam_g, A->lam_x: device outputs)
extern "C" hipError_t bmpc_pipe_launch_mult(const PipeArgsH* A, hipStream_t st) {
    LAUNCH_DYN(bmpc_k_mult, waves_for(A->N, A->B), 64, pair_lds_doubles(A->N, false));
    LAUNCH(bmpc_k_mult_sweep, (A->B + 63) / 64, 64);
    return hipGetLastError();
}

extern "C" void bmpc_pipe_build_table(int* tbl) { build_scatter_table(tbl); }
extern "C" int bmpc_pipe_hrec(void) { return HREC; }
extern "C" int bmpc_pipe_krec(void) { return KREC; }
extern "C" int bmpc_pipe_npart(void) { return NPART; }
extern "C" size_t bmpc_pipe_state_bytes(void) { return sizeof(InstState); }
