// Platform bindings of the device code for gfx950 (HIP).  The device code in bmpc_device.hpp is
// written against these few names so that tests/emu can run the identical source on CPU threads
// for debugging (test infrastructure only; the product always compiles this header with hipcc).
#pragma once
#include <hip/hip_runtime.h>

#define BMPC_DEV __device__
// the Riccati kernel reads its argument block from the kernel-argument segment (bmpc_ric_kernel.hpp, RicArgs)
#define BMPC_KERNARG_ARGS 1
#define BMPC_INL __device__ __forceinline__
// kernel bodies: always inlined into their __global__ wrapper.  As separate functions two of them (k_init, k_step) never returned on
// gfx950: their early exit is a long branch, and LLVM's branch relaxation expands it through s[30:31] -- the return address of a leaf
// function, which nothing has saved -- so a wavefront without work "returns" to the exit block for ever
// (tools/repro_noinline_hang/README.md; -DBMPC_KBODY_CALL rebuilds that variant for the static check there)
#ifdef BMPC_KBODY_CALL
#define BMPC_KBODY __device__ __attribute__((noinline))
#else
#define BMPC_KBODY __device__ __forceinline__
#endif
#define BMPC_SYNC() __syncthreads()
// optimisation barrier on a double in a vector register: the value after it is a new one for the register allocator
// (used to end the live range of a spilled value and start a register-resident copy for a hot block)
#define BMPC_PIN(x) asm volatile("" : "+v"(x))
// an integer the compiler must treat as new from here on (32-bit vector register): keeps per-lane index arithmetic that
// depends on it inside the loop iteration instead of hoisting hundreds of offsets out of the loop
#define BMPC_OPAQUE_I(x) asm volatile("" : "+v"(x))
// barrier after which the GLOBAL-memory writes of the workgroup's threads are visible to each other
#define BMPC_FENCE_SYNC() do { __threadfence_block(); __syncthreads(); } while (0)
// nothing is scheduled across this point: keeps the loads of a later phase of a long kernel from being hoisted into an earlier one,
// where they would only lengthen live ranges (the thread-per-pair kernels live at the edge of the register file)
#define BMPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define BMPC_LANE() ((int)threadIdx.x)
// a value that is the same in every lane, moved to a scalar register (frees a vector register across calls)
#define BMPC_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define BMPC_BLOCK() ((int)blockIdx.x)
#define BMPC_NBLOCKS() ((int)gridDim.x)
#define BMPC_HD __host__ __device__ inline
#define BMPC_ATOMIC_INC(ptr) __hip_atomic_fetch_add((ptr), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// doubles that live in LDS: address-space-qualified so that every access is a ds_* instruction
// (generic pointers to __shared__ memory compile to flat_* loads through the vector-memory path)
typedef __attribute__((address_space(3))) double LDSD;
#ifdef BMPC_NO_NOINLINE
#define BMPC_NOINL __device__
#else
#define BMPC_NOINL static __device__ __attribute__((noinline))
#endif

// 16-byte LDS vector access (ds_read_b128 / ds_write_b128) and fast reciprocal square root
typedef double bmpc_v2d __attribute__((vector_size(16)));
typedef __attribute__((address_space(3))) bmpc_v2d LDSV2;
#define BMPC_RSQRT(x) rsqrt(x)
// reciprocal for the row arithmetic (hundreds per thread and kernel): v_rcp_f64 + two Newton steps (5 instructions, last-bit
// accurate to ~1 ulp) instead of the ~25-instruction IEEE division sequence
__device__ __forceinline__ double bmpc_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
#define BMPC_RCP(x) bmpc_rcp(x)
#define BMPC_MUL24(a, b) __mul24((a), (b))       // full-rate 24-bit integer multiply (v_mul_lo_u32 runs at a quarter of the rate)
// sin and cos of a joint angle together: ONE Cody-Waite reduction by pi/2 (two-term, exact to 1e-16 for the |x| < 1e5 that joint
// angles and line-search trial points stay within) and the fdlibm kernel polynomials on [-pi/4, pi/4] -- ~45 instructions for the
// pair, against ~260 for the library's separate sin(x), cos(x) with their large-argument paths (14 calls per kinematic chain, which
// every thread-per-pair kernel evaluates: 9 % of their instructions)
__device__ __forceinline__ void bmpc_sincos(double x, double& s, double& c) {
    const double kd = rint(x * 0.63661977236758138);                 // 2 / pi
    double r = fma(-kd, 1.5707963267948966, x);
    r = fma(-kd, 6.123233995736766e-17, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                         -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                         2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double sr = fma(r * z, ps, r);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)kd & 3;
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}
#define BMPC_SINCOS(x, s, c) bmpc_sincos((x), (s), (c))
// LDS atomic add without return (ds_add_f64): used where every address receives at most one add per
// phase, so the result does not depend on the order
#define BMPC_LDS_ADD(ptr, v) __hip_atomic_fetch_add((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// global-memory pointers are address-space qualified in device code so that every access is a
// global_* instruction (generic pointers compile to flat_*, which also count against the LDS counter)
#ifndef BMPC_AS1
#define BMPC_AS1 __attribute__((address_space(1)))
#endif
// asynchronous global -> LDS copy of NCH chunks of 1 KiB (LDS-DMA, no registers) by a workgroup of NT
// lanes: each wavefront instruction moves one chunk (lane L of the wavefront: 16 bytes to chunk + L*16).
// Completion: BMPC_ASYNC_WAIT() in every wavefront, then a barrier.
template <int NCH, int NT>
__device__ __forceinline__ void bmpc_async_copy(__attribute__((address_space(1))) const double* gsrc, LDSD* lds_dst, int lane) {
    const int wave = lane >> 6, wl = lane & 63;
#pragma unroll
    for (int i = 0; i < (NCH + NT / 64 - 1) / (NT / 64); i++) {
        const int c = i * (NT / 64) + wave;
        if (c < NCH)
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)(gsrc + 2 * wl + 128 * c),
                                             (__attribute__((address_space(3))) void*)(lds_dst + 128 * c), 16, 0, 0);
    }
}
#define BMPC_ASYNC_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// bring the cache line of a global address into the L2 without a register and without anyone waiting for it: a 4-byte
// LDS-DMA load per lane into a junk area of LDS (64 lanes x 4 bytes from lds_junk on)
#define BMPC_TOUCH_LINE(gptr, lds_junk) \
    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lds_junk), 4, 0, 0)
