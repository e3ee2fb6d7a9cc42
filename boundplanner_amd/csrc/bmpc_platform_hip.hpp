// Platform bindings of the device code for gfx950 (HIP).  The device code in bmpc_device.hpp is
// written against these few names so that tests/emu can run the identical source on CPU threads
// for debugging (test infrastructure only; the product always compiles this header with hipcc).
#pragma once
#include <hip/hip_runtime.h>

#define BMPC_DEV __device__
#define BMPC_INL __device__ __forceinline__
#define BMPC_SYNC() __syncthreads()
#define BMPC_LANE() ((int)threadIdx.x)
#define BMPC_BLOCK() ((int)blockIdx.x)
#define BMPC_NBLOCKS() ((int)gridDim.x)
#define BMPC_HD __host__ __device__ inline
#define BMPC_ATOMIC_INC(ptr) atomicAdd((ptr), 1)
// doubles that live in LDS: address-space-qualified so that every access is a ds_* instruction
// (generic pointers to __shared__ memory compile to flat_* loads through the vector-memory path)
typedef __attribute__((address_space(3))) double LDSD;
#ifdef BMPC_NO_NOINLINE
#define BMPC_NOINL __device__
#else
#define BMPC_NOINL __device__ __attribute__((noinline))
#endif
