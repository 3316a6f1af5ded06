// Shared device/host helpers for libkaolin_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#define KAMD_WAVE 64
#define KAMD_NUM_CU 256

// Every entry point returns hipError_t as int; a launch error is picked up right
// after enqueue (same place the reference calls AT_CUDA_CHECK(cudaGetLastError())).
#define KAMD_RETURN_LAST_ERROR() return (int)hipGetLastError()
#define KAMD_CHECK(expr)                      \
  do {                                        \
    int _e = (int)(expr);                     \
    if (_e != 0) return _e;                   \
  } while (0)

static inline int kamd_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- atomics ---------------------------------------------------------------
// fp32/fp64 global atomic add are single hardware instructions on gfx950
// (global_atomic_add_f32 / _f64); unsafeAtomicAdd selects them instead of a CAS
// loop.  The memory the shim hands us is ordinary coarse-grained device memory.
__device__ __forceinline__ void kamd_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void kamd_atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// fp16 add through a 32-bit CAS on the containing word (the reference uses
// THCAtomics' atomicAdd(at::Half*) which is the same construction).
__device__ __forceinline__ void kamd_atomic_add(__half* p, __half v) {
  uintptr_t a = (uintptr_t)p;
  unsigned int* w = (unsigned int*)(a & ~(uintptr_t)3);
  const bool hi = (a & 2) != 0;
  unsigned int old = *w, assumed;
  do {
    assumed = old;
    unsigned short cur = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
    __half h = __ushort_as_half(cur);
    // c10::Half semantics: float add, round to half
    __half r = __float2half(__half2float(h) + __half2float(v));
    unsigned short rb = __half_as_ushort(r);
    unsigned int nw = hi ? ((assumed & 0x0000ffffu) | ((unsigned int)rb << 16))
                         : ((assumed & 0xffff0000u) | (unsigned int)rb);
    old = atomicCAS(w, assumed, nw);
  } while (old != assumed);
}

// ---- c10::Half-style arithmetic -------------------------------------------
// The reference instantiates its kernels on at::Half, whose operators compute in
// float and round to half after EVERY operation (c10/util/Half-inl.h).  hround()
// reproduces that rounding step on a float carrier.
__device__ __forceinline__ float kamd_hround(float x) { return __half2float(__float2half(x)); }
