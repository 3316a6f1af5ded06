// Shared device/host helpers for libkaolin_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <stdlib.h>

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

// ---- persistent-kernel grids -------------------------------------------------------------------------------------
// Persistent kernels (static round-robin over a worklist, `for (i = blockIdx.x; i < n; i += gridDim.x)`) are launched with
// a fixed number of workgroups per CU, found by sweeps (12-32: see the launch sites).  Sizing the grid to exactly one
// resident set from the runtime's occupancy query was tried at the end of round 1 and measured at the start of round 2:
// slower every time (soft_search 133 -> 146 us) -- items differ in cost, and more, smaller static shares balance better.
// Measurement knobs (grids of the persistent kernels, thresholds, A/B switches of DESIGN.md's tables): an integer from the environment
// -- in EXPERIMENT builds only (`make -C kaolin_amd/csrc variant NAME=... DEFS="-DKAMD_EXPERIMENT ..."`).  The product build
// compiles every kamd_env_int(...) to its default: no getenv on a launch path, nothing a stray variable can change.
#ifdef KAMD_EXPERIMENT
static inline int kamd_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (v == nullptr || *v == 0) return dflt;
  const int x = atoi(v);
  return x > 0 ? x : dflt;
}
#else
static inline int kamd_env_int(const char*, int dflt) { return dflt; }
#endif
// The three switches the TESTS flip inside one process (they force a search path the sizes would not pick, so that both paths are
// compared with the oracle and with each other): read per call in every build -- KAMD_SIDED_DISTANCE=brute, KAMD_TRIANGLE_DISTANCE=
// brute|sweep (their call sites), KAMD_TS_THREADS (the sweep's workgroup size) through this.
static inline int kamd_test_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (v == nullptr || *v == 0) return dflt;
  const int x = atoi(v);
  return x > 0 ? x : dflt;
}

// ---- fills -------------------------------------------------------------------------------------------------------------
// hipMemsetAsync runs a generic fill kernel at ~2 TB/s on this stack, and as a memset NODE of a captured HIP graph it
// faulted on replay (tools/graph_step.py).  The workspaces cleared every call (tile bitmasks: ~50 MB per DIB-R call,
// the 256^3 voxel grid: 67 MB) are 16-byte aligned, so a grid-stride kernel of 16-byte stores (the K-buffer fill
// measured 5.7 TB/s) does the same job 2-3x faster; the < 16-byte edges of an unaligned range go through a one-wave
// byte kernel.  Nothing in the library calls hipMemsetAsync: every operator is a sequence of kernel nodes, capturable.
__global__ __launch_bounds__(256) static void kamd_fill16_kernel(uint4* __restrict__ p, size_t n16, unsigned int word) {
  const size_t stride = (size_t)gridDim.x * 256;
  const uint4 z = make_uint4(word, word, word, word);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p[i] = z;
}
__global__ __launch_bounds__(64) static void kamd_fill_edges_kernel(unsigned char* head, int nh, unsigned char* tail, int nt,
                                                                    unsigned char v) {
  if ((int)threadIdx.x < nh) head[threadIdx.x] = v;
  if ((int)threadIdx.x < nt) tail[threadIdx.x] = v;
}
// [ptr, ptr + bytes) <- the byte `byte`, any alignment and size
static inline int kamd_fill_async(void* ptr, size_t bytes, int byte, hipStream_t st) {
  if (bytes == 0) return 0;
  unsigned char* c = (unsigned char*)ptr;
  size_t head = ((uintptr_t)c & 15) ? 16 - ((uintptr_t)c & 15) : 0;
  if (head > bytes) head = bytes;
  const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
  if (n16) {
    size_t blocks = (n16 + 255) / 256;
    if (blocks > (size_t)KAMD_NUM_CU * 16) blocks = (size_t)KAMD_NUM_CU * 16;
    hipLaunchKernelGGL(kamd_fill16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint4*)(c + head), n16,
                       0x01010101u * (unsigned int)(byte & 0xFF));
  }
  if (head || tail)
    hipLaunchKernelGGL(kamd_fill_edges_kernel, dim3(1), dim3(64), 0, st, c, (int)head, c + head + n16 * 16, (int)tail,
                       (unsigned char)(byte & 0xFF));
  return (int)hipGetLastError();
}
static inline int kamd_zero_async(void* ptr, size_t bytes, hipStream_t st) { return kamd_fill_async(ptr, bytes, 0, st); }

// up to three ranges cleared by ONE launch (a launch costs ~5 us of stream time however little it does: the DIB-R forward
// clears its list heads, its work-list header and the gradient buffer its backward will accumulate into).  Ranges that are
// not whole 16-byte chunks fall back to a launch of their own.
__global__ __launch_bounds__(256) static void kamd_zero3_kernel(uint4* __restrict__ p0, size_t n0, uint4* __restrict__ p1, size_t n1,
                                                               uint4* __restrict__ p2, size_t n2) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  const size_t total = n0 + n1 + n2, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    if (i < n0)
      p0[i] = z;
    else if (i < n0 + n1)
      p1[i - n0] = z;
    else
      p2[i - n0 - n1] = z;
  }
}
static inline int kamd_zero3_async(void* a, size_t na, void* b, size_t nb, void* c, size_t nc, hipStream_t st) {
  void* p[3] = {a, b, c};
  size_t n[3] = {a ? na : 0, b ? nb : 0, c ? nc : 0};
  size_t n16[3] = {0, 0, 0};
  for (int i = 0; i < 3; ++i) {
    if (n[i] == 0) continue;
    if (((uintptr_t)p[i] & 15) == 0 && n[i] % 16 == 0) {
      n16[i] = n[i] / 16;
    } else {
      const int rc = kamd_zero_async(p[i], n[i], st);
      if (rc != 0) return rc;
    }
  }
  const size_t total = n16[0] + n16[1] + n16[2];
  if (total == 0) return 0;
  size_t blocks = (total + 255) / 256;
  if (blocks > (size_t)KAMD_NUM_CU * 16) blocks = (size_t)KAMD_NUM_CU * 16;
  hipLaunchKernelGGL(kamd_zero3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint4*)p[0], n16[0], (uint4*)p[1], n16[1],
                     (uint4*)p[2], n16[2]);
  return (int)hipGetLastError();
}
