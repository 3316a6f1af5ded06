// sided_distance forward / backward for MI355X (gfx950).
//
// Replaces kaolin/csrc/metrics/sided_distance_cuda.cu:52-242 (K5, K6) behind the
// C ABI in include/kaolin_amd.h.  Semantics kept from the reference:
//   dist[b,i] = min_j d(p1[b,i], p2[b,j]),  d = dx*dx + dy*dy + dz*dz with
//   dx = p2.x - p1.x (sided_distance_cuda.cu:83-86), idx = LOWEST j attaining
//   the minimum (strict '<' inside a tile :88,100, strict '>' across tiles :193),
//   a NaN distance to target 0 sticks (the `k == 0 ||` seed, :88), M == 0 leaves
//   the at::zeros outputs untouched (sided_distance.cpp:80-81).
// Arithmetic contract (pinned identically in oracle/kaolin_oracle.c so device and
// oracle agree bit-for-bit): fp32/fp64 evaluate d = fma(dz,dz, fma(dy,dy, dx*dx))
// -- the contraction nvcc's default -fmad=true applies to the reference source;
// fp16 follows c10::Half (float op, round to half after every operation).
//
// MI355X design (this is a VALU-bound all-pairs search, 8300 FLOP/B, see
// DESIGN.md): instead of the reference's fixed 32x16 grid with one query per
// thread and a global read-modify-write per 512-tile, the fp32 fast path
//   * keeps Q=4 queries per lane in registers so one broadcast ds_read_b128 of a
//     target feeds 4 distance evaluations (LDS pipe ~30% busy instead of >100%),
//   * splits the target set across blockIdx.y so >= 6 workgroups/CU are resident
//     for any N (100k queries alone are only 390 waves for 1024 SIMDs),
//   * tracks only the running MIN in the hot loop (v_min3_f32: 0.5 VALU/pair)
//     and the 16-target chunk in which it last improved (3 VALU per 16 pairs);
//     the exact lowest index is recovered afterwards by re-evaluating that one
//     chunk with the identical arithmetic (kernel sd_final_f32).
// That is 6.7 VALU/pair instead of 9 for the compare+2xselect formulation.
#include "common.h"
#include <stdlib.h>
#include "profile.h"
#include "sided_distance_grid.h"
#include "reseed.h"
#include <type_traits>
#include "../../include/kaolin_amd.h"

namespace {

// ---- per-dtype arithmetic ---------------------------------------------------
template <typename T> struct SdArith;
template <> struct SdArith<float> {
  using acc_t = float;
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float dist(float tx, float ty, float tz, float qx, float qy, float qz) {
    float dx = tx - qx, dy = ty - qy, dz = tz - qz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  }
  // 2 * (a - b) * g
  static __device__ __forceinline__ float grad(float a, float b, float g) { return 2.f * (a - b) * g; }
};
template <> struct SdArith<double> {
  using acc_t = double;
  static __device__ __forceinline__ double load(const double* p) { return *p; }
  static __device__ __forceinline__ void store(double* p, double v) { *p = v; }
  static __device__ __forceinline__ double dist(double tx, double ty, double tz, double qx, double qy, double qz) {
    double dx = tx - qx, dy = ty - qy, dz = tz - qz;
    return __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  }
  static __device__ __forceinline__ double grad(double a, double b, double g) { return 2. * (a - b) * g; }
};
template <> struct SdArith<__half> {
  using acc_t = float;  // half values carried in float, rounded after every op
  static __device__ __forceinline__ float load(const __half* p) { return __half2float(*p); }
  static __device__ __forceinline__ void store(__half* p, float v) { *p = __float2half(v); }
  static __device__ __forceinline__ float dist(float tx, float ty, float tz, float qx, float qy, float qz) {
    float dx = kamd_hround(tx - qx), dy = kamd_hround(ty - qy), dz = kamd_hround(tz - qz);
    float xx = kamd_hround(dx * dx), yy = kamd_hround(dy * dy), zz = kamd_hround(dz * dz);
    return kamd_hround(kamd_hround(xx + yy) + zz);
  }
  static __device__ __forceinline__ float grad(float a, float b, float g) {
    // 2 * (a - b) * g with c10::Half rounding: int*Half -> Half, Half*Half -> Half
    return kamd_hround(kamd_hround(2.f * kamd_hround(a - b)) * g);
  }
};

// Integer clouds (the reference dispatches Byte / Short / Int / Long too, kaolin/csrc/utils.h:50-64): its kernel spells
// every intermediate as scalar_t, so differences, squares and sums are computed with C's usual promotions and TRUNCATED
// to the element type at each assignment (sided_distance_cuda.cu:83-86,229-236) -- e.g. uint8 differences wrap mod 256.
template <typename I, typename W>   // W: the type C promotes I's arithmetic to (made unsigned where overflow would be UB)
struct SdIntArith {
  using acc_t = I;
  static __device__ __forceinline__ I load(const I* p) { return *p; }
  static __device__ __forceinline__ void store(I* p, I v) { *p = v; }
  static __device__ __forceinline__ I dist(I tx, I ty, I tz, I qx, I qy, I qz) {
    const I dx = (I)((W)tx - (W)qx), dy = (I)((W)ty - (W)qy), dz = (I)((W)tz - (W)qz);
    return (I)((W)dx * (W)dx + (W)dy * (W)dy + (W)dz * (W)dz);
  }
  static __device__ __forceinline__ I grad(I a, I b, I g) { return (I)((W)2 * ((W)a - (W)b) * (W)g); }
};
template <> struct SdArith<uint8_t> : SdIntArith<uint8_t, int> {};
template <> struct SdArith<int16_t> : SdIntArith<int16_t, int> {};
template <> struct SdArith<int32_t> : SdIntArith<int32_t, unsigned int> {};
template <> struct SdArith<int64_t> : SdIntArith<int64_t, unsigned long long> {};

// atomic adds for the integer gradients (sub-word types through a CAS on the containing 32-bit word)
using ::kamd_atomic_add;
__device__ __forceinline__ void kamd_atomic_add(int32_t* p, int32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ void kamd_atomic_add(int64_t* p, int64_t v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
template <typename I>
__device__ __forceinline__ void kamd_atomic_add_subword(I* p, I v) {
  const uintptr_t a = (uintptr_t)p;
  unsigned int* w = (unsigned int*)(a & ~(uintptr_t)3);
  const unsigned int shift = (unsigned int)(a & 3) * 8u, mask = (sizeof(I) == 1 ? 0xFFu : 0xFFFFu) << shift;
  unsigned int old = *w, assumed;
  do {
    assumed = old;
    const unsigned int cur = (assumed & mask) >> shift;
    const unsigned int sum = ((cur + (unsigned int)(typename std::make_unsigned<I>::type)v) << shift) & mask;
    old = atomicCAS(w, assumed, (assumed & ~mask) | sum);
  } while (old != assumed);
}
__device__ __forceinline__ void kamd_atomic_add(uint8_t* p, uint8_t v) { kamd_atomic_add_subword<uint8_t>(p, v); }
__device__ __forceinline__ void kamd_atomic_add(int16_t* p, int16_t v) { kamd_atomic_add_subword<int16_t>(p, v); }

// ---- generic forward: one query per lane, exact reference semantics ---------
constexpr int SDG_THREADS = 256;
constexpr int SDG_TILE = 1024;

template <typename T>
__global__ __launch_bounds__(SDG_THREADS) void sd_forward_generic(
    int N, int M, const T* __restrict__ p1, const T* __restrict__ p2,
    T* __restrict__ dist, int64_t* __restrict__ idx) {
  using A = SdArith<T>;
  using acc_t = typename A::acc_t;
  __shared__ acc_t tile[SDG_TILE * 3];
  const int b = blockIdx.y;
  const int i = blockIdx.x * SDG_THREADS + threadIdx.x;
  const T* P1 = p1 + (size_t)b * N * 3;
  const T* P2 = p2 + (size_t)b * M * 3;
  const bool active = i < N;
  acc_t qx = 0, qy = 0, qz = 0;
  if (active) {
    qx = A::load(P1 + (size_t)i * 3 + 0);
    qy = A::load(P1 + (size_t)i * 3 + 1);
    qz = A::load(P1 + (size_t)i * 3 + 2);
  }
  acc_t best = 0;
  int best_i = 0;
  for (int t0 = 0; t0 < M; t0 += SDG_TILE) {
    const int cnt = min(SDG_TILE, M - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt * 3; k += SDG_THREADS) tile[k] = A::load(P2 + (size_t)t0 * 3 + k);
    __syncthreads();
    if (active) {
      // the reference's loop exactly: tiles of 512 targets, each seeded by its first target whatever it yields (`k == 0 ||`,
      // sided_distance_cuda.cu:88) and merged with `k2 == 0 || result > best` (:193) -- a NaN at a tile's start hides the tile
      for (int kb = 0; kb < cnt; kb += kamd::SD_REF_TILE) {
        const int ke = min(cnt, kb + kamd::SD_REF_TILE);
        acc_t tb = A::dist(tile[kb * 3 + 0], tile[kb * 3 + 1], tile[kb * 3 + 2], qx, qy, qz);
        int tbi = t0 + kb;
        for (int k = kb + 1; k < ke; ++k) {
          acc_t d = A::dist(tile[k * 3 + 0], tile[k * 3 + 1], tile[k * 3 + 2], qx, qy, qz);
          if (d < tb) {
            tb = d;
            tbi = t0 + k;
          }
        }
        if ((t0 + kb) == 0 || best > tb) {
          best = tb;
          best_i = tbi;
        }
      }
    }
  }
  if (active && M > 0) {
    A::store(dist + (size_t)b * N + i, best);
    idx[(size_t)b * N + i] = best_i;
  }
}

// ---- fp32 fast path -----------------------------------------------------------
constexpr int SD_THREADS = 256;
constexpr int SD_Q = 4;        // queries per lane
constexpr int SD_TILE = 512;   // targets per LDS tile (xyz packed -> 6 KiB)
constexpr int SD_CHUNK = 16;   // targets per index chunk

__device__ __forceinline__ float sd_min3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__global__ __launch_bounds__(SD_THREADS) void sd_main_f32(
    int B, int N, int M, int Ms, const float* __restrict__ p1, const float* __restrict__ p2,
    float* __restrict__ part_d, int* __restrict__ part_c) {
  // Targets sit in LDS exactly as in HBM (xyz xyz ...): 4 targets = 3 float4, so a
  // 16-target chunk costs 12 broadcast ds_read_b128 (4 LDS cycles each) instead of
  // 16 ds_read_b96 (8 cycles each).
  __shared__ __attribute__((aligned(16))) float tile[SD_TILE * 3];
  const int b = blockIdx.z, s = blockIdx.y;
  const float* P1 = p1 + (size_t)b * N * 3;
  const float* P2 = p2 + (size_t)b * M * 3;
  const int q0 = blockIdx.x * (SD_THREADS * SD_Q) + threadIdx.x;

  float qx[SD_Q], qy[SD_Q], qz[SD_Q], best[SD_Q];
  int bc[SD_Q];
#pragma unroll
  for (int q = 0; q < SD_Q; ++q) {
    const int j = q0 + q * SD_THREADS;
    const int jj = j < N ? j : N - 1;
    qx[q] = P1[(size_t)jj * 3 + 0];
    qy[q] = P1[(size_t)jj * 3 + 1];
    qz[q] = P1[(size_t)jj * 3 + 2];
    best[q] = INFINITY;
    bc[q] = 0;
  }

  const int m0 = s * Ms;
  const int m1 = min(M, m0 + Ms);
  for (int t0 = m0; t0 < m1; t0 += SD_TILE) {
    const int cnt = min(SD_TILE, m1 - t0);
    __syncthreads();
    {
      const float* src = P2 + (size_t)t0 * 3;
#pragma unroll
      for (int k = threadIdx.x; k < SD_TILE * 3; k += SD_THREADS)
        tile[k] = k < cnt * 3 ? src[k] : INFINITY;  // padding never wins: d = inf or NaN
    }
    __syncthreads();
    const int nchunks = (cnt + SD_CHUNK - 1) / SD_CHUNK;
    const int cid0 = t0 / SD_CHUNK;
    const float4* tile4 = reinterpret_cast<const float4*>(tile);
    for (int c = 0; c < nchunks; ++c) {
      float old[SD_Q];
#pragma unroll
      for (int q = 0; q < SD_Q; ++q) old[q] = best[q];
#pragma unroll
      for (int g = 0; g < SD_CHUNK / 4; ++g) {
        const float4 a = tile4[(c * (SD_CHUNK / 4) + g) * 3 + 0];
        const float4 bb = tile4[(c * (SD_CHUNK / 4) + g) * 3 + 1];
        const float4 cc = tile4[(c * (SD_CHUNK / 4) + g) * 3 + 2];
#pragma unroll
        for (int q = 0; q < SD_Q; ++q) {
          const float d0 = SdArith<float>::dist(a.x, a.y, a.z, qx[q], qy[q], qz[q]);
          const float d1 = SdArith<float>::dist(a.w, bb.x, bb.y, qx[q], qy[q], qz[q]);
          const float d2 = SdArith<float>::dist(bb.z, bb.w, cc.x, qx[q], qy[q], qz[q]);
          const float d3 = SdArith<float>::dist(cc.y, cc.z, cc.w, qx[q], qy[q], qz[q]);
          best[q] = sd_min3(sd_min3(best[q], d0, d1), d2, d3);
        }
      }
#pragma unroll
      for (int q = 0; q < SD_Q; ++q) bc[q] = best[q] < old[q] ? (cid0 + c) : bc[q];
    }
  }
#pragma unroll
  for (int q = 0; q < SD_Q; ++q) {
    const int j = q0 + q * SD_THREADS;
    if (j < N) {
      const size_t o = ((size_t)s * B + b) * N + j;
      part_d[o] = best[q];
      part_c[o] = bc[q];
    }
  }
}

__global__ __launch_bounds__(256) void sd_final_f32(
    int B, int N, int M, int S, const float* __restrict__ p1, const float* __restrict__ p2,
    const float* __restrict__ part_d, const int* __restrict__ part_c,
    float* __restrict__ dist, int64_t* __restrict__ idx) {
  const size_t total = (size_t)B * N;
  const size_t gid_ = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = gid_ < total;
  const size_t gid = live ? gid_ : total - 1;
  const int b = (int)(gid / N);
  float best = part_d[gid];
  int c = part_c[gid];
  for (int s = 1; s < S; ++s) {
    const float d = part_d[(size_t)s * total + gid];
    if (d < best) {  // strict: the lower split (lower indices) keeps ties
      best = d;
      c = part_c[(size_t)s * total + gid];
    }
  }
  const float qx = p1[gid * 3 + 0], qy = p1[gid * 3 + 1], qz = p1[gid * 3 + 2];
  const float* T = p2 + (size_t)b * M * 3;
  const float d0 = SdArith<float>::dist(T[0], T[1], T[2], qx, qy, qz);
  int found;
  if (d0 != d0) {  // the reference's `k == 0 ||` seed: a NaN distance to target 0 sticks
    best = d0;
    found = 0;
  } else {
    const int k0 = c * SD_CHUNK;
    const int k1 = min(M, k0 + SD_CHUNK);
    found = k0;
    for (int k = k0; k < k1; ++k) {
      const float d = SdArith<float>::dist(T[(size_t)k * 3], T[(size_t)k * 3 + 1], T[(size_t)k * 3 + 2], qx, qy, qz);
      if (d == best) {
        found = k;
        break;
      }
    }
  }
  // the reference re-seeds at every tile of 512 targets: a winner inside a tile whose first target yields NaN is not its answer
  // (reseed.h; the test is one more gather per query whose winner is past the first tile)
  {
    auto dist_f = [](float tx, float ty, float tz, float x, float y, float z) { return SdArith<float>::dist(tx, ty, tz, x, y, z); };
    auto load_f = [](const float* p) { return *p; };
    const bool need = live && kamd::sd_winner_in_dead_tile<float>(T, found, qx, qy, qz, dist_f, load_f);
    if (__any(need)) kamd::sd_reseed_fix<float>(need, qx, qy, qz, T, M, best, found, dist_f, load_f);
  }
  if (live) {
    dist[gid] = best;
    idx[gid] = found;
  }
}

// split plan shared by the workspace query and the launcher
struct SdPlan {
  bool fast;
  int nx, S, Ms;
};
inline SdPlan sd_plan(int B, int N, int M) {
  SdPlan p;
  p.fast = (long long)N * M >= (1ll << 22) && N >= 1024 && M >= 2 * SD_TILE;
  p.nx = kamd_cdiv(N, SD_THREADS * SD_Q);
  const int ntiles = kamd_cdiv(M, SD_TILE);
  long long want = (long long)KAMD_NUM_CU * 6;  // ~6 resident workgroups per CU
  int S = (int)((want + (long long)p.nx * B - 1) / ((long long)p.nx * B));
  if (S < 1) S = 1;
  if (S > ntiles) S = ntiles;
  int tiles_per = kamd_cdiv(ntiles, S);
  p.Ms = tiles_per * SD_TILE;
  p.S = kamd_cdiv(M, p.Ms);
  return p;
}

template <typename T>
int sd_forward_generic_launch(hipStream_t st, int B, int N, int M, const T* p1, const T* p2, T* dist, int64_t* idx) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;  // M == 0: outputs keep the caller's zeros
  kamd::ProfScope prof_(kamd::K_SD_GENERIC, st);
  for (int b0 = 0; b0 < B; b0 += 65535) {  // the batch rides on the grid's y extent: slabs of at most 65535 items
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    dim3 grid(kamd_cdiv(N, SDG_THREADS), nb);
    hipLaunchKernelGGL(sd_forward_generic<T>, grid, dim3(SDG_THREADS), 0, st, N, M, p1 + (size_t)b0 * N * 3,
                       p2 + (size_t)b0 * M * 3, dist + (size_t)b0 * N, idx + (size_t)b0 * N);
  }
  KAMD_RETURN_LAST_ERROR();
}

// ---- backward (K6) ------------------------------------------------------------
// One thread per (point, coordinate): consecutive lanes read and write consecutive scalars of p1 / g1, and the three atomics a
// point sends to its nearest target sit in three consecutive lanes of ONE instruction -- one request per point (global float
// atomics cost per request, ~60 ps chip-wide on MI355X: a thread per point adding x, y, z in turn made three).
template <typename T>
__global__ __launch_bounds__(256) void sd_backward(
    int N, int M, const T* __restrict__ grad, const T* __restrict__ p1, const T* __restrict__ p2,
    const int64_t* __restrict__ idx, T* __restrict__ g1, T* __restrict__ g2) {
  using A = SdArith<T>;
  using acc_t = typename A::acc_t;
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // scalar index into the item's (N, 3) arrays
  if (e >= (long long)N * 3) return;
  const int i = (int)(e / 3), a = (int)(e - (long long)i * 3);
  const size_t main_id = (size_t)b * N + i;
  const acc_t c1 = A::load(p1 + main_id * 3 + a);
  const size_t t = ((size_t)idx[main_id] + (size_t)b * M) * 3;
  const acc_t c2 = A::load(p2 + t + a);
  const acc_t g = A::load(grad + main_id);
  A::store(g1 + main_id * 3 + a, A::grad(c1, c2, g));
  T r;
  A::store(&r, A::grad(c2, c1, g));
  kamd_atomic_add(g2 + t + a, r);
}

template <typename T>
int sd_backward_launch(hipStream_t st, int B, int N, int M, const T* grad, const T* p1, const T* p2,
                       const int64_t* idx, T* g1, T* g2) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  kamd::ProfScope prof_(kamd::K_SD_BACKWARD, st);
  for (int b0 = 0; b0 < B; b0 += 65535) {
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    dim3 grid(kamd_cdiv((long long)N * 3, 256), nb);
    hipLaunchKernelGGL(sd_backward<T>, grid, dim3(256), 0, st, N, M, grad + (size_t)b0 * N, p1 + (size_t)b0 * N * 3,
                       p2 + (size_t)b0 * M * 3, idx + (size_t)b0 * N, g1 + (size_t)b0 * N * 3, g2 + (size_t)b0 * M * 3);
  }
  KAMD_RETURN_LAST_ERROR();
}

// ---- chamfer backward: both directions, both clouds ------------------------------------------------------------------
// chamfer_distance = w1 * mean_i f(dist1_i) + w2 * mean_j f(dist2_j), f = identity or sqrt (kaolin/metrics/pointcloud.py
// :120-136).  autograd of that composition hands every point of direction d the upstream value
// grad[b] * w_d * (1 / n_d) [/ (2 * sqrt(dist))], then runs the sided_distance backward of each direction
// (sided_distance_cuda.cu:203-242) and adds the two results per cloud.  One thread per point of either cloud does all of
// it in two launches: SCATTER = false stores the point's own term 2 * (q - t) * g (this initialises g1 / g2, no fill
// needed), SCATTER = true then adds -that to the nearest point of the other cloud with float atomics.
template <bool SCATTER>
__global__ __launch_bounds__(256) void sd_chamfer_backward(int N, int M, const float* __restrict__ grad, float w1, float w2,
                                                           float inv_n, float inv_m, int squared,
                                                           const float* __restrict__ p1, const float* __restrict__ p2,
                                                           const int64_t* __restrict__ idx1,
                                                           const int64_t* __restrict__ idx2,
                                                           const float* __restrict__ dist1,
                                                           const float* __restrict__ dist2, float* __restrict__ g1,
                                                           float* __restrict__ g2) {
  // (one thread per (point, coordinate), as sd_backward: coalesced scalars, one atomic request per point)
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= ((long long)N + M) * 3) return;
  int i = (int)(e / 3);
  const int a = (int)(e - (long long)i * 3);
  const bool fwd = i < N;
  if (!fwd) i -= N;
  const int nq = fwd ? N : M, nt = fwd ? M : N;
  const float* Q = (fwd ? p1 : p2) + ((size_t)b * nq + i) * 3;
  const size_t t = (size_t)(fwd ? idx1 : idx2)[(size_t)b * nq + i] + (size_t)b * nt;
  const float* T = (fwd ? p2 : p1) + t * 3;
  float g = grad[b];
  const float w = fwd ? w1 : w2;
  if (w != 1.f) g = g * w;
  g = g * (fwd ? inv_n : inv_m);
  if (!squared) g = g / (2.f * sqrtf((fwd ? dist1 : dist2)[(size_t)b * nq + i]));
  float* GQ = (fwd ? g1 : g2) + ((size_t)b * nq + i) * 3;
  float* GT = (fwd ? g2 : g1) + t * 3;
  const float q = Q[a], x = T[a];
  if (SCATTER)
    kamd_atomic_add(GT + a, 2.f * (x - q) * g);
  else
    GQ[a] = 2.f * (q - x) * g;
}

// KAMD_SIDED_DISTANCE=brute keeps the all-pairs kernels for every size (A/B timing, tests of both paths)
inline bool sd_force_brute() {
  const char* e = getenv("KAMD_SIDED_DISTANCE");
  return e != nullptr && e[0] == 'b';
}

}  // namespace

extern "C" {

size_t kamd_sided_distance_forward_workspace(int B, int N, int M, int elem_size) {
  if ((elem_size != 2 && elem_size != 4 && elem_size != 8) || B <= 0 || N <= 0 || M <= 0) return 0;
  if (kamd::sdgrid_applicable(B, N, M) && !sd_force_brute()) return kamd::sdgrid_workspace_bytes(B, N, M, elem_size);
  if (elem_size != 4) return 0;
  SdPlan p = sd_plan(B, N, M);
  if (!p.fast) return 0;
  return (size_t)p.S * B * N * (sizeof(float) + sizeof(int));
}

int kamd_sided_distance_forward_f32(void* stream, int B, int N, int M, const float* p1, const float* p2,
                                    float* dist, int64_t* idx, void* workspace) {
  hipStream_t st = (hipStream_t)stream;
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  if (workspace != nullptr && kamd::sdgrid_applicable(B, N, M) && !sd_force_brute())
    return kamd::sdgrid_forward_f32(st, B, N, M, p1, p2, dist, idx, workspace);
  SdPlan p = sd_plan(B, N, M);
  if (!p.fast || workspace == nullptr) return sd_forward_generic_launch<float>(st, B, N, M, p1, p2, dist, idx);
  float* part_d = (float*)workspace;
  int* part_c = (int*)(part_d + (size_t)p.S * B * N);
  {
    kamd::ProfScope prof_(kamd::K_SD_MAIN, st);
    hipLaunchKernelGGL(sd_main_f32, dim3(p.nx, p.S, B), dim3(SD_THREADS), 0, st, B, N, M, p.Ms, p1, p2, part_d, part_c);
  }
  KAMD_CHECK(hipGetLastError());
  {
    kamd::ProfScope prof_(kamd::K_SD_FINAL, st);
    hipLaunchKernelGGL(sd_final_f32, dim3(kamd_cdiv((long long)B * N, 256)), dim3(256), 0, st, B, N, M, p.S, p1, p2,
                     part_d, part_c, dist, idx);
  }
  KAMD_RETURN_LAST_ERROR();
}

size_t kamd_sided_distance_pair_forward_workspace(int B, int N, int M, int elem_size) {
  if ((elem_size != 4 && elem_size != 8) || !kamd::sdgrid_pair_applicable(B, N, M) || sd_force_brute()) return 0;
  return kamd::sdgrid_pair_workspace_bytes(B, N, M, elem_size);
}
int kamd_sided_distance_pair_forward_f64(void* stream, int B, int N, int M, const double* p1, const double* p2,
                                         double* dist1, int64_t* idx1, double* dist2, int64_t* idx2, void* workspace) {
  if (workspace == nullptr || !kamd::sdgrid_pair_applicable(B, N, M)) return (int)hipErrorInvalidValue;
  return kamd::sdgrid_pair_forward_f64((hipStream_t)stream, B, N, M, p1, p2, dist1, idx1, dist2, idx2, workspace);
}

int kamd_sided_distance_pair_forward_f32(void* stream, int B, int N, int M, const float* p1, const float* p2,
                                         float* dist1, int64_t* idx1, float* dist2, int64_t* idx2, void* workspace) {
  if (workspace == nullptr || !kamd::sdgrid_pair_applicable(B, N, M)) return (int)hipErrorInvalidValue;
  return kamd::sdgrid_pair_forward_f32((hipStream_t)stream, B, N, M, p1, p2, dist1, idx1, dist2, idx2, workspace);
}

int kamd_chamfer_distance_backward_f32(void* stream, int B, int N, int M, const float* grad, float w1, float w2,
                                       int squared, const float* p1, const float* p2, const int64_t* idx1,
                                       const int64_t* idx2, const float* dist1, const float* dist2, float* g1, float* g2) {
  hipStream_t st = (hipStream_t)stream;
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  {
    kamd::ProfScope prof_(kamd::K_SD_BACKWARD, st);
    const dim3 grid(kamd_cdiv(((long long)N + M) * 3, 256), B);  // (a thread per (point, coordinate))
    hipLaunchKernelGGL(sd_chamfer_backward<false>, grid, dim3(256), 0, st, N, M, grad, w1, w2, 1.f / (float)N,
                       1.f / (float)M, squared, p1, p2, idx1, idx2, dist1, dist2, g1, g2);
    hipLaunchKernelGGL(sd_chamfer_backward<true>, grid, dim3(256), 0, st, N, M, grad, w1, w2, 1.f / (float)N,
                       1.f / (float)M, squared, p1, p2, idx1, idx2, dist1, dist2, g1, g2);
  }
  KAMD_RETURN_LAST_ERROR();
}

size_t kamd_chamfer_distance_forward_workspace(int B, int N, int M, int with_grad) {
  if (B <= 0 || N <= 0 || M <= 0 || !kamd::sdgrid_pair_applicable(B, N, M) || sd_force_brute()) return 0;
  return kamd::sdgrid_chamfer_workspace_bytes(B, N, M, with_grad != 0);
}

int kamd_chamfer_distance_forward_f32(void* stream, int B, int N, int M, const float* p1, const float* p2, float w1,
                                      float w2, int squared, int with_grad, float* out, float* dist1, int64_t* idx1,
                                      float* dist2, int64_t* idx2, void* workspace) {
  if (workspace == nullptr || out == nullptr || !kamd::sdgrid_pair_applicable(B, N, M)) return (int)hipErrorInvalidValue;
  return kamd::sdgrid_chamfer_forward_f32((hipStream_t)stream, B, N, M, p1, p2, w1, w2, squared, with_grad != 0, out, dist1,
                                          idx1, dist2, idx2, workspace);
}

int kamd_chamfer_distance_backward_fused_f32(void* stream, int B, int N, int M, const float* grad, void* workspace,
                                             float* g1, float* g2) {
  if (workspace == nullptr || !kamd::sdgrid_pair_applicable(B, N, M)) return (int)hipErrorInvalidValue;
  return kamd::sdgrid_chamfer_backward_f32((hipStream_t)stream, B, N, M, grad, workspace, g1, g2);
}

int kamd_sided_distance_forward_f64(void* stream, int B, int N, int M, const double* p1, const double* p2,
                                    double* dist, int64_t* idx, void* workspace) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  // large clouds: the exact grid search (the grid itself lives in float, distances are the reference's double expression)
  if (workspace != nullptr && kamd::sdgrid_applicable(B, N, M) && !sd_force_brute())
    return kamd::sdgrid_forward_f64((hipStream_t)stream, B, N, M, p1, p2, dist, idx, workspace);
  return sd_forward_generic_launch<double>((hipStream_t)stream, B, N, M, p1, p2, dist, idx);
}

int kamd_sided_distance_forward_f16(void* stream, int B, int N, int M, const uint16_t* p1, const uint16_t* p2,
                                    uint16_t* dist, int64_t* idx, void* workspace) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  // large clouds: the exact grid search with c10::Half's arithmetic (the reference dispatches half on the same kernel as float)
  if (workspace != nullptr && kamd::sdgrid_applicable(B, N, M) && !sd_force_brute())
    return kamd::sdgrid_forward_f16((hipStream_t)stream, B, N, M, p1, p2, dist, idx, workspace);
  return sd_forward_generic_launch<__half>((hipStream_t)stream, B, N, M, (const __half*)p1, (const __half*)p2,
                                           (__half*)dist, idx);
}

#define KAMD_SD_INT_ENTRY(SFX, I)                                                                                       \
  int kamd_sided_distance_forward_##SFX(void* stream, int B, int N, int M, const I* p1, const I* p2, I* dist,          \
                                        int64_t* idx, void* workspace) {                                              \
    (void)workspace;                                                                                                   \
    return sd_forward_generic_launch<I>((hipStream_t)stream, B, N, M, p1, p2, dist, idx);                              \
  }                                                                                                                    \
  int kamd_sided_distance_backward_##SFX(void* stream, int B, int N, int M, const I* grad, const I* p1, const I* p2,  \
                                         const int64_t* idx, I* g1, I* g2) {                                           \
    return sd_backward_launch<I>((hipStream_t)stream, B, N, M, grad, p1, p2, idx, g1, g2);                             \
  }
KAMD_SD_INT_ENTRY(u8, uint8_t)
KAMD_SD_INT_ENTRY(i16, int16_t)
KAMD_SD_INT_ENTRY(i32, int32_t)
KAMD_SD_INT_ENTRY(i64, int64_t)
#undef KAMD_SD_INT_ENTRY

int kamd_sided_distance_backward_f32(void* stream, int B, int N, int M, const float* grad, const float* p1,
                                     const float* p2, const int64_t* idx, float* g1, float* g2) {
  return sd_backward_launch<float>((hipStream_t)stream, B, N, M, grad, p1, p2, idx, g1, g2);
}
int kamd_sided_distance_backward_f64(void* stream, int B, int N, int M, const double* grad, const double* p1,
                                     const double* p2, const int64_t* idx, double* g1, double* g2) {
  return sd_backward_launch<double>((hipStream_t)stream, B, N, M, grad, p1, p2, idx, g1, g2);
}
int kamd_sided_distance_backward_f16(void* stream, int B, int N, int M, const uint16_t* grad, const uint16_t* p1,
                                     const uint16_t* p2, const int64_t* idx, uint16_t* g1, uint16_t* g2) {
  return sd_backward_launch<__half>((hipStream_t)stream, B, N, M, (const __half*)grad, (const __half*)p1,
                                    (const __half*)p2, idx, (__half*)g1, (__half*)g2);
}

}  // extern "C"
