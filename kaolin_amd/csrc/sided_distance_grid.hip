// sided_distance forward, exact uniform-grid search (fp32 and fp64) for MI355X (gfx950).
//
// The reference (kaolin/csrc/metrics/sided_distance_cuda.cu:52-201) is an all-pairs search: N*M distance
// evaluations against 24*(N+M) bytes of input, 8 300 FLOP/B at 100k x 100k -- VALU-bound by construction
// (sd_main_f32 already issues at 97 % of the measured FMA peak).  The only way to go faster is to evaluate fewer
// pairs while returning the SAME answer: dist = min_j d(p1_i, p2_j) with the identical fp32 expression
// d = fma(dz,dz, fma(dy,dy, dx*dx)), idx = the lowest j attaining it, and a NaN distance to target 0 sticking
// (the reference's `k == 0 ||` seed).  A uniform grid over the targets does that:
//   sdg_build (ONE persistent launch, phases separated by grid barriers):
//     1. bounding box of the finite targets (per batch item): workgroup minima / maxima merged with integer atomicMax on
//        an order-preserving encoding, so that consumers read six words;
//     2. cell id of every point and its rank inside the cell (the value the counting atomicAdd returns);
//     3. exclusive scan of the cell counts (per-block totals, then every 512-cell block adds up the totals before it);
//     4. counting sort without further atomics: point -> start[cell] + rank, stored as float4 {x, y, z, original index};
//   sdg_query  per query: seed with target 0 exactly as the reference does, then visit the cube of cells around
//                    the query ring by ring; after each ring every unvisited target is provably farther than the
//                    distance from the query to the cube's faces (minus a rounding margin), so the search stops as soon
//                    as the best distance is below that bound.  Ties are resolved towards the lower original index
//                    explicitly, so the arbitrary order inside a cell does not matter.
// Queries are processed in cell order too (the 64 queries of a wavefront are spatial neighbours: same cells, same
// cache lines).  One direction (sided_distance) bins the queries on the targets' grid for that.  Both directions
// (chamfer_distance: sided_distance(p1, p2) and sided_distance(p2, p1)) bin EACH cloud ONCE on its own grid: the sorted
// copy is the target list of one direction and the spatially coherent query list of the other, and every kernel of
// the pipeline is launched once for both clouds -- 6 launches instead of 14.
// Work per query is O(points in a few cells) instead of O(M); the result is bit-identical to the brute-force kernels
// (tests/test_sided_distance.py compares both with the oracle, incl. duplicates, NaNs, queries outside the box,
// degenerate boxes).  Non-finite targets are binned at a clamped cell: they yield NaN/inf distances that can never win
// against a finite one, as in the reference.
// fp64 clouds (the reference dispatches half / float / double: sided_distance_cuda.cu:252) use the same pipeline: the GRID is
// only an acceleration structure, so points are binned by their coordinates rounded to float (a monotone map: build and query
// agree on every cell), the sorted copy keeps the doubles, every distance is the reference's double expression, and the
// geometric bound's head-room (67 float ulps) covers the rounding of the coordinates it is computed from.  A query whose
// coordinates do not fit a float walks all targets (exact, slow, rare).
#include "common.h"
#include "profile.h"
#include "sided_distance_grid.h"
#include "grid_common.h"
#include "reseed.h"

namespace kamd {
namespace {

inline int sdg_cells_per_axis(int M) {
  // ~2 targets per cell on average for a volume-filling cloud; surfaces leave most cells empty, which is fine
  int G = (int)floor(cbrt((double)M / 2.0) + 0.5);
  if (G < 1) G = 1;
  if (G > SDG_MAXG) G = SDG_MAXG;
  return G;
}
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T> struct Vec4Of;
template <> struct Vec4Of<float> { typedef float4 type; };
template <> struct Vec4Of<double> { typedef double4 type; };
// the original index travels in the fourth component of a sorted point
__device__ __forceinline__ float sdg_pack_idx(int i, float) { return __int_as_float(i); }
__device__ __forceinline__ double sdg_pack_idx(int i, double) { return __hiloint2double(0, i); }
__device__ __forceinline__ int sdg_unpack_idx(float w) { return __float_as_int(w); }
__device__ __forceinline__ int sdg_unpack_idx(double w) { return __double2loint(w); }

// one cloud of the pipeline (kernel argument, by value)
template <typename T>
struct CloudT {
  typedef typename Vec4Of<T>::type V4;
  int n, G;             // points per batch item; cells per axis of the grid it is binned on
  const T* pts;         // (B, n, 3)
  unsigned int* box;    // (B, 8 words, SDG_BOX_STRIDE words apart) encoded {lo[3], hi[3]} of the grid's box (floats), 0 = no finite point yet, [6] = flag
  int* count;           // (B, G^3) zero before the build
  int* start;           // (B, G^3 + 1)
  int2* cellrank;       // (B, n) {cell, rank inside the cell}
  V4* sorted;           // (B, n) {x, y, z, original index} in cell order
};
typedef CloudT<float> Cloud;
template <typename T>
inline CloudT<T> cloud_as(const Cloud& c) {  // (the layout below is laid out by element size; the pointers are retyped)
  CloudT<T> r;
  r.n = c.n;
  r.G = c.G;
  r.pts = reinterpret_cast<const T*>(c.pts);
  r.box = c.box;
  r.count = c.count;
  r.start = c.start;
  r.cellrank = c.cellrank;
  r.sorted = reinterpret_cast<typename Vec4Of<T>::type*>(c.sorted);
  return r;
}

// what chamfer_distance adds to the search (SDG_VALUE / SDG_GRAD): everything the value and the gradient need is
// produced by the query launch itself
struct Fuse {
  double* sums;         // (2, B, query workgroups per item and direction) partial sums of f(dist); no init needed
  float* own_a;         // (B, N, 3) d value / d p1_i through p1_i's own nearest-point term (plain stores)
  float* own_b;         // (B, M, 3)
  float* scat_a;        // (B, N, 3) ... through the terms of the p2 points whose nearest point is p1_i (atomics); zero before
  float* scat_b;        // (B, M, 3)
  float w1, w2, c1, c2; // weights; w1 / N and w2 / M
  int squared;
};
enum { SDG_PLAIN = 0, SDG_VALUE = 1, SDG_GRAD = 2 };

struct SdgWs {
  Cloud a, b;           // a = p1 (N points), b = p2 (M points)
  int* scan_sums;       // (2, B, scan blocks): per-block totals, large grids only
  int scan_blocks;
  unsigned int* barrier;  // grid-barrier arrival counter of the build kernel; zero before it
  Fuse fuse;
  size_t zero_bytes;    // prefix of the workspace that must be zeroed (counts, boxes, barrier, sums, scatter sides)
  size_t total;
};
#ifndef KAMD_SDG_BUILD_THREADS
#define KAMD_SDG_BUILD_THREADS 512  // build knob for experiments
#endif
constexpr int SDG_BUILD_THREADS = KAMD_SDG_BUILD_THREADS;  // workgroup of the build kernel = cells per scan block
#ifndef KAMD_SDG_BUILD_UNROLL
#define KAMD_SDG_BUILD_UNROLL 1  // points a build thread bins / scatters at a time (4 / 8: their round trips in flight together -- no faster, profiles/r06r_*)
#endif
constexpr int SDG_BUILD_UNROLL = KAMD_SDG_BUILD_UNROLL;
constexpr int SDG_BAR_GROUPS = 8;   // groups of the grid barrier's arrivals (sdg_grid_barrier)
constexpr int SDG_BOX_STRIDE = 32;  // words between the words of a box record: every word on a 128-byte line of its own (phase 1 merges the
                                    // workgroups' extents with atomicMax, and atomics on one line are performed one at a time: 64 workgroups x 6
                                    // words on ONE line were ~15 us of the build)
// query workgroups of a chamfer launch at most (each leaves one partial sum)
inline size_t sdg_partials(int B) { return (size_t)2 * B > 8192 ? (size_t)2 * B : 8192; }

// pair = false: sided_distance(p1, p2): both clouds on p2's grid.  pair = true: each cloud on its own grid.
inline SdgWs sdg_layout(void* base, int B, int N, int M, const void* p1, const void* p2, bool pair, int mode, int esz = 4) {
  SdgWs w;
  w.b.n = M;
  w.b.G = sdg_cells_per_axis(M);
  w.b.pts = (const float*)p2;
  w.a.n = N;
  w.a.G = pair ? sdg_cells_per_axis(N) : w.b.G;
  w.a.pts = (const float*)p1;
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p + off;
    off += al(bytes);
    return r;
  };
  const size_t nca = (size_t)w.a.G * w.a.G * w.a.G, ncb = (size_t)w.b.G * w.b.G * w.b.G;
  w.b.count = (int*)take((size_t)B * ncb * 4);
  w.a.count = (int*)take((size_t)B * nca * 4);
  w.b.box = (unsigned int*)take((size_t)B * 8 * SDG_BOX_STRIDE * 4);
  w.a.box = pair ? (unsigned int*)take((size_t)B * 8 * SDG_BOX_STRIDE * 4) : w.b.box;
  w.barrier = (unsigned int*)take((size_t)(1 + SDG_BAR_GROUPS) * 128);
  w.fuse = Fuse{};
  if (mode >= SDG_GRAD) {
    w.fuse.scat_a = (float*)take((size_t)B * N * 12);
    w.fuse.scat_b = (float*)take((size_t)B * M * 12);
  }
  w.zero_bytes = off;
  if (mode >= SDG_VALUE) w.fuse.sums = (double*)take(sdg_partials(B) * 8);
  if (mode >= SDG_GRAD) {
    w.fuse.own_a = (float*)take((size_t)B * N * 12);
    w.fuse.own_b = (float*)take((size_t)B * M * 12);
  }
  w.b.start = (int*)take((size_t)B * (ncb + 1) * 4);
  w.a.start = (int*)take((size_t)B * (nca + 1) * 4);
  w.b.cellrank = (int2*)take((size_t)B * M * 8);
  w.a.cellrank = (int2*)take((size_t)B * N * 8);
  w.b.sorted = (float4*)take((size_t)B * M * 4 * esz);
  w.a.sorted = (float4*)take((size_t)B * N * 4 * esz);
  w.scan_blocks = (int)(((nca > ncb ? nca : ncb) + SDG_BUILD_THREADS - 1) / SDG_BUILD_THREADS);
  w.scan_sums = (int*)take((size_t)2 * B * w.scan_blocks * 4);
  w.total = off;
  return w;
}

// ---- the build: ONE persistent launch -----------------------------------------------------------------------------------
// bounding box -> cell id + rank -> scan of the cell counts -> counting-sort scatter are four dependent passes over at most
// a few hundred thousand points: as four launches they cost more in launch gaps and host time than in work (36 us for
// ~10 us of memory traffic at 100k + 100k points).  They are phases of one kernel whose workgroups are all resident
// (grid <= the CU count, 512 threads each: a quarter of a CU's wave slots, so several such kernels can share the GPU
// without starving each other) and meet at a grid barrier between phases: every wave waits for its own memory operations,
// thread 0 of each workgroup arrives on one counter and spins until all have.

// order-preserving float -> uint (negative values reversed below the positives); atomicMax on it is a float max, on
// its complement a float min, and the all-zero word the workspace fill leaves is below every encoded value
__device__ __forceinline__ unsigned int sdg_ord(float v) {
  const unsigned int u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float sdg_unord(unsigned int o) {
  return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// Data that crosses workgroups between phases (box words, cell counts, cell starts, scan totals) is written and read with
// agent-scope relaxed atomics: on gfx950 those are write-through / cache-bypassing accesses (sc1), coherent across the
// XCDs' L2s without any L2 write-back or invalidate.  A release / acquire fence at agent scope instead costs an L2
// write-back per workgroup per barrier: the first version of this kernel, with fences, took 214 us for ~10 us of work.
template <typename V>
__device__ __forceinline__ V sdg_ld(const V* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename V>
__device__ __forceinline__ void sdg_st(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Arrivals are counted in two levels: atomics on ONE line are performed one at a time (~40 ns each, memory-side), so 128 workgroups
// arriving on one counter spent ~5 us per barrier doing just that.  A workgroup arrives on the counter of its group (id % 8, a
// 128-byte line each); the group's last arrival -- the one whose returned count completes the group for this phase -- arrives on
// the top counter, which everybody polls.
__device__ __forceinline__ void sdg_grid_barrier(unsigned int* bar, unsigned int phase /* 1, 2, ... */, int nwg, int naps) {
  __builtin_amdgcn_s_waitcnt(0);  // this wave's stores and atomics have been performed
  __syncthreads();
  if (threadIdx.x == 0) {
    const int g = blockIdx.x % SDG_BAR_GROUPS;
    const unsigned int members = (unsigned int)((nwg - g + SDG_BAR_GROUPS - 1) / SDG_BAR_GROUPS);
    const unsigned int groups = (unsigned int)(nwg < SDG_BAR_GROUPS ? nwg : SDG_BAR_GROUPS);
    const unsigned int before = __hip_atomic_fetch_add(bar + 32 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1u == members * phase) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every poll is a coherent load of ONE address: hundreds of workgroups polling back to back queue up on its memory
    // channel, in front of the arrivals they are waiting for -- nap between polls
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups * phase)
      for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
}

// the grid geometry every consumer derives from the six words (same rules as grid_common.h's sdg_box)
template <bool COHERENT = false>
__device__ __forceinline__ Box sdg_box_decode(const unsigned int* w, int G) {
  Box bx;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const unsigned int ulo = COHERENT ? sdg_ld(w + a * SDG_BOX_STRIDE) : w[a * SDG_BOX_STRIDE];
    const unsigned int uhi = COHERENT ? sdg_ld(w + (3 + a) * SDG_BOX_STRIDE) : w[(3 + a) * SDG_BOX_STRIDE];
    float lo = 0.f, hi = 0.f;  // no finite point on this axis
    if (ulo != 0u && uhi != 0u) {
      lo = sdg_unord(~ulo);
      hi = sdg_unord(uhi);
    }
    float size = (hi - lo) / (float)G;
    if (!(size > 0.f) || !isfinite(size)) size = 1.f;  // degenerate extent: a single slab holds everything
    bx.lo[a] = lo;
    bx.size[a] = size;
    bx.inv[a] = 1.f / size;
  }
  return bx;
}

// X = the targets' cloud, Y = the other one; own_box_y: Y is binned on its own box (chamfer), else on X's (sided_distance)
template <typename T>
#ifdef KAMD_SDG_PHASE_TIMES  // development: workgroup 0 prints how long each phase of the build and each wait at a barrier took (100 MHz clock)
#define KAMD_SDG_T(i) const unsigned long long sdg_t##i = wall_clock64();
#define KAMD_SDG_T_PRINT()                                                                                                     \
  if (blockIdx.x == 0 && threadIdx.x == 0)                                                                                     \
    printf("sdg_build wg 0 (us): bbox %.2f wait %.2f | rank %.2f wait %.2f | scan a %.2f wait %.2f | scan b %.2f wait %.2f | scatter %.2f\n", \
           (sdg_t1 - sdg_t0) * 0.01, (sdg_t2 - sdg_t1) * 0.01, (sdg_t3 - sdg_t2) * 0.01, (sdg_t4 - sdg_t3) * 0.01, (sdg_t5 - sdg_t4) * 0.01, \
           (sdg_t6 - sdg_t5) * 0.01, (sdg_t7 - sdg_t6) * 0.01, (sdg_t8 - sdg_t7) * 0.01, (sdg_t9 - sdg_t8) * 0.01);
#else
#define KAMD_SDG_T(i)
#define KAMD_SDG_T_PRINT()
#endif
__global__ __launch_bounds__(SDG_BUILD_THREADS) void sdg_build(CloudT<T> X, CloudT<T> Y, int B, int own_box_y, int* sums, int scan_blocks,
                                                               unsigned int* barrier, int naps) {
  __shared__ float s_red[6][SDG_BUILD_THREADS / 64];
  __shared__ int s_wave[SDG_BUILD_THREADS / 64];
  __shared__ int s_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = gridDim.x, wg = blockIdx.x;
  unsigned int phase = 0;
  KAMD_SDG_T(0)

  // ---- phase 1: bounding boxes.  Units = (cloud, batch item); a unit is shared by P workgroups
  {
    const int U = (own_box_y ? 2 : 1) * B;
    const int P = nwg / U > 0 ? nwg / U : 1, step = nwg / P;
    for (int u = wg / P; u < U; u += step) {
      const int part = wg % P;
      const bool first = u < B;
      const int b = first ? u : u - B;
      const int n = first ? X.n : Y.n;
      const T* Pt = (first ? X.pts : Y.pts) + (size_t)b * n * 3;
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      bool odd_first = false;  // a non-finite coordinate at a point index 512 k: as a TARGET it can hide its tile (reseed.h)
      for (int i = part * SDG_BUILD_THREADS + tid; i < n; i += P * SDG_BUILD_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const T vt = Pt[(size_t)i * 3 + a];
          const float v = (float)vt;  // (the grid lives in float, whatever the points' type)
          if (isfinite(v)) {
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
          }
          odd_first = odd_first || (i >= SD_REF_TILE && (i & (SD_REF_TILE - 1)) == 0 && !(fabs((double)vt) < (double)INFINITY));
        }
      }
      if (odd_first) atomicMax((first ? X.box : Y.box) + ((size_t)b * 8 + 6) * SDG_BOX_STRIDE, 1u);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
          hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
        }
      }
      __syncthreads();  // s_red of the previous unit has been read
      if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          s_red[a][wave] = lo[a];
          s_red[3 + a][wave] = hi[a];
        }
      }
      __syncthreads();
      if (tid < 3) {
        float l = INFINITY, h = -INFINITY;
        for (int w = 0; w < SDG_BUILD_THREADS / 64; ++w) {
          l = fminf(l, s_red[tid][w]);
          h = fmaxf(h, s_red[3 + tid][w]);
        }
        unsigned int* box = (first ? X.box : Y.box) + (size_t)b * 8 * SDG_BOX_STRIDE;
        if (l <= h) {  // this workgroup saw a finite value on the axis
          atomicMax(box + tid * SDG_BOX_STRIDE, ~sdg_ord(l));
          atomicMax(box + (3 + tid) * SDG_BOX_STRIDE, sdg_ord(h));
        }
      }
    }
  }
  KAMD_SDG_T(1)
  sdg_grid_barrier(barrier, ++phase, nwg, naps);
  KAMD_SDG_T(2)

  // ---- phase 2: cell id + rank inside the cell (the value the counting atomicAdd returns).  A thread takes SDG_BUILD_UNROLL
  // points at a time: their coordinates are requested together, then their counting atomics -- a point is a chain of two
  // memory-side round trips (~1 + 2 us) and 65 536 threads hold ~3 points each: one after the other that was three chains.
  // The grid's box (six coherent words) is decoded once per (cloud, batch item) a thread meets, not per point.
  const long long per_b = (long long)X.n + Y.n, total = per_b * B;
  const long long stride = (long long)nwg * SDG_BUILD_THREADS;
  {
    int box_u = -1;
    Box bx{};
    for (long long t0 = (long long)wg * SDG_BUILD_THREADS + tid; t0 < total; t0 += stride * SDG_BUILD_UNROLL) {
      float p[SDG_BUILD_UNROLL][3];
      int iu[SDG_BUILD_UNROLL], bu[SDG_BUILD_UNROLL], cu[SDG_BUILD_UNROLL], rk[SDG_BUILD_UNROLL];
      bool fu[SDG_BUILD_UNROLL], on[SDG_BUILD_UNROLL];
#pragma unroll
      for (int u = 0; u < SDG_BUILD_UNROLL; ++u) {
        const long long t = t0 + (long long)u * stride;
        on[u] = t < total;
        const long long tt = on[u] ? t : t0;
        bu[u] = (int)(tt / per_b);
        int i = (int)(tt - (long long)bu[u] * per_b);
        fu[u] = i < X.n;
        if (!fu[u]) i -= X.n;
        iu[u] = i;
        const T* Pt = (fu[u] ? X.pts : Y.pts) + ((size_t)bu[u] * (fu[u] ? X.n : Y.n) + i) * 3;
        p[u][0] = (float)Pt[0];
        p[u][1] = (float)Pt[1];
        p[u][2] = (float)Pt[2];
      }
#pragma unroll
      for (int u = 0; u < SDG_BUILD_UNROLL; ++u) {
        const int G = fu[u] ? X.G : Y.G;
        const int uu = fu[u] ? bu[u] : B + bu[u];
        if (uu != box_u) {
          bx = sdg_box_decode<true>((fu[u] ? X.box : Y.box) + (size_t)bu[u] * 8 * SDG_BOX_STRIDE, G);
          box_u = uu;
        }
        const int cx = sdg_axis_cell(p[u][0], bx.lo[0], bx.inv[0], G);
        const int cy = sdg_axis_cell(p[u][1], bx.lo[1], bx.inv[1], G);
        const int cz = sdg_axis_cell(p[u][2], bx.lo[2], bx.inv[2], G);
        cu[u] = (cz * G + cy) * G + cx;
      }
#pragma unroll
      for (int u = 0; u < SDG_BUILD_UNROLL; ++u) {
        const int G = fu[u] ? X.G : Y.G;
        rk[u] = on[u] ? atomicAdd((fu[u] ? X.count : Y.count) + (size_t)bu[u] * ((size_t)G * G * G) + cu[u], 1) : 0;
      }
#pragma unroll
      for (int u = 0; u < SDG_BUILD_UNROLL; ++u)
        if (on[u]) (fu[u] ? X.cellrank : Y.cellrank)[(size_t)bu[u] * (fu[u] ? X.n : Y.n) + iu[u]] = make_int2(cu[u], rk[u]);
    }
  }
  KAMD_SDG_T(3)
  sdg_grid_barrier(barrier, ++phase, nwg, naps);
  KAMD_SDG_T(4)

  // ---- phase 3: exclusive scan of the counts: start[c] = points in cells < c, start[NC] = n.  A unit = (cloud, batch item,
  // block of SDG_BUILD_THREADS cells).  3a leaves every block's total, 3b adds up the totals before its block and scans it:
  // one or two coherent loads per thread in either phase (adding up the raw counts before a block instead would save the
  // barrier but costs block-index dependent coherent loads per thread, which the hardware does not pipeline: 40 us).
  const int units3 = 2 * B * scan_blocks;
  for (int u = wg; u < units3; u += nwg) {
    const int z = u / (B * scan_blocks), b = (u / scan_blocks) % B, blk = u % scan_blocks;
    const bool first = z == 0;
    const int G = first ? X.G : Y.G, NC = G * G * G;
    const int base = blk * SDG_BUILD_THREADS;
    if (base >= NC || (first ? X.n : Y.n) == 0) continue;
    const int* cnt = (first ? X.count : Y.count) + (size_t)b * NC;
    const int i = base + tid;
    __syncthreads();
    const int tot = sdg_block_inclusive(i < NC ? sdg_ld(cnt + i) : 0, s_wave);
    if (tid == SDG_BUILD_THREADS - 1) sdg_st(sums + ((size_t)z * B + b) * scan_blocks + blk, tot);
  }
  KAMD_SDG_T(5)
  sdg_grid_barrier(barrier, ++phase, nwg, naps);
  KAMD_SDG_T(6)
  for (int u = wg; u < units3; u += nwg) {
    const int z = u / (B * scan_blocks), b = (u / scan_blocks) % B, blk = u % scan_blocks;
    const bool first = z == 0;
    const int G = first ? X.G : Y.G, NC = G * G * G;
    const int base = blk * SDG_BUILD_THREADS;
    if (base >= NC || (first ? X.n : Y.n) == 0) continue;
    const int* cnt = (first ? X.count : Y.count) + (size_t)b * NC;
    int* start = (first ? X.start : Y.start) + (size_t)b * (NC + 1);
    const int* my = sums + ((size_t)z * B + b) * scan_blocks;
    int part = 0;
    for (int k = tid; k < blk; k += SDG_BUILD_THREADS) part += sdg_ld(my + k);
    const int i = base + tid;
    const int v = i < NC ? sdg_ld(cnt + i) : 0;
    __syncthreads();  // s_wave / s_off of the previous unit have been read
    const int before = sdg_block_inclusive(part, s_wave);
    if (tid == SDG_BUILD_THREADS - 1) s_off = before;
    __syncthreads();
    const int off = s_off;
    const int inc = sdg_block_inclusive(v, s_wave);
    if (i < NC) sdg_st(start + i, off + inc - v);
    if (i == NC - 1) sdg_st(start + NC, off + inc);
  }
  KAMD_SDG_T(7)
  sdg_grid_barrier(barrier, ++phase, nwg, naps);
  KAMD_SDG_T(8)

  // ---- phase 4: counting-sort scatter without further atomics: point -> start[cell] + rank, as float4 {xyz, index};
  // SDG_BUILD_UNROLL points of a thread at a time, as in phase 2 (record -> start of its cell -> store: two round trips)
  for (long long t0 = (long long)wg * SDG_BUILD_THREADS + tid; t0 < total; t0 += stride * SDG_BUILD_UNROLL) {
    int iu[SDG_BUILD_UNROLL], bu[SDG_BUILD_UNROLL], st[SDG_BUILD_UNROLL];
    int2 cr[SDG_BUILD_UNROLL];
    bool fu[SDG_BUILD_UNROLL], on[SDG_BUILD_UNROLL];
    T p[SDG_BUILD_UNROLL][3];
#pragma unroll
    for (int u = 0; u < SDG_BUILD_UNROLL; ++u) {
      const long long t = t0 + (long long)u * stride;
      on[u] = t < total;
      const long long tt = on[u] ? t : t0;
      bu[u] = (int)(tt / per_b);
      int i = (int)(tt - (long long)bu[u] * per_b);
      fu[u] = i < X.n;
      if (!fu[u]) i -= X.n;
      iu[u] = i;
      const int n = fu[u] ? X.n : Y.n;
      cr[u] = (fu[u] ? X.cellrank : Y.cellrank)[(size_t)bu[u] * n + i];
      const T* Pt = (fu[u] ? X.pts : Y.pts) + ((size_t)bu[u] * n + i) * 3;
      p[u][0] = Pt[0];
      p[u][1] = Pt[1];
      p[u][2] = Pt[2];
    }
#pragma unroll
    for (int u = 0; u < SDG_BUILD_UNROLL; ++u) {
      const int G = fu[u] ? X.G : Y.G;
      st[u] = sdg_ld((fu[u] ? X.start : Y.start) + (size_t)bu[u] * ((size_t)G * G * G + 1) + cr[u].x);
    }
#pragma unroll
    for (int u = 0; u < SDG_BUILD_UNROLL; ++u) {
      if (!on[u]) continue;
      typename CloudT<T>::V4 rec;
      rec.x = p[u][0];
      rec.y = p[u][1];
      rec.z = p[u][2];
      rec.w = sdg_pack_idx(iu[u], (T)0);
      (fu[u] ? X.sorted : Y.sorted)[(size_t)bu[u] * (fu[u] ? X.n : Y.n) + st[u] + cr[u].y] = rec;
    }
  }
  KAMD_SDG_T(9)
  KAMD_SDG_T_PRINT()
}

// ---- 5. query ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sdg_dist(float tx, float ty, float tz, float qx, float qy, float qz) {
  const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ double sdg_dist(double tx, double ty, double tz, double qx, double qy, double qz) {
  const double dx = tx - qx, dy = ty - qy, dz = tz - qz;
  return __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));  // (the same pin as the all-pairs kernel and the oracle)
}

// at::Half clouds (the reference instantiates its kernel on c10::Half: every operator computes in float and rounds to half,
// sided_distance_cuda.cu:252; restated in oracle_sided_distance_forward_f16): the coordinates travel as floats that hold half
// values, the distance is the same expression with a rounding to half after every operation
__device__ __forceinline__ float sdg_dist_half(float tx, float ty, float tz, float qx, float qy, float qz) {
  const float dx = kamd_hround(tx - qx), dy = kamd_hround(ty - qy), dz = kamd_hround(tz - qz);
  const float xx = kamd_hround(dx * dx), yy = kamd_hround(dy * dy), zz = kamd_hround(dz * dz);
  return kamd_hround(kamd_hround(xx + yy) + zz);
}
template <bool HALF>
__device__ __forceinline__ float sdg_dist_sel(float tx, float ty, float tz, float qx, float qy, float qz) {
  return HALF ? sdg_dist_half(tx, ty, tz, qx, qy, qz) : sdg_dist(tx, ty, tz, qx, qy, qz);
}
template <bool HALF>
__device__ __forceinline__ double sdg_dist_sel(double tx, double ty, double tz, double qx, double qy, double qz) {
  return sdg_dist(tx, ty, tz, qx, qy, qz);
}

#ifndef KAMD_SDG_QUERY_WAVES
#define KAMD_SDG_QUERY_WAVES 7  // waves per SIMD the search kernel is compiled for (72 VGPRs): occupancy is what hides the search's dependent loads
#endif
#ifndef KAMD_SDG_GROUP
#define KAMD_SDG_GROUP 4  // build knob; 100k x 100k (profiles/r02y_sdg.txt): 2 lanes 60 us, 4: 56, 8: 61, 16: 85 for both directions
#endif
constexpr int SDG_GROUP = KAMD_SDG_GROUP;
#ifndef KAMD_SDG_R0
#define KAMD_SDG_R0 1     // first ring of the search: 1 = start with the 3 x 3 x 3 cube of cells (0: with the query's own cell, as in round 2)
#endif
#ifndef KAMD_SDG_BATCH
#define KAMD_SDG_BATCH 1  // rows of a ring a lane takes at a time (> 1: their cell ranges are loaded together; measured at 100k x 100k: 62.9 us either way
                          // on a uniform cloud, 131 (1) vs 139 us (3) on a sphere surface: more registers, lower occupancy)
#endif
#ifndef KAMD_SDG_XCD_CHUNKS
#define KAMD_SDG_XCD_CHUNKS 1  // the search's chunks of queries dealt to the XCDs in contiguous eighths (0: by blockIdx)
#endif
constexpr int SDG_R0 = KAMD_SDG_R0;
[[maybe_unused]] constexpr int SDG_BATCH = KAMD_SDG_BATCH;

// the search for one query, shared by the one-direction and the two-direction kernels.  All SDG_GROUP lanes of a query
// call it with the same (qx, qy, qz, c); on return every lane holds the query's (best, best_i).
// TRACK: best_k = the winner's position in the sorted targets (-1 while the seed holds: the caller then looks target 0 up).
// HALF (T = float): the clouds hold half values and distances are computed as c10::Half would (sdg_dist_half).  A computed
// distance then sits within 5 roundings of 2^-11 of the true one (plus the half subnormals' 6e-8 step), so the stopping rule
// leaves that margin; the many exact ties of an 11-bit mantissa are broken by index like any other.
template <bool TRACK, typename T, bool HALF = false>
__device__ __forceinline__ void sdg_search(const Box& s_box, int G, T qx, T qy, T qz, int cx, int cy, int cz,
                                           const T* __restrict__ T0, const int* __restrict__ start,
                                           const typename Vec4Of<T>::type* __restrict__ TS, int nt, int sub, T& best, int& best_i,
                                           int& best_k) {
  typedef typename Vec4Of<T>::type V4;
  // the reference's seed: target 0 unconditionally (a NaN distance sticks)
  best = sdg_dist_sel<HALF>(T0[0], T0[1], T0[2], qx, qy, qz);
  best_i = 0;
  best_k = -1;
  if (sizeof(T) == 8 && !(fabs((double)qx) < 1e30 && fabs((double)qy) < 1e30 && fabs((double)qz) < 1e30)) {
    // a double query outside float's range (or not finite): the float grid says nothing about it -- walk every target
    if (best == best) {
      for (int k = sub; k < nt; k += SDG_GROUP) {
        const V4 t = TS[k];
        const T d = sdg_dist_sel<HALF>(t.x, t.y, t.z, qx, qy, qz);
        const int ti = sdg_unpack_idx(t.w);
        if (d < best || (d == best && ti < best_i)) {
          best = d;
          best_i = ti;
          if (TRACK) best_k = k;
        }
      }
#pragma unroll
      for (int m = 1; m < SDG_GROUP; m <<= 1) {
        const T od = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(best_i, m, 64);
        const int ok = TRACK ? __shfl_xor(best_k, m, 64) : 0;
        if (od < best || (od == best && oi < best_i)) {
          best = od;
          best_i = oi;
          if (TRACK) best_k = ok;
        }
      }
    }
    return;
  }
  if (best == best) {  // uniform within the group (same query)
    // rounding head-room of the geometric bound: cell membership is decided by a rounded (v - lo) * inv
    const float q[3] = {(float)qx, (float)qy, (float)qz};
    float slack[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) slack[a] = 4e-6f * (fabsf(q[a]) + fabsf(s_box.lo[a]) + s_box.size[a] * (float)G);
    const int cq[3] = {cx, cy, cz};
    // The first ring visited is the 3 x 3 x 3 cube (SDG_R0 = 1): with ~2 targets per cell the single cell of ring 0 almost
    // never settles a query (the nearest target is ~0.44 cells away, the cell's walls ~0.17), and every ring is a chain of
    // dependent round trips (cell ranges -> targets -> merge).  Visiting more cells than needed changes nothing: the
    // search is exact and ties are broken explicitly.
    const int r_first = G >= 3 ? SDG_R0 : 0;
    for (int r = r_first; r < G; ++r) {
      const int z0 = max(cz - r, 0), z1 = min(cz + r, G - 1);
      const int y0 = max(cy - r, 0), y1 = min(cy + r, G - 1);
      const int x0 = max(cx - r, 0), x1 = min(cx + r, G - 1);
      const int ny = y1 - y0 + 1, nrows = (z1 - z0 + 1) * ny;
#if KAMD_SDG_BATCH > 1
      // a lane's rows are taken SDG_BATCH at a time: the cell ranges of the whole batch are requested before any of its
      // targets, the targets two at a time -- fewer dependent round trips per ring (was: range -> target -> target ... per row)
      for (int j0 = sub; j0 < nrows; j0 += SDG_BATCH * SDG_GROUP) {
        int k0[SDG_BATCH][2], k1[SDG_BATCH][2], nseg[SDG_BATCH];
#pragma unroll
        for (int u = 0; u < SDG_BATCH; ++u) {
          const int j = j0 + u * SDG_GROUP;
          nseg[u] = 0;
          k0[u][0] = k0[u][1] = k1[u][0] = k1[u][1] = 0;
          if (j < nrows) {
            const int z = z0 + j / ny, y = y0 + j % ny;
            const int row = (z * G + y) * G;
            const bool shell_row = r == r_first || (abs(z - cz) == r) || (abs(y - cy) == r);
            if (shell_row) {
              k0[u][0] = start[row + x0];
              k1[u][0] = start[row + x1 + 1];
              nseg[u] = 1;
            } else {
              // (segment 0: the cell at cx - r, segment 1: the cell at cx + r; an absent one stays empty)
              if (cx - r >= 0) {
                k0[u][0] = start[row + cx - r];
                k1[u][0] = start[row + cx - r + 1];
              }
              if (cx + r <= G - 1) {
                k0[u][1] = start[row + cx + r];
                k1[u][1] = start[row + cx + r + 1];
              }
              nseg[u] = 2;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < SDG_BATCH; ++u)
#pragma unroll
          for (int sgm = 0; sgm < 2; ++sgm) {  // (compile-time indices: the arrays stay in registers)
            const int kb = k0[u][sgm], ke = sgm < nseg[u] ? k1[u][sgm] : kb;
            for (int k = kb; k < ke; k += 2) {
              const bool two = k + 1 < ke;
              const V4 ta = TS[k];
              const V4 tb = TS[two ? k + 1 : k];
              {
                const T d = sdg_dist_sel<HALF>(ta.x, ta.y, ta.z, qx, qy, qz);
                const int ti = sdg_unpack_idx(ta.w);
                if (d < best || (d == best && ti < best_i)) {
                  best = d;
                  best_i = ti;
                  if (TRACK) best_k = k;
                }
              }
              if (two) {
                const T d = sdg_dist_sel<HALF>(tb.x, tb.y, tb.z, qx, qy, qz);
                const int ti = sdg_unpack_idx(tb.w);
                if (d < best || (d == best && ti < best_i)) {
                  best = d;
                  best_i = ti;
                  if (TRACK) best_k = k + 1;
                }
              }
            }
          }
      }
#else
      // The first ring (at most 3 x 3 rows; fp32) by the query's four lanes TOGETHER: every lane requests the cell ranges of its
      // rows (sub, sub + 4, sub + 8) at once, the ranges travel inside the quad (DPP quad_perm: register moves), and the four
      // lanes walk every row side by side -- lane s takes the targets k0 + s, k0 + s + 4, ... -- so that a quad's loads fall on
      // ONE 64-byte line.  The search is bound by the lines a gather instruction touches (the CU's texture path takes one line
      // per clock: with a lane per row every target load of a wavefront touched 64 different lines, ~25 M line accesses per
      // call at 100k x 100k = the launch's duration; measured profiles/r06q_*: neither fewer iterations nor more loads in flight
      // changed it, on the contrary).  (d, idx) compare as one 64-bit key: a distance is a sum of squares, never negative, so
      // its bit pattern orders as the value does, and the index in the low word breaks ties towards the lower one.
      bool ring_done = false;
      if constexpr (sizeof(T) == 4 && SDG_GROUP == 4) {
        if (r == r_first && nrows <= 3 * SDG_GROUP) {
          int ks[3], ke[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int j = sub + u * SDG_GROUP;
            ks[u] = 0;
            ke[u] = 0;
            if (j < nrows) {
              const int zq = (j >= ny ? 1 : 0) + (j >= 2 * ny ? 1 : 0);  // (j / ny: at most three rows of rows)
              const int row = ((z0 + zq) * G + (y0 + j - zq * ny)) * G;
              ks[u] = start[row + x0];
              ke[u] = start[row + x1 + 1];
            }
          }
          unsigned long long kbest = ((unsigned long long)__float_as_uint((float)best) << 32) | (unsigned int)best_i;
          int kk = best_k;
          // a row's range from the lane that holds it (quad_perm: lane SRC of the quad for everybody), then its targets, four at a time.
          // (Measured and not kept, profiles/r06s_*: the next target of ALL nine rows requested together, 2 or 3 such passes -- 72 -> 128
          // registers, 7 -> 4 wavefronts per SIMD: 75.9 vs 53.5 us; occupancy is what hides this search's dependent loads.)
#define KAMD_SDG_ROW(U, SRC)                                                                                              \
          {                                                                                                               \
            const int ka = __builtin_amdgcn_update_dpp(0, ks[U], (SRC) * 0x55, 0xF, 0xF, false);                            \
            const int kz = __builtin_amdgcn_update_dpp(0, ke[U], (SRC) * 0x55, 0xF, 0xF, false);                            \
            for (int k = ka + sub; k < kz; k += SDG_GROUP) {                                                              \
              const V4 t = TS[k];                                                                                         \
              const float d = (float)sdg_dist_sel<HALF>(t.x, t.y, t.z, qx, qy, qz);                                      \
              const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)sdg_unpack_idx(t.w); \
              if (key < kbest) {                                                                                          \
                kbest = key;                                                                                              \
                kk = k;                                                                                                   \
              }                                                                                                           \
            }                                                                                                             \
          }
          KAMD_SDG_ROW(0, 0) KAMD_SDG_ROW(0, 1) KAMD_SDG_ROW(0, 2) KAMD_SDG_ROW(0, 3)
          KAMD_SDG_ROW(1, 0) KAMD_SDG_ROW(1, 1) KAMD_SDG_ROW(1, 2) KAMD_SDG_ROW(1, 3)
          KAMD_SDG_ROW(2, 0)
#undef KAMD_SDG_ROW
          best = (T)__uint_as_float((unsigned int)(kbest >> 32));
          best_i = (int)(unsigned int)kbest;
          if (TRACK) best_k = kk;
          ring_done = true;
        }
      }
      for (int j = sub; !ring_done && j < nrows; j += SDG_GROUP) {
        const int z = z0 + j / ny, y = y0 + j % ny;
        const int row = (z * G + y) * G;
        const bool shell_row = r == r_first || (abs(z - cz) == r) || (abs(y - cy) == r);
        // cells are numbered x-fastest and targets are sorted by cell: the cells x0..x1 of a row own ONE contiguous
        // slice of the sorted targets.  On a shell row every x is new; elsewhere only the two end cells are.
        int k0[2], k1[2], nseg;
        if (shell_row) {
          k0[0] = start[row + x0];
          k1[0] = start[row + x1 + 1];
          nseg = 1;
        } else {
          nseg = 0;
          if (cx - r >= 0) {
            k0[nseg] = start[row + cx - r];
            k1[nseg] = start[row + cx - r + 1];
            ++nseg;
          }
          if (cx + r <= G - 1) {
            k0[nseg] = start[row + cx + r];
            k1[nseg] = start[row + cx + r + 1];
            ++nseg;
          }
        }
        for (int sgm = 0; sgm < nseg; ++sgm)
          for (int k = k0[sgm]; k < k1[sgm]; ++k) {
            const V4 t = TS[k];
            const T d = sdg_dist_sel<HALF>(t.x, t.y, t.z, qx, qy, qz);
            const int ti = sdg_unpack_idx(t.w);
            if (d < best || (d == best && ti < best_i)) {
              best = d;
              best_i = ti;
              if (TRACK) best_k = k;
            }
          }
      }
#endif
#pragma unroll
      for (int m = 1; m < SDG_GROUP; m <<= 1) {  // lexicographic (dist, idx) minimum over the group
        const T od = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(best_i, m, 64);
        const int ok = TRACK ? __shfl_xor(best_k, m, 64) : 0;
        if (od < best || (od == best && oi < best_i)) {
          best = od;
          best_i = oi;
          if (TRACK) best_k = ok;
        }
      }
      // every target outside the cube of cells [c - r, c + r] is at least `bound` away from the query
      float bound = INFINITY;
      bool whole_grid = true;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (cq[a] - r > 0) {
          whole_grid = false;
          bound = fminf(bound, q[a] - (s_box.lo[a] + (float)(cq[a] - r) * s_box.size[a]) - slack[a]);
        }
        if (cq[a] + r < G - 1) {
          whole_grid = false;
          bound = fminf(bound, (s_box.lo[a] + (float)(cq[a] + r + 1) * s_box.size[a]) - q[a] - slack[a]);
        }
      }
      if (whole_grid) break;
      // (squared in T: a float square overflows to +inf from bound ~1.8e19 on, and any finite fp64 `best` would then stop the
      // search after the first ring -- fp64 clouds reach the grid path with coordinates up to 1e30)
      if (HALF) {
        // every unvisited target's computed half distance is >= bound^2 (1 - 5 * 2^-11) - 1e-7 (an overflow to +inf included)
        if (bound > 0.f && (float)best < bound * bound * 0.9975f - 2e-7f) break;
      } else if (bound > 0.f && best < (T)bound * (T)bound * (T)0.99999f) {
        break;
      }
    }
  }
}


// The reference re-seeds at every tile of 512 targets (reseed.h): a winner inside a tile whose first target yields NaN is redone
// over the live tiles by the whole wavefront.  Only clouds with a non-finite point at an index 512 k get here (the build leaves a
// flag in the box record); out of line, so that the search kernels' registers are the search's.
template <typename S>
struct Reseeded {
  S best;
  int best_i, redone;
};
template <typename S, bool HALF>
__device__ __attribute__((noinline)) Reseeded<S> sdg_reseed(bool live, int sub, S qx, S qy, S qz, const S* Tp, int nt, S best, int best_i) {
  auto dist_f = [](S tx, S ty, S tz, S x, S y, S z) { return sdg_dist_sel<HALF>(tx, ty, tz, x, y, z); };
  auto load_f = [](const S* p) { return *p; };
  const bool need = live && sd_winner_in_dead_tile<S>(Tp, best_i, qx, qy, qz, dist_f, load_f);  // (uniform within a query's group)
  Reseeded<S> r{best, best_i, 0};
  if (!__any(need)) return r;
  sd_reseed_fix<S>(need && sub == 0, qx, qy, qz, Tp, nt, best, best_i, dist_f, load_f);
  // the group's other lanes take lane 0's result
  const int g0 = (threadIdx.x & 63) - sub;
  const S fb = reseed_shfl(best, g0);
  const int fi = reseed_shfl(best_i, g0);
  if (need) r = Reseeded<S>{fb, fi, 1};
  return r;
}

// blockIdx.z = direction: 0 answers the queries A against the targets T (dist1 / idx1), 1 the reverse (dist2 / idx2).
// The search runs on the TARGETS' grid; the queries only have to arrive in a spatially coherent order, which their own
// sorted copy provides whichever grid it was sorted on.
// MODE >= SDG_VALUE (chamfer_distance, kaolin/metrics/pointcloud.py:120-136): the launch also reduces f(dist) (f = identity
// or sqrt) per direction -- wave shuffle, LDS, one partial sum per workgroup; sdg_chamfer_value closes
// w1 * mean1 + w2 * mean2.  MODE == SDG_GRAD: every query also leaves d value / d (its own point) and adds
// d value / d (its nearest point) with float atomics, both already scaled by w / n [/ (2 sqrt(dist))]: the atomics ride in
// a kernel that waits on dependent loads anyway, and the backward pass is one multiply by the upstream gradient.
// O = the element type of the distance outputs (S, or __half for at::Half clouds whose coordinates arrive as floats: HALF)
template <typename O, typename S>
__device__ __forceinline__ O sdg_out(S v) { return (O)v; }
template <>
__device__ __forceinline__ __half sdg_out<__half, float>(float v) { return __float2half(v); }   // (exact: v holds a half value)
template <int MODE, typename S, bool HALF = false, typename O = S>
__global__ __launch_bounds__(256, (MODE == SDG_GRAD ? KAMD_SDG_QUERY_WAVES : 1)) void sdg_query(CloudT<S> A, CloudT<S> T, O* __restrict__ dist1, int64_t* __restrict__ idx1,
                                                 O* __restrict__ dist2, int64_t* __restrict__ idx2, Fuse fz) {
  static_assert(MODE == SDG_PLAIN || (sizeof(S) == 4 && !HALF), "the chamfer modes are fp32");
  typedef typename Vec4Of<S>::type V4;
  __shared__ Box s_box;
  __shared__ double s_sum[4];
  __shared__ unsigned int s_odd_first;
  const int b = blockIdx.y;
  const bool fwd = blockIdx.z == 0;
  const int nq = fwd ? A.n : T.n, nt = fwd ? T.n : A.n, G = fwd ? T.G : A.G;
  const int NC = G * G * G;
  const S* Tp = (fwd ? T.pts : A.pts) + (size_t)b * nt * 3;
  const int* Tstart = (fwd ? T.start : A.start) + (size_t)b * (NC + 1);
  const V4* Tsorted = (fwd ? T.sorted : A.sorted) + (size_t)b * nt;
  if (threadIdx.x == 0) {
    s_box = sdg_box_decode((fwd ? T.box : A.box) + (size_t)b * 8 * SDG_BOX_STRIDE, G);
    s_odd_first = ((fwd ? T.box : A.box) + (size_t)b * 8 * SDG_BOX_STRIDE)[6 * SDG_BOX_STRIDE];  // the build found a non-finite target at an index 512 k
  }
  __syncthreads();
  const bool odd_first = s_odd_first != 0u;  // (workgroup-uniform)
  // 100k queries are only ~1.5 wavefronts per SIMD and every query is a chain of dependent loads (cell range ->
  // targets): SDG_GROUP lanes share a query so that as many more loads are in flight; the lanes' (dist, idx) are merged
  // with a butterfly after every ring
  const int sub = threadIdx.x % SDG_GROUP;
  double term = 0.0;
  // a workgroup takes the chunks of 256 / SDG_GROUP queries blockIdx.x, blockIdx.x + gridDim.x, ...: one chunk each unless the chamfer
  // modes' partial-sum buffer holds fewer workgroups than there are chunks
  const int nchunks = (int)(((long long)nq * SDG_GROUP + 255) / 256);
  // Which chunk a workgroup starts with: workgroups are dealt to the eight XCDs round-robin by their flat id, and every XCD has an
  // L2 of its own.  Queries arrive in cell order, so CONSECUTIVE chunks read neighbouring targets: the workgroups of one XCD take
  // one contiguous eighth of the chunks and its L2 fetches an eighth of the targets (plus a rim) instead of all of them -- with the
  // plain blockIdx order every XCD missed on every line (TCC_MISS 414 k of 846 k requests per call, profiles/r06s_*).
  int chunk0 = blockIdx.x;
  if (KAMD_SDG_XCD_CHUNKS) {
    const unsigned int gx = gridDim.x, base = (gx * (blockIdx.y + gridDim.y * blockIdx.z)) & 7u;
    const unsigned int xcd = (base + blockIdx.x) & 7u;
    unsigned int before = 0u;  // workgroups of this slice on the XCDs dealt before this one (in the order of their first workgroup)
    for (unsigned int r = 0; r < 8u; ++r) {
      const unsigned int first = (r + 8u - base) & 7u;         // first blockIdx.x of the slice on XCD r
      const unsigned int cnt = first < gx ? (gx - first + 7u) / 8u : 0u;
      if (r < xcd) before += cnt;
    }
    const unsigned int first_mine = (xcd + 8u - base) & 7u;
    chunk0 = (int)(before + (blockIdx.x - first_mine) / 8u);
  }
  for (int chunk = chunk0; chunk < nchunks; chunk += gridDim.x) {
    const int slot = (chunk * 256 + threadIdx.x) / SDG_GROUP;
    const bool live = slot < nq;
    const V4 q = (fwd ? A.sorted : T.sorted)[(size_t)b * nq + (live ? slot : 0)];
    const int cx = sdg_axis_cell((float)q.x, s_box.lo[0], s_box.inv[0], G);
    const int cy = sdg_axis_cell((float)q.y, s_box.lo[1], s_box.inv[1], G);
    const int cz = sdg_axis_cell((float)q.z, s_box.lo[2], s_box.inv[2], G);
    S best;
    int best_i, best_k;
    sdg_search<MODE == SDG_GRAD, S, HALF>(s_box, G, q.x, q.y, q.z, cx, cy, cz, Tp, Tstart, Tsorted, nt, sub, best, best_i, best_k);
    if (__builtin_expect(odd_first, 0)) {
      const Reseeded<S> r = sdg_reseed<S, HALF>(live, sub, q.x, q.y, q.z, Tp, nt, best, best_i);
      best = r.best;
      best_i = r.best_i;
      if (r.redone) best_k = -1;  // (looked up below, like the seed's)
    }
    // Results leave from the first lanes of the query's group (every lane holds the merged result): lane 0 writes distance and
    // index and adds the term; in the chamfer gradient mode lanes 0..2 take one coordinate each -- the 12 bytes of the query's own
    // term are then consecutive lanes of ONE store instruction and the nearest target's three atomics ONE request (a global
    // float atomic costs per line touched by an instruction, ~60 ps chip-wide: lane 0 adding x, y, z in turn made 600k
    // requests per call at 100k x 100k, 36 us of this launch's 63)
    constexpr int OUT_LANES = (MODE == SDG_GRAD && SDG_GROUP >= 3) ? 3 : 1;
    if (live && sub < OUT_LANES) {
      if (sub == 0) {
        const size_t o = (size_t)b * nq + sdg_unpack_idx(q.w);
        O* dist = fwd ? dist1 : dist2;
        int64_t* idx = fwd ? idx1 : idx2;
        if (dist != nullptr) dist[o] = sdg_out<O, S>(best);
        if (idx != nullptr) idx[o] = best_i;
      }
      if constexpr (MODE >= SDG_VALUE) {
        const float root = fz.squared ? best : sqrtf(best);
        if (sub == 0) term += (double)root;
        if (MODE == SDG_GRAD) {
          // value = sum_b up_b * (w1 / N * sum_i f(dist1_i) + w2 / M * sum_j f(dist2_j)); d dist / d q = 2 (q - t)
          // Both gradient arrays are laid out in the SORTED order of the cloud they belong to: this query's own term is a
          // coalesced store at its slot, and the atomics of neighbouring queries land on neighbouring targets
          float k = fwd ? fz.c1 : fz.c2;
          if (!fz.squared) k = k / (2.f * root);
          if (best_k < 0) {  // the seed (target 0) is the nearest, or the winner was redone: its place in the sorted targets
            const int2 cr = (fwd ? T.cellrank : A.cellrank)[(size_t)b * nt + best_i];
            best_k = Tstart[cr.x] + cr.y;
          }
          const V4 t = Tsorted[best_k];
          float* own = (fwd ? fz.own_a : fz.own_b) + ((size_t)b * nq + slot) * 3;
          float* scat = (fwd ? fz.scat_b : fz.scat_a) + ((size_t)b * nt + best_k) * 3;
          if (OUT_LANES == 3) {
            const float qa = sub == 0 ? (float)q.x : (sub == 1 ? (float)q.y : (float)q.z);
            const float ta = sub == 0 ? (float)t.x : (sub == 1 ? (float)t.y : (float)t.z);
            own[sub] = 2.f * (qa - ta) * k;
            kamd_atomic_add(scat + sub, 2.f * (ta - qa) * k);
          } else {
            const float qv[3] = {(float)q.x, (float)q.y, (float)q.z}, tv[3] = {(float)t.x, (float)t.y, (float)t.z};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              own[a] = 2.f * (qv[a] - tv[a]) * k;
              kamd_atomic_add(scat + a, 2.f * (tv[a] - qv[a]) * k);
            }
          }
        }
      }
    }
  }
  if (MODE >= SDG_VALUE) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) term += __shfl_xor(term, d, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0)
      fz.sums[((size_t)blockIdx.z * gridDim.y + b) * gridDim.x + blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
  }
}

// Closes chamfer_distance: out[b] = w1 * mean1 + w2 * mean2 from the query workgroups' partial sums, added up in a fixed
// order (the value is deterministic).  A workgroup per batch item, every thread a strided share of the partials with
// independent loads in flight.  (Measured on the way here, profiles/r02n_chamfer.txt: the same sum as ONE wavefront's loop of
// dependent load -> add steps -- first inside the query launch's last workgroup, then here -- is a chain of ~50 memory
// round trips: +35 us on a 61 us search.)
__global__ __launch_bounds__(256) void sdg_chamfer_value(int B, int gx, int N, int M, const double* __restrict__ partial, float w1,
                                                         float w2, float* __restrict__ out) {
  __shared__ double s_part[2][4];
  const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* p1 = partial + (size_t)i * gx;
  const double* p2 = partial + ((size_t)B + i) * gx;
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  for (int x0 = threadIdx.x; x0 < gx; x0 += 4 * 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = x0 + k * 256;
      a1[k] += x < gx ? p1[x] : 0.0;
      a2[k] += x < gx ? p2[x] : 0.0;
    }
  }
  double s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    s1 += __shfl_xor(s1, d, 64);
    s2 += __shfl_xor(s2, d, 64);
  }
  if (lane == 0) {
    s_part[0][wave] = s1;
    s_part[1][wave] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = (s_part[0][0] + s_part[0][1]) + (s_part[0][2] + s_part[0][3]);
    s2 = (s_part[1][0] + s_part[1][1]) + (s_part[1][2] + s_part[1][3]);
    const float m1 = (float)(s1 / (double)N), m2 = (float)(s2 / (double)M);
    out[i] = (w1 == 1.f && w2 == 1.f) ? m1 + m2 : w1 * m1 + w2 * m2;
  }
}

// chamfer backward after a SDG_GRAD forward: grad_p[original index] = upstream[b] * (own + scattered)[sorted position]
__global__ __launch_bounds__(256) void sdg_chamfer_apply(int B, Cloud A, Cloud T, const float* __restrict__ grad,
                                                         const float* __restrict__ own_a, const float* __restrict__ scat_a,
                                                         const float* __restrict__ own_b, const float* __restrict__ scat_b,
                                                         float* __restrict__ g1, float* __restrict__ g2) {
  const long long na = (long long)B * A.n, total = na + (long long)B * T.n;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const bool first = t < na;
    const long long k = first ? t : t - na;            // (batch item, sorted position) flattened
    const int n = first ? A.n : T.n;
    const int b = (int)(k / n);
    const float g = grad[b];
    const int orig = __float_as_int((first ? A.sorted : T.sorted)[k].w);
    const float* own = (first ? own_a : own_b) + k * 3;
    const float* scat = (first ? scat_a : scat_b) + k * 3;
    float* out = (first ? g1 : g2) + ((long long)b * n + orig) * 3;
    out[0] = g * (own[0] + scat[0]);
    out[1] = g * (own[1] + scat[1]);
    out[2] = g * (own[2] + scat[2]);
  }
}

inline int sdg_num_cus() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 64;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 64;
    cached[dev] = n;
  }
  return cached[dev];
}

template <typename S>
int sdg_run(hipStream_t st, int B, int N, int M, const S* p1, const S* p2, S* dist1, int64_t* idx1,
            S* dist2, int64_t* idx2, void* workspace, bool pair, int mode, float w1, float w2, int squared, float* out) {
  SdgWs w = sdg_layout(workspace, B, N, M, p1, p2, pair, mode, (int)sizeof(S));
  const CloudT<S> ca = cloud_as<S>(w.a), cb = cloud_as<S>(w.b);
  KAMD_CHECK(kamd_zero_async(workspace, w.zero_bytes, st));
  {
    ProfScope p(K_SDG_BUILD, st);
    // every workgroup must be resident (grid barrier): at most one per CU, fewer for small inputs
    int nwg = kamd_cdiv((long long)B * ((long long)N + M), SDG_BUILD_THREADS);
    const int cus = sdg_num_cus();
    if (nwg > cus) nwg = cus;
    // One workgroup per CU.  Measured (profiles/r06zz_sdg_build_workgroups.txt) since the barrier is two-level and the box words
    // sit on lines of their own: 100k + 100k points 37.3 us with 128 workgroups, 32.8 with 256, 33.7 with 512; the 8 items of C3
    // as one call 155 / 124 / 118 us.  Not two per CU: the kernel spins on a grid barrier, every workgroup must become resident,
    // and up to four queues of a process run kernels side by side -- four concurrent builds of 256 workgroups of 8 wavefronts are
    // exactly the chip's 8 192 wavefront slots, four of 512 could wait for each other for ever.
    const int cap = kamd_env_int("KAMD_SDG_WGS", cus);
    if (nwg > cap) nwg = cap;
    const int naps = kamd_env_int("KAMD_SDG_NAPS", 4);
    hipLaunchKernelGGL(sdg_build<S>, dim3(nwg), dim3(SDG_BUILD_THREADS), 0, st, cb, ca, B, pair ? 1 : 0, w.scan_sums,
                       w.scan_blocks, w.barrier, naps);
  }
  KAMD_CHECK(hipGetLastError());
  {
    ProfScope p(K_SDG_QUERY, st);
    const int big = (pair && M > N) ? M : N;
    int gx = kamd_cdiv((long long)big * SDG_GROUP, 256);
    if (mode != SDG_PLAIN) {  // every workgroup leaves one partial sum: at most as many as the buffer holds
      const int cap = (int)(sdg_partials(B) / ((size_t)2 * B));
      if (gx > cap) gx = cap;
    }
    const dim3 grid(gx, B, pair ? 2 : 1);
    w.fuse.w1 = w1;
    w.fuse.w2 = w2;
    w.fuse.c1 = w1 * (1.f / (float)N);
    w.fuse.c2 = w2 * (1.f / (float)M);
    w.fuse.squared = squared;
    if constexpr (sizeof(S) == 4) {
      if (mode == SDG_PLAIN)
        hipLaunchKernelGGL((sdg_query<SDG_PLAIN, S>), grid, dim3(256), 0, st, ca, cb, dist1, idx1, dist2, idx2, w.fuse);
      else if (mode == SDG_VALUE)
        hipLaunchKernelGGL((sdg_query<SDG_VALUE, S>), grid, dim3(256), 0, st, ca, cb, dist1, idx1, dist2, idx2, w.fuse);
      else
        hipLaunchKernelGGL((sdg_query<SDG_GRAD, S>), grid, dim3(256), 0, st, ca, cb, dist1, idx1, dist2, idx2, w.fuse);
    } else {
      if (mode != SDG_PLAIN) return (int)hipErrorInvalidValue;
      hipLaunchKernelGGL((sdg_query<SDG_PLAIN, S>), grid, dim3(256), 0, st, ca, cb, dist1, idx1, dist2, idx2, w.fuse);
    }
    if (mode != SDG_PLAIN)
      hipLaunchKernelGGL(sdg_chamfer_value, dim3(B), dim3(256), 0, st, B, gx, N, M, (const double*)w.fuse.sums, w1, w2, out);
  }
  KAMD_RETURN_LAST_ERROR();
}

// at::Half clouds: converted to float once (exact), binned and searched by the float pipeline with half-rounded distances
__global__ __launch_bounds__(256) void sdg_half_to_float(const __half* __restrict__ a, size_t na, const __half* __restrict__ b, size_t nb,
                                                         float* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += stride)
    out[i] = __half2float(i < na ? a[i] : b[i - na]);
}
inline size_t sdg_half_copy_bytes(int B, int N, int M) { return al((size_t)B * ((size_t)N + M) * 3 * 4); }

}  // namespace

int sdgrid_forward_f16(hipStream_t st, int B, int N, int M, const void* p1, const void* p2, void* dist, int64_t* idx, void* workspace) {
  float* f1 = (float*)workspace;
  float* f2 = f1 + (size_t)B * N * 3;
  void* rest = (char*)workspace + sdg_half_copy_bytes(B, N, M);
  const size_t na = (size_t)B * N * 3, nb = (size_t)B * M * 3;
  {
    ProfScope p(K_SDG_BUILD, st);
    size_t blocks = (na + nb + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sdg_half_to_float, dim3((unsigned)blocks), dim3(256), 0, st, (const __half*)p1, na, (const __half*)p2, nb, f1);
  }
  KAMD_CHECK(hipGetLastError());
  SdgWs w = sdg_layout(rest, B, N, M, f1, f2, false, SDG_PLAIN, 4);
  KAMD_CHECK(kamd_zero_async(rest, w.zero_bytes, st));
  {
    ProfScope p(K_SDG_BUILD, st);
    int nwg = kamd_cdiv((long long)B * ((long long)N + M), SDG_BUILD_THREADS);
    const int cus = sdg_num_cus();
    if (nwg > cus) nwg = cus;
    if (nwg > 128) nwg = 128;
    hipLaunchKernelGGL(sdg_build<float>, dim3(nwg), dim3(SDG_BUILD_THREADS), 0, st, w.b, w.a, B, 0, w.scan_sums, w.scan_blocks, w.barrier, 4);
  }
  KAMD_CHECK(hipGetLastError());
  {
    ProfScope p(K_SDG_QUERY, st);
    const dim3 grid(kamd_cdiv((long long)N * SDG_GROUP, 256), B, 1);
    hipLaunchKernelGGL((sdg_query<SDG_PLAIN, float, true, __half>), grid, dim3(256), 0, st, w.a, w.b, (__half*)dist, idx, (__half*)nullptr,
                       (int64_t*)nullptr, w.fuse);
  }
  KAMD_RETURN_LAST_ERROR();
}

bool sdgrid_applicable(int B, int N, int M) {
  // below this the brute-force kernels are as fast as the launches of the grid pipeline
  return B >= 1 && B <= 65535 && M >= 8192 && N >= 2048 && (long long)B * (long long)(M > N ? M : N) < (1ll << 30);
}
size_t sdgrid_workspace_bytes(int B, int N, int M, int elem_size) {
  if (elem_size == 2)  // at::Half: float copies of both clouds in front of the float pipeline's workspace
    return sdg_half_copy_bytes(B, N, M) + sdg_layout(nullptr, B, N, M, nullptr, nullptr, false, SDG_PLAIN, 4).total;
  return sdg_layout(nullptr, B, N, M, nullptr, nullptr, false, SDG_PLAIN, elem_size).total;
}
int sdgrid_forward_f64(hipStream_t st, int B, int N, int M, const double* p1, const double* p2, double* dist, int64_t* idx,
                       void* workspace) {
  return sdg_run<double>(st, B, N, M, p1, p2, dist, idx, nullptr, nullptr, workspace, false, SDG_PLAIN, 1.f, 1.f, 1, nullptr);
}

int sdgrid_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist, int64_t* idx,
                       void* workspace) {
  return sdg_run<float>(st, B, N, M, p1, p2, dist, idx, nullptr, nullptr, workspace, false, SDG_PLAIN, 1.f, 1.f, 1, nullptr);
}

bool sdgrid_pair_applicable(int B, int N, int M) { return sdgrid_applicable(B, N, M) && sdgrid_applicable(B, M, N); }
size_t sdgrid_pair_workspace_bytes(int B, int N, int M, int elem_size) {
  return sdg_layout(nullptr, B, N, M, nullptr, nullptr, true, SDG_PLAIN, elem_size).total;
}
int sdgrid_pair_forward_f64(hipStream_t st, int B, int N, int M, const double* p1, const double* p2, double* dist1,
                            int64_t* idx1, double* dist2, int64_t* idx2, void* workspace) {
  return sdg_run<double>(st, B, N, M, p1, p2, dist1, idx1, dist2, idx2, workspace, true, SDG_PLAIN, 1.f, 1.f, 1, nullptr);
}
int sdgrid_pair_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist1,
                            int64_t* idx1, float* dist2, int64_t* idx2, void* workspace) {
  return sdg_run<float>(st, B, N, M, p1, p2, dist1, idx1, dist2, idx2, workspace, true, SDG_PLAIN, 1.f, 1.f, 1, nullptr);
}

size_t sdgrid_chamfer_workspace_bytes(int B, int N, int M, bool with_grad) {
  return sdg_layout(nullptr, B, N, M, nullptr, nullptr, true, with_grad ? SDG_GRAD : SDG_VALUE).total;
}
int sdgrid_chamfer_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float w1, float w2,
                               int squared, bool with_grad, float* out, float* dist1, int64_t* idx1, float* dist2,
                               int64_t* idx2, void* workspace) {
  return sdg_run<float>(st, B, N, M, p1, p2, dist1, idx1, dist2, idx2, workspace, true, with_grad ? SDG_GRAD : SDG_VALUE, w1, w2,
                 squared, out);
}
int sdgrid_chamfer_backward_f32(hipStream_t st, int B, int N, int M, const float* grad, void* workspace, float* g1,
                                float* g2) {
  const SdgWs w = sdg_layout(workspace, B, N, M, nullptr, nullptr, true, SDG_GRAD);
  const long long total = (long long)B * ((long long)N + M);
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  ProfScope p(K_SD_BACKWARD, st);
  hipLaunchKernelGGL(sdg_chamfer_apply, dim3((unsigned)blocks), dim3(256), 0, st, B, w.a, w.b, grad, w.fuse.own_a, w.fuse.scat_a,
                     w.fuse.own_b, w.fuse.scat_b, g1, g2);
  KAMD_RETURN_LAST_ERROR();
}

}  // namespace kamd
