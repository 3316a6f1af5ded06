// sided_distance forward, exact uniform-grid search (fp32) for MI355X (gfx950).
//
// The reference (kaolin/csrc/metrics/sided_distance_cuda.cu:52-201) is an all-pairs search: N*M distance
// evaluations against 24*(N+M) bytes of input, 8 300 FLOP/B at 100k x 100k -- VALU-bound by construction
// (sd_main_f32 already issues at 97 % of the measured FMA peak).  The only way to go faster is to evaluate fewer
// pairs while returning the SAME answer: dist = min_j d(p1_i, p2_j) with the identical fp32 expression
// d = fma(dz,dz, fma(dy,dy, dx*dx)), idx = the lowest j attaining it, and a NaN distance to target 0 sticking
// (the reference's `k == 0 ||` seed).  A uniform grid over the targets does that:
//   1. sdg_bbox      bounding box of the finite targets (per batch item): block minima / maxima merged with integer
//                    atomicMax on an order-preserving encoding, so that consumers read six words;
//   2. sdg_cells     cell id of every point and its rank inside the cell (the value the counting atomicAdd returns);
//   3. sdg_scan      exclusive scan of the cell counts: one launch (every 1024-cell block sums the counts before it),
//                    two on grids of more than 160 blocks;
//   4. sdg_scatter   counting sort without further atomics: point -> start[cell] + rank, stored as float4
//                    {x, y, z, original index};
//   5. sdg_query     per query: seed with target 0 exactly as the reference does, then visit the cube of cells around
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
#include "common.h"
#include "profile.h"
#include "sided_distance_grid.h"
#include "grid_common.h"

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

// one cloud of the pipeline (kernel argument, by value)
struct Cloud {
  int n, G;             // points per batch item; cells per axis of the grid it is binned on
  const float* pts;     // (B, n, 3)
  unsigned int* box;    // (B, 8) encoded {lo[3], hi[3]} of the grid's box, 0 = no finite point yet
  int* count;           // (B, G^3) zero before sdg_cells
  int* start;           // (B, G^3 + 1)
  int2* cellrank;       // (B, n) {cell, rank inside the cell}
  float4* sorted;       // (B, n) {x, y, z, original index} in cell order
};

struct SdgWs {
  Cloud a, b;           // a = p1 (N points), b = p2 (M points)
  int* scan_sums;       // (2, B, blocks of 1024 cells): per-block totals, large grids only
  int scan_blocks;
  size_t zero_bytes;    // prefix of the workspace that must be zeroed (counts + boxes)
  size_t total;
};
// pair = false: sided_distance(p1, p2): both clouds on p2's grid.  pair = true: each cloud on its own grid.
inline SdgWs sdg_layout(void* base, int B, int N, int M, const float* p1, const float* p2, bool pair) {
  SdgWs w;
  w.b.n = M;
  w.b.G = sdg_cells_per_axis(M);
  w.b.pts = p2;
  w.a.n = N;
  w.a.G = pair ? sdg_cells_per_axis(N) : w.b.G;
  w.a.pts = p1;
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
  w.b.box = (unsigned int*)take((size_t)B * 8 * 4);
  w.a.box = pair ? (unsigned int*)take((size_t)B * 8 * 4) : w.b.box;
  w.zero_bytes = off;
  w.b.start = (int*)take((size_t)B * (ncb + 1) * 4);
  w.a.start = (int*)take((size_t)B * (nca + 1) * 4);
  w.b.cellrank = (int2*)take((size_t)B * M * 8);
  w.a.cellrank = (int2*)take((size_t)B * N * 8);
  w.b.sorted = (float4*)take((size_t)B * M * 16);
  w.a.sorted = (float4*)take((size_t)B * N * 16);
  w.scan_blocks = (int)(((nca > ncb ? nca : ncb) + 1023) / 1024);
  w.scan_sums = (int*)take((size_t)2 * B * w.scan_blocks * 4);
  w.total = off;
  return w;
}

// ---- 1. bounding box ---------------------------------------------------------------------------------------------------
// order-preserving float -> uint (negative values reversed below the positives); atomicMax on it is a float max, on
// its complement a float min, and the all-zero word the workspace memset leaves is below every encoded value
__device__ __forceinline__ unsigned int sdg_ord(float v) {
  const unsigned int u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float sdg_unord(unsigned int o) {
  return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// blocks [0, nbx) reduce cloud X into X.box, blocks [nbx, gridDim.x) cloud Y into Y.box (Y.n = 0: one cloud only)
__global__ __launch_bounds__(256) void sdg_bbox_atomic(Cloud X, Cloud Y, int nbx) {
  __shared__ float s[6][256];
  const int b = blockIdx.y;
  const bool first = (int)blockIdx.x < nbx;
  const int n = first ? X.n : Y.n;
  const int blk = first ? blockIdx.x : blockIdx.x - nbx, nblk = first ? nbx : gridDim.x - nbx;
  const float* P = (first ? X.pts : Y.pts) + (size_t)b * n * 3;
  unsigned int* box = (first ? X.box : Y.box) + (size_t)b * 8;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blk * 256 + threadIdx.x; i < n; i += nblk * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = P[(size_t)i * 3 + a];
      if (isfinite(v)) {
        lo[a] = fminf(lo[a], v);
        hi[a] = fmaxf(hi[a], v);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    s[a][threadIdx.x] = lo[a];
    s[3 + a][threadIdx.x] = hi[a];
  }
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (threadIdx.x < d) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        s[a][threadIdx.x] = fminf(s[a][threadIdx.x], s[a][threadIdx.x + d]);
        s[3 + a][threadIdx.x] = fmaxf(s[3 + a][threadIdx.x], s[3 + a][threadIdx.x + d]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3 && s[threadIdx.x][0] <= s[3 + threadIdx.x][0]) {  // this block saw a finite value on the axis
    atomicMax(box + threadIdx.x, ~sdg_ord(s[threadIdx.x][0]));
    atomicMax(box + 3 + threadIdx.x, sdg_ord(s[3 + threadIdx.x][0]));
  }
}

// the grid geometry every consumer derives from the six words (same rules as grid_common.h's sdg_box)
__device__ __forceinline__ Box sdg_box_decode(const unsigned int* __restrict__ w, int G) {
  Box bx;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const unsigned int ulo = w[a], uhi = w[3 + a];
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

// ---- 2. cell id + rank inside the cell (both clouds in one launch) ------------------------------------------------
__global__ __launch_bounds__(256) void sdg_cells(Cloud X, Cloud Y) {
  __shared__ Box s_box;
  const int b = blockIdx.y;
  const int xb = (X.n + 255) / 256;
  const bool first = (int)blockIdx.x < xb;
  const int n = first ? X.n : Y.n, G = first ? X.G : Y.G;
  if (threadIdx.x == 0) s_box = sdg_box_decode((first ? X.box : Y.box) + (size_t)b * 8, G);
  __syncthreads();
  const int i = (first ? blockIdx.x : blockIdx.x - xb) * 256 + threadIdx.x;
  if (i >= n) return;
  const float* P = (first ? X.pts : Y.pts) + ((size_t)b * n + i) * 3;
  const int cx = sdg_axis_cell(P[0], s_box.lo[0], s_box.inv[0], G);
  const int cy = sdg_axis_cell(P[1], s_box.lo[1], s_box.inv[1], G);
  const int cz = sdg_axis_cell(P[2], s_box.lo[2], s_box.inv[2], G);
  const int c = (cz * G + cy) * G + cx;
  const int rank = atomicAdd((first ? X.count : Y.count) + (size_t)b * (G * G * G) + c, 1);
  (first ? X.cellrank : Y.cellrank)[(size_t)b * n + i] = make_int2(c, rank);
}

// ---- 3. exclusive scan of the counts: start[c] = points in cells < c, start[NC] = n ---------------------------------
// a block owns 1024 cells and first needs the number of points before them.  Small grids (<= SDG_SCAN_DIRECT blocks,
// i.e. clouds up to ~330k points): one launch, every block adds up the raw counts before it (block k reads k * 4 KB from
// L2: 5 MB in total at 100k points) -- no second pass, no inter-block dependency.  Larger grids would make that
// quadratic read matter (8 GB at the 128^3 grid), so a first launch leaves per-block totals and blocks add up those.
constexpr int SDG_SCAN_DIRECT = 160;

__global__ __launch_bounds__(1024) void sdg_scan_sums(Cloud X, Cloud Y, int* __restrict__ sums, int nblk) {
  __shared__ int s_wave[16];
  const bool first = blockIdx.z == 0;
  const int G = first ? X.G : Y.G, NC = G * G * G;
  const int base = blockIdx.x * 1024;
  if (base >= NC || (first ? X.n : Y.n) == 0) return;
  const int* cnt = (first ? X.count : Y.count) + (size_t)blockIdx.y * NC;
  const int i = base + threadIdx.x;
  const int tot = sdg_block_inclusive(i < NC ? cnt[i] : 0, s_wave);
  if (threadIdx.x == 1023) sums[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * nblk + blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void sdg_scan(Cloud X, Cloud Y, const int* __restrict__ sums, int nblk) {
  __shared__ int s_wave[16];
  __shared__ int s_off;
  const bool first = blockIdx.z == 0;
  const int G = first ? X.G : Y.G, NC = G * G * G;
  const int base = blockIdx.x * 1024;
  if (base >= NC || (first ? X.n : Y.n) == 0) return;
  const int* cnt = (first ? X.count : Y.count) + (size_t)blockIdx.y * NC;
  int* start = (first ? X.start : Y.start) + (size_t)blockIdx.y * (NC + 1);
  int part = 0;
  if (sums != nullptr) {
    const int* my = sums + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * nblk;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) part += my[k];
  } else {
    for (int k = threadIdx.x; k < base; k += 1024) part += cnt[k];
  }
  const int before = sdg_block_inclusive(part, s_wave);
  if (threadIdx.x == 1023) s_off = before;
  __syncthreads();
  const int off = s_off;
  __syncthreads();
  const int i = base + threadIdx.x;
  const int v = i < NC ? cnt[i] : 0;
  const int inc = sdg_block_inclusive(v, s_wave);
  if (i < NC) start[i] = off + inc - v;
  if (i == NC - 1) start[NC] = off + inc;
}

// ---- 4. counting-sort scatter (both clouds in one launch) -------------------------------------------------------------
__global__ __launch_bounds__(256) void sdg_scatter(Cloud X, Cloud Y) {
  const int b = blockIdx.y;
  const int xb = (X.n + 255) / 256;
  const bool first = (int)blockIdx.x < xb;
  const int n = first ? X.n : Y.n, G = first ? X.G : Y.G;
  const int i = (first ? blockIdx.x : blockIdx.x - xb) * 256 + threadIdx.x;
  if (i >= n) return;
  const int2 cr = (first ? X.cellrank : Y.cellrank)[(size_t)b * n + i];
  const int pos = (first ? X.start : Y.start)[(size_t)b * (G * G * G + 1) + cr.x] + cr.y;
  const float* P = (first ? X.pts : Y.pts) + ((size_t)b * n + i) * 3;
  (first ? X.sorted : Y.sorted)[(size_t)b * n + pos] = make_float4(P[0], P[1], P[2], __int_as_float(i));
}

// ---- 5. query ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sdg_dist(float tx, float ty, float tz, float qx, float qy, float qz) {
  const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

constexpr int SDG_GROUP = 8;  // lanes cooperating on one query (rows of the cell cube are dealt round-robin)

// the search for one query, shared by the one-direction and the two-direction kernels.  All SDG_GROUP lanes of a query
// call it with the same (qx, qy, qz, c); on return every lane holds the query's (best, best_i).
__device__ __forceinline__ void sdg_search(const Box& s_box, int G, float qx, float qy, float qz, int c,
                                           const float* __restrict__ T0, const int* __restrict__ start,
                                           const float4* __restrict__ TS, int sub, float& best, int& best_i) {
  // the reference's seed: target 0 unconditionally (a NaN distance sticks)
  best = sdg_dist(T0[0], T0[1], T0[2], qx, qy, qz);
  best_i = 0;
  if (best == best) {  // uniform within the group (same query)
    const int cx = c % G, cy = (c / G) % G, cz = c / (G * G);
    // rounding head-room of the geometric bound: cell membership is decided by a rounded (v - lo) * inv
    const float q[3] = {qx, qy, qz};
    float slack[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) slack[a] = 4e-6f * (fabsf(q[a]) + fabsf(s_box.lo[a]) + s_box.size[a] * (float)G);
    const int cq[3] = {cx, cy, cz};
    for (int r = 0; r < G; ++r) {
      const int z0 = max(cz - r, 0), z1 = min(cz + r, G - 1);
      const int y0 = max(cy - r, 0), y1 = min(cy + r, G - 1);
      const int x0 = max(cx - r, 0), x1 = min(cx + r, G - 1);
      const int ny = y1 - y0 + 1, nrows = (z1 - z0 + 1) * ny;
      for (int j = sub; j < nrows; j += SDG_GROUP) {
        const int z = z0 + j / ny, y = y0 + j % ny;
        const int row = (z * G + y) * G;
        const bool shell_row = (abs(z - cz) == r) || (abs(y - cy) == r);
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
            const float4 t = TS[k];
            const float d = sdg_dist(t.x, t.y, t.z, qx, qy, qz);
            const int ti = __float_as_int(t.w);
            if (d < best || (d == best && ti < best_i)) {
              best = d;
              best_i = ti;
            }
          }
      }
#pragma unroll
      for (int m = 1; m < SDG_GROUP; m <<= 1) {  // lexicographic (dist, idx) minimum over the group
        const float od = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(best_i, m, 64);
        if (od < best || (od == best && oi < best_i)) {
          best = od;
          best_i = oi;
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
      if (bound > 0.f && best < bound * bound * 0.99999f) break;
    }
  }
}


// blockIdx.z = direction: 0 answers the queries A against the targets T (dist1 / idx1), 1 the reverse (dist2 / idx2).
// The search runs on the TARGETS' grid; the queries only have to arrive in a spatially coherent order, which their own
// sorted copy provides whichever grid it was sorted on.
__global__ __launch_bounds__(256) void sdg_query(Cloud A, Cloud T, float* __restrict__ dist1, int64_t* __restrict__ idx1,
                                                 float* __restrict__ dist2, int64_t* __restrict__ idx2) {
  __shared__ Box s_box;
  const int b = blockIdx.y;
  const bool fwd = blockIdx.z == 0;
  const int nq = fwd ? A.n : T.n, nt = fwd ? T.n : A.n, G = fwd ? T.G : A.G;
  if ((long long)blockIdx.x * 256 >= (long long)nq * SDG_GROUP) return;  // the launch is sized for the larger cloud
  if (threadIdx.x == 0) s_box = sdg_box_decode((fwd ? T.box : A.box) + (size_t)b * 8, G);
  __syncthreads();
  // 100k queries are only ~1.5 wavefronts per SIMD and every query is a chain of dependent loads (cell range ->
  // targets): 8 lanes share a query so that 8x more loads are in flight; the lanes' (dist, idx) are merged with a
  // 3-step butterfly after every ring
  const int sub = threadIdx.x % SDG_GROUP;
  const int slot = (blockIdx.x * 256 + threadIdx.x) / SDG_GROUP;
  const bool live = slot < nq;
  const int NC = G * G * G;
  const float4 q = (fwd ? A.sorted : T.sorted)[(size_t)b * nq + (live ? slot : 0)];
  const int cx = sdg_axis_cell(q.x, s_box.lo[0], s_box.inv[0], G);
  const int cy = sdg_axis_cell(q.y, s_box.lo[1], s_box.inv[1], G);
  const int cz = sdg_axis_cell(q.z, s_box.lo[2], s_box.inv[2], G);
  float best;
  int best_i;
  sdg_search(s_box, G, q.x, q.y, q.z, (cz * G + cy) * G + cx, (fwd ? T.pts : A.pts) + (size_t)b * nt * 3,
             (fwd ? T.start : A.start) + (size_t)b * (NC + 1), (fwd ? T.sorted : A.sorted) + (size_t)b * nt, sub, best,
             best_i);
  if (live && sub == 0) {
    const size_t o = (size_t)b * nq + __float_as_int(q.w);
    (fwd ? dist1 : dist2)[o] = best;
    (fwd ? idx1 : idx2)[o] = best_i;
  }
}

int sdg_run(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist1, int64_t* idx1,
            float* dist2, int64_t* idx2, void* workspace, bool pair) {
  const SdgWs w = sdg_layout(workspace, B, N, M, p1, p2, pair);
  const Cloud none = {0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  auto blocks = [](int n) { return kamd_cdiv(n, 2048) < 128 ? kamd_cdiv(n, 2048) : 128; };
  KAMD_CHECK(kamd_zero_async(workspace, w.zero_bytes, st));
  {
    ProfScope p(K_SDG_BUILD, st);
    if (pair)
      hipLaunchKernelGGL(sdg_bbox_atomic, dim3(blocks(M) + blocks(N), B), dim3(256), 0, st, w.b, w.a, blocks(M));
    else  // the queries are binned on the targets' box
      hipLaunchKernelGGL(sdg_bbox_atomic, dim3(blocks(M), B), dim3(256), 0, st, w.b, none, blocks(M));
    hipLaunchKernelGGL(sdg_cells, dim3(kamd_cdiv(M, 256) + kamd_cdiv(N, 256), B), dim3(256), 0, st, w.b, w.a);
    const dim3 scan_grid(w.scan_blocks, B, 2);
    const int* sums = nullptr;
    if (w.scan_blocks > SDG_SCAN_DIRECT) {
      hipLaunchKernelGGL(sdg_scan_sums, scan_grid, dim3(1024), 0, st, w.b, w.a, w.scan_sums, w.scan_blocks);
      sums = w.scan_sums;
    }
    hipLaunchKernelGGL(sdg_scan, scan_grid, dim3(1024), 0, st, w.b, w.a, sums, w.scan_blocks);
    hipLaunchKernelGGL(sdg_scatter, dim3(kamd_cdiv(M, 256) + kamd_cdiv(N, 256), B), dim3(256), 0, st, w.b, w.a);
  }
  KAMD_CHECK(hipGetLastError());
  {
    ProfScope p(K_SDG_QUERY, st);
    const int big = (pair && M > N) ? M : N;
    hipLaunchKernelGGL(sdg_query, dim3(kamd_cdiv((long long)big * SDG_GROUP, 256), B, pair ? 2 : 1), dim3(256), 0, st, w.a,
                       w.b, dist1, idx1, dist2, idx2);
  }
  KAMD_RETURN_LAST_ERROR();
}

}  // namespace

bool sdgrid_applicable(int B, int N, int M) {
  // below this the brute-force kernels are as fast as the launches of the grid pipeline
  return B >= 1 && B <= 65535 && M >= 8192 && N >= 2048 && (long long)B * (long long)(M > N ? M : N) < (1ll << 30);
}
size_t sdgrid_workspace_bytes(int B, int N, int M) { return sdg_layout(nullptr, B, N, M, nullptr, nullptr, false).total; }

int sdgrid_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist, int64_t* idx,
                       void* workspace) {
  return sdg_run(st, B, N, M, p1, p2, dist, idx, nullptr, nullptr, workspace, false);
}

bool sdgrid_pair_applicable(int B, int N, int M) { return sdgrid_applicable(B, N, M) && sdgrid_applicable(B, M, N); }
size_t sdgrid_pair_workspace_bytes(int B, int N, int M) {
  return sdg_layout(nullptr, B, N, M, nullptr, nullptr, true).total;
}
int sdgrid_pair_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist1,
                            int64_t* idx1, float* dist2, int64_t* idx2, void* workspace) {
  return sdg_run(st, B, N, M, p1, p2, dist1, idx1, dist2, idx2, workspace, true);
}

}  // namespace kamd
