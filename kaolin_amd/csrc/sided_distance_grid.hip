// sided_distance forward, exact uniform-grid search (fp32) for MI355X (gfx950).
//
// The reference (kaolin/csrc/metrics/sided_distance_cuda.cu:52-201) is an all-pairs search: N*M distance
// evaluations against 24*(N+M) bytes of input, 8 300 FLOP/B at 100k x 100k -- VALU-bound by construction
// (sd_main_f32 already issues at 97 % of the measured FMA peak).  The only way to go faster is to evaluate fewer
// pairs while returning the SAME answer: dist = min_j d(p1_i, p2_j) with the identical fp32 expression
// d = fma(dz,dz, fma(dy,dy, dx*dx)), idx = the lowest j attaining it, and a NaN distance to target 0 sticking
// (the reference's `k == 0 ||` seed).  A uniform grid over the targets does that:
//   1. sdg_bbox      bounding box of the finite targets (per batch item), in partials;
//   2. sdg_cells     cell id of every target and every query; per-cell counts (atomicAdd); one launch for both;
//   3. sdg_scan      exclusive scan of the counts (one workgroup per batch item and array);
//   4. sdg_scatter   counting-sort scatter: targets as float4 {x, y, z, original index}; queries as an index list in
//                    cell order, so that the 64 queries of a wavefront are spatial neighbours (same cells, same lines);
//   5. sdg_query     per query: seed with target 0 exactly as the reference does, then visit the cube of cells around
//                    the query ring by ring; after each ring every unvisited target is provably farther than the
//                    distance from the query to the cube's faces (minus a rounding margin), so the search stops as soon
//                    as the best distance is below that bound.  Ties are resolved towards the lower original index
//                    explicitly, so the arbitrary order inside a cell does not matter.
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

struct SdgGeom {
  int G, NC;
};
inline SdgGeom sdg_geom(int M) {
  // ~2 targets per cell on average for a volume-filling cloud; surfaces leave most cells empty, which is fine
  int G = (int)floor(cbrt((double)M / 2.0) + 0.5);
  if (G < 1) G = 1;
  if (G > SDG_MAXG) G = SDG_MAXG;
  return SdgGeom{G, G * G * G};
}
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct SdgWs {
  float* bbox_part;   // B * SDG_NB * 6
  int* t_count;       // B * (NC + 1)   counts -> starts (after the scan)
  int* q_count;       // B * (NC + 1)
  int* t_fill;        // B * NC
  int* q_fill;        // B * NC
  int* t_cell;        // B * M
  int* q_cell;        // B * N
  float4* t_sorted;   // B * M
  int* q_sorted;      // B * N
  int* scan_sums;     // 2 * B * ceil(NC / 1024)
  size_t zero_bytes;  // prefix of the workspace that must be zeroed (counts + fills)
  size_t total;
};
inline SdgWs sdg_layout(void* base, int B, int N, int M) {
  const SdgGeom g = sdg_geom(M);
  SdgWs w;
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p + off;
    off += al(bytes);
    return r;
  };
  w.t_count = (int*)take((size_t)B * (g.NC + 1) * 4);
  w.q_count = (int*)take((size_t)B * (g.NC + 1) * 4);
  w.t_fill = (int*)take((size_t)B * g.NC * 4);
  w.q_fill = (int*)take((size_t)B * g.NC * 4);
  w.zero_bytes = off;
  w.bbox_part = (float*)take((size_t)B * SDG_NB * 6 * 4);
  w.t_cell = (int*)take((size_t)B * M * 4);
  w.q_cell = (int*)take((size_t)B * N * 4);
  w.t_sorted = (float4*)take((size_t)B * M * 16);
  w.q_sorted = (int*)take((size_t)B * N * 4);
  w.scan_sums = (int*)take((size_t)2 * B * ((g.NC + 1023) / 1024) * 4);
  w.total = off;
  return w;
}

// ---- 2. cell ids + counts (targets and queries in one launch) ------------------------------------------------
__global__ __launch_bounds__(256) void sdg_cells(int M, int N, int nb, int G, const float* __restrict__ p2,
                                                 const float* __restrict__ p1, const float* __restrict__ part,
                                                 int* __restrict__ t_cell, int* __restrict__ q_cell,
                                                 int* __restrict__ t_count, int* __restrict__ q_count) {
  __shared__ Box s_box;
  const int b = blockIdx.y;
  if (threadIdx.x == 0) s_box = sdg_box(part, b, nb, G);
  __syncthreads();
  const int mb = (M + 255) / 256;
  const bool is_t = (int)blockIdx.x < mb;
  const int n = is_t ? M : N;
  const int i = (is_t ? blockIdx.x : blockIdx.x - mb) * 256 + threadIdx.x;
  if (i >= n) return;
  const float* P = (is_t ? p2 : p1) + ((size_t)b * n + i) * 3;
  const int cx = sdg_axis_cell(P[0], s_box.lo[0], s_box.inv[0], G);
  const int cy = sdg_axis_cell(P[1], s_box.lo[1], s_box.inv[1], G);
  const int cz = sdg_axis_cell(P[2], s_box.lo[2], s_box.inv[2], G);
  const int c = (cz * G + cy) * G + cx;
  (is_t ? t_cell : q_cell)[(size_t)b * n + i] = c;
  atomicAdd((is_t ? t_count : q_count) + (size_t)b * (G * G * G + 1) + c, 1);
}

// ---- 4. counting-sort scatter (targets and queries in one launch) --------------------------------------------------
__global__ __launch_bounds__(256) void sdg_scatter(int M, int N, int NC, const float* __restrict__ p2,
                                                   const int* __restrict__ t_cell, const int* __restrict__ q_cell,
                                                   const int* __restrict__ t_start, const int* __restrict__ q_start,
                                                   int* __restrict__ t_fill, int* __restrict__ q_fill,
                                                   float4* __restrict__ t_sorted, int* __restrict__ q_sorted) {
  const int b = blockIdx.y;
  const int mb = (M + 255) / 256;
  if ((int)blockIdx.x < mb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int c = t_cell[(size_t)b * M + i];
    const int pos = t_start[(size_t)b * (NC + 1) + c] + atomicAdd(t_fill + (size_t)b * NC + c, 1);
    const float* P = p2 + ((size_t)b * M + i) * 3;
    t_sorted[(size_t)b * M + pos] = make_float4(P[0], P[1], P[2], __int_as_float(i));
  } else {
    const int i = (blockIdx.x - mb) * 256 + threadIdx.x;
    if (i >= N) return;
    const int c = q_cell[(size_t)b * N + i];
    const int pos = q_start[(size_t)b * (NC + 1) + c] + atomicAdd(q_fill + (size_t)b * NC + c, 1);
    q_sorted[(size_t)b * N + pos] = i;
  }
}

// ---- 5. query ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sdg_dist(float tx, float ty, float tz, float qx, float qy, float qz) {
  const float dx = tx - qx, dy = ty - qy, dz = tz - qz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

constexpr int SDG_GROUP = 8;  // lanes cooperating on one query (rows of the cell cube are dealt round-robin)

__global__ __launch_bounds__(256) void sdg_query(int N, int M, int nb, int G, const float* __restrict__ p1,
                                                 const float* __restrict__ p2, const float* __restrict__ part,
                                                 const int* __restrict__ q_sorted, const int* __restrict__ q_cell,
                                                 const int* __restrict__ t_start, const float4* __restrict__ t_sorted,
                                                 float* __restrict__ dist, int64_t* __restrict__ idx) {
  __shared__ Box s_box;
  const int b = blockIdx.y;
  if (threadIdx.x == 0) s_box = sdg_box(part, b, nb, G);
  __syncthreads();
  // 100k queries are only ~1.5 wavefronts per SIMD and every query is a chain of dependent loads (cell range ->
  // targets): 8 lanes share a query so that 8x more loads are in flight; the lanes' (dist, idx) are merged with a
  // 3-step butterfly after every ring
  const int sub = threadIdx.x % SDG_GROUP;
  const int slot = (blockIdx.x * 256 + threadIdx.x) / SDG_GROUP;
  const bool live = slot < N;
  const int NC = G * G * G;
  const int qi = live ? q_sorted[(size_t)b * N + slot] : 0;
  const float* Q = p1 + ((size_t)b * N + qi) * 3;
  const float qx = Q[0], qy = Q[1], qz = Q[2];
  const float* T0 = p2 + (size_t)b * M * 3;
  // the reference's seed: target 0 unconditionally (a NaN distance sticks)
  float best = sdg_dist(T0[0], T0[1], T0[2], qx, qy, qz);
  int best_i = 0;
  if (best == best) {  // uniform within the group (same query)
    const int c = q_cell[(size_t)b * N + qi];
    const int cx = c % G, cy = (c / G) % G, cz = c / (G * G);
    const int* start = t_start + (size_t)b * (NC + 1);
    const float4* TS = t_sorted + (size_t)b * M;
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
  if (live && sub == 0) {
    dist[(size_t)b * N + qi] = best;
    idx[(size_t)b * N + qi] = best_i;
  }
}

}  // namespace

bool sdgrid_applicable(int B, int N, int M) {
  // below this the brute-force kernels are as fast as the seven launches of the grid pipeline
  return B >= 1 && M >= 8192 && N >= 2048 && (long long)B * (long long)(M > N ? M : N) < (1ll << 30);
}
size_t sdgrid_workspace_bytes(int B, int N, int M) { return sdg_layout(nullptr, B, N, M).total; }

int sdgrid_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist, int64_t* idx,
                       void* workspace) {
  const SdgGeom g = sdg_geom(M);
  const SdgWs w = sdg_layout(workspace, B, N, M);
  const int nb = kamd_cdiv(M, 4096) < SDG_NB ? kamd_cdiv(M, 4096) : SDG_NB;
  KAMD_CHECK(hipMemsetAsync(workspace, 0, w.zero_bytes, st));
  {
    ProfScope p(K_SDG_BUILD, st);
    hipLaunchKernelGGL(sdg_bbox, dim3(nb, B), dim3(256), 0, st, M, p2, w.bbox_part);
    hipLaunchKernelGGL(sdg_cells, dim3(kamd_cdiv(M, 256) + kamd_cdiv(N, 256), B), dim3(256), 0, st, M, N, nb, g.G, p2, p1,
                       w.bbox_part, w.t_cell, w.q_cell, w.t_count, w.q_count);
    const int nblk = kamd_cdiv(g.NC, 1024);
    hipLaunchKernelGGL(sdg_scan_sums, dim3(nblk, B, 2), dim3(1024), 0, st, g.NC, nblk, w.t_count, w.q_count, w.scan_sums);
    hipLaunchKernelGGL(sdg_scan_apply, dim3(nblk, B, 2), dim3(1024), 0, st, g.NC, nblk, w.t_count, w.q_count, w.scan_sums);
    hipLaunchKernelGGL(sdg_scatter, dim3(kamd_cdiv(M, 256) + kamd_cdiv(N, 256), B), dim3(256), 0, st, M, N, g.NC, p2,
                       w.t_cell, w.q_cell, w.t_count, w.q_count, w.t_fill, w.q_fill, w.t_sorted, w.q_sorted);
  }
  KAMD_CHECK(hipGetLastError());
  {
    ProfScope p(K_SDG_QUERY, st);
    hipLaunchKernelGGL(sdg_query, dim3(kamd_cdiv((long long)N * SDG_GROUP, 256), B), dim3(256), 0, st, N, M, nb, g.G, p1, p2, w.bbox_part, w.q_sorted,
                       w.q_cell, w.t_count, w.t_sorted, dist, idx);
  }
  KAMD_RETURN_LAST_ERROR();
}

}  // namespace kamd
