// Pieces of the uniform-grid pipeline shared by the exact nearest-point (sided_distance_grid.hip) and
// nearest-triangle (triangle_distance.hip) searches: bounding box of a packed xyz array in partials, the grid geometry
// every consumer re-derives from them, the cell index of a coordinate, and a workgroup scan.
#pragma once
#include "common.h"

namespace kamd {
namespace {

constexpr int SDG_MAXG = 128;      // cells per axis at most
constexpr int SDG_NB = 64;         // bbox partial blocks per batch item

// ---- 1. bounding box partials ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sdg_bbox(int M, const float* __restrict__ p2, float* __restrict__ part) {
  __shared__ float s[6][256];
  const int b = blockIdx.y;
  const float* P = p2 + (size_t)b * M * 3;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
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
  if (threadIdx.x < 6) part[((size_t)b * SDG_NB + blockIdx.x) * 6 + threadIdx.x] = s[threadIdx.x][0];
}

struct Box {
  float lo[3], size[3], inv[3];  // origin, cell size, 1 / cell size per axis
};
// every consumer re-reduces the <= 64 partials (a few hundred bytes from L2) into the grid geometry
__device__ __forceinline__ Box sdg_box(const float* __restrict__ part, int b, int nb, int G) {
  Box bx;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float lo = INFINITY, hi = -INFINITY;
    for (int k = 0; k < nb; ++k) {
      lo = fminf(lo, part[((size_t)b * SDG_NB + k) * 6 + a]);
      hi = fmaxf(hi, part[((size_t)b * SDG_NB + k) * 6 + 3 + a]);
    }
    if (!(lo <= hi)) {  // no finite target on this axis
      lo = 0.f;
      hi = 0.f;
    }
    float size = (hi - lo) / (float)G;
    if (!(size > 0.f) || !isfinite(size)) size = 1.f;  // degenerate extent: a single slab holds everything
    bx.lo[a] = lo;
    bx.size[a] = size;
    bx.inv[a] = 1.f / size;
  }
  return bx;
}
__device__ __forceinline__ int sdg_axis_cell(float v, float lo, float inv, int G) {
  const float t = (v - lo) * inv;
  int c = (t >= 0.f) ? (t < (float)G ? (int)t : G - 1) : 0;  // NaN -> 0, +-inf -> clamped
  return c;
}

// ---- inclusive scan over a 1024-thread workgroup (the last thread's result is the total) ----------------------------
__device__ __forceinline__ int sdg_block_inclusive(int v, int* s_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += s_wave[w];
  return woff + inc;
}
}  // namespace
}  // namespace kamd
