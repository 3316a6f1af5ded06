// Small pieces shared by the DIB-R kernels (gfx950): pixel centres, wide box / record types, register broadcasts, DPP row
// shifts, wave scans and the 64 x 64 bit transpose.  (The round-1 binning that lived here -- one bit per (32 x 32 tile,
// face), cleared and scanned on every call -- was replaced by the tile face lists of tile_lists.h.)
#pragma once
#include "common.h"

namespace kamd {

constexpr int SUB_W = 16, SUB_H = 4;     // pixels per wavefront sub-tile (64 lanes)

// four scalars read with ONE wide LDS / global load (a short-circuit chain of compares on four separately indexed
// scalars compiles to dependent ds_read_b32 + branch pairs: measured ~300 cycles per box in the soft-mask search)
template <typename T>
struct alignas(16) Box4 {
  T x0, y0, x1, y1;  // xmin, ymin, xmax, ymax
};
// the reference's reject test `x < xmin || x >= xmax || y < ymin || y >= ymax`, branch-free (NaN limits never reject)
template <typename T>
__device__ __forceinline__ bool box_rejects(const Box4<T>& b, T x, T y) {
  return (x < b.x0) | (x >= b.x1) | (y < b.y0) | (y >= b.y1);
}

// value of lane j (wave-uniform j) in every lane: v_readlane, no LDS crossbar
__device__ __forceinline__ float wave_bcast_f(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
template <typename T>
__device__ __forceinline__ T wave_bcast(T v, int j);
template <>
__device__ __forceinline__ float wave_bcast<float>(float v, int j) { return wave_bcast_f(v, j); }
template <>
__device__ __forceinline__ double wave_bcast<double>(double v, int j) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j));
}

// pixel centre in the reference's float arithmetic (rasterization_cuda.cu:85-86, dibr_soft_mask_cuda.cu:75-76):
//   x0 = multiplier / width * (2*col + 1 - width),  y0 = multiplier / height * (height - 2*row - 1)
__device__ __forceinline__ float pixel_x(float multiplier, int W, int col) { return multiplier / W * (2 * col + 1 - W); }
__device__ __forceinline__ float pixel_y(float multiplier, int H, int row) { return multiplier / H * (H - 2 * row - 1); }
// the same with the quotients `multiplier / W`, `multiplier / H` (one IEEE float division each, identical on host and
// device) taken once per launch instead of once per evaluation
struct PixelScale {
  float mw, mh;
  int W, H;
};
inline PixelScale pixel_scale(float multiplier, int H, int W) { return PixelScale{multiplier / W, multiplier / H, W, H}; }
__device__ __forceinline__ float pixel_x(const PixelScale& ps, int col) { return ps.mw * (2 * col + 1 - ps.W); }
__device__ __forceinline__ float pixel_y(const PixelScale& ps, int row) { return ps.mh * (ps.H - 2 * row - 1); }

// row i of a 64 x 64 bit matrix in lane i  ->  column j in lane j
__device__ __forceinline__ unsigned long long wave_transpose64(unsigned long long x) {
  const int lane = threadIdx.x & 63;
#define KAMD_T64_STAGE(S, M)                                                        \
  {                                                                                 \
    const unsigned long long y = __shfl_xor(x, S, 64);                              \
    x = (lane & S) ? ((x & ~(M)) | ((y & ~(M)) >> S)) : ((x & (M)) | ((y & (M)) << S)); \
  }
  KAMD_T64_STAGE(32, 0x00000000FFFFFFFFull)
  KAMD_T64_STAGE(16, 0x0000FFFF0000FFFFull)
  KAMD_T64_STAGE(8, 0x00FF00FF00FF00FFull)
  KAMD_T64_STAGE(4, 0x0F0F0F0F0F0F0F0Full)
  KAMD_T64_STAGE(2, 0x3333333333333333ull)
  KAMD_T64_STAGE(1, 0x5555555555555555ull)
#undef KAMD_T64_STAGE
  return x;
}

// element strides of the optional per-face inputs of the fused front doors (dense: z_face 3, z_vertex 1, front_stride 1)
struct FaceLayout {
  long long z_face, z_vertex, front_stride;
};

template <typename T>
struct alignas(16) Rec4 {
  T a, b, c, d;
};

// value of the lane N places to the left / right inside the same row of 16 lanes (DPP row_shr / row_shl: a register
// move, no LDS crossbar); lanes without a source read 0
template <int N>
__device__ __forceinline__ int row_shr(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xF, 0xF, true); }
template <int N>
__device__ __forceinline__ int row_shl(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true); }
template <int N>
__device__ __forceinline__ float row_shr(float v) { return __int_as_float(row_shr<N>(__float_as_int(v))); }
template <int N>
__device__ __forceinline__ double row_shr(double v) {
  return __hiloint2double(row_shr<N>(__double2hiint(v)), row_shr<N>(__double2loint(v)));
}

// ---- inclusive scan over a wavefront ----------------------------------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
}  // namespace kamd
