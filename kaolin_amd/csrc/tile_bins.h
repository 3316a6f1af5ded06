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

// row i of a 64 x 64 bit matrix in lane i  ->  column j in lane j.
// Six butterfly stages: stage S swaps lane bit S with word bit S -- lanes l and l ^ S exchange the halves of every 2S-bit group.
// Round 4: no LDS crossbar (the six __shfl_xor of a 64-bit word were twelve ds_bpermute, each a dependent ~100-cycle round trip,
// with divergent selects between them: the per-chunk latency of the soft mask's select kernel).  Stage 32 is ONE
// v_permlane32_swap of the word's two registers (gfx950); stage 16 a v_permlane16_swap per register + a byte permute; stages
// 8 .. 1 a DPP move per register (row_ror:8, row_shl / row_shr:4 by bank, quad_perm) and a rotate + bit-field insert whose
// rotate amount and mask depend on the lane's side of the exchange.  39 vector instructions, no memory operation.
__device__ __forceinline__ unsigned int t64_stage16(unsigned int x, bool upper) {
  // y = the word of lane ^ 16: the odd rows of 16 lanes receive it in the first result, the even rows in the second
#if __has_builtin(__builtin_amdgcn_permlane16_swap)
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const unsigned int y = upper ? r[0] : r[1];
#else
  const unsigned int y = (unsigned int)__shfl_xor((int)x, 16, 64);
#endif
  // lower side: {x.b0, x.b1, y.b0, y.b1}; upper side: {y.b2, y.b3, x.b2, x.b3}   (v_perm_b32: selector 0-3 = bytes of the 2nd operand, 4-7 = of the 1st)
  return __builtin_amdgcn_perm(y, x, upper ? 0x03020706u : 0x05040100u);
}
__device__ __forceinline__ unsigned int t64_stage8(unsigned int x, bool upper) {
  const unsigned int y = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);  // row_ror:8 = lane ^ 8
  // lower side: {x.b0, y.b0, x.b2, y.b2}; upper side: {y.b1, x.b1, y.b3, x.b3}
  return __builtin_amdgcn_perm(y, x, upper ? 0x03070105u : 0x06020400u);
}
// stages 4, 2, 1: groups of 2 * S bits, `m` = the low half of every group, y = the word of lane ^ S
__device__ __forceinline__ unsigned int t64_merge(unsigned int x, unsigned int y, unsigned int m, int S, bool upper) {
  // lower side keeps its low halves and takes the partner's low halves into the high halves: (x & m) | ((y & m) << S);
  // upper side: (x & ~m) | ((y & ~m) >> S).  A rotate serves both shifts: the bits that wrap land where the mask discards them.
  const unsigned int rot = __builtin_amdgcn_alignbit(y, y, upper ? (unsigned int)S : (unsigned int)(32 - S));
  const unsigned int keep = upper ? ~m : m;
  return (keep & x) | (~keep & rot);
}
__device__ __forceinline__ unsigned int t64_stage4(unsigned int x, bool upper) {
  // lane ^ 4: the banks (groups of 4 lanes) 0 and 2 of a row read 4 lanes up (row_shl:4), the banks 1 and 3 read 4 lanes down
  int y = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0x5, false);
  y = __builtin_amdgcn_update_dpp(y, (int)x, 0x114, 0xF, 0xA, false);
  return t64_merge(x, (unsigned int)y, 0x0F0F0F0Fu, 4, upper);
}
__device__ __forceinline__ unsigned int t64_stage2(unsigned int x, bool upper) {
  const unsigned int y = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);  // quad_perm [2, 3, 0, 1] = lane ^ 2
  return t64_merge(x, y, 0x33333333u, 2, upper);
}
__device__ __forceinline__ unsigned int t64_stage1(unsigned int x, bool upper) {
  const unsigned int y = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2] = lane ^ 1
  return t64_merge(x, y, 0x55555555u, 1, upper);
}
__device__ __forceinline__ unsigned long long wave_transpose64(unsigned long long x) {
  const int lane = threadIdx.x & 63;
  unsigned int lo = (unsigned int)x, hi = (unsigned int)(x >> 32);
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  {
    // lanes 32..63 of the first operand <-> lanes 0..31 of the second: the lower lanes' high word becomes the partner's low word,
    // the upper lanes' low word the partner's high word -- the whole stage
    const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
    lo = r[0];
    hi = r[1];
  }
#else
  {
    const unsigned long long y = __shfl_xor(x, 32, 64);
    const unsigned long long z = (lane & 32) ? ((x & 0xFFFFFFFF00000000ull) | (y >> 32)) : ((x & 0xFFFFFFFFull) | (y << 32));
    lo = (unsigned int)z;
    hi = (unsigned int)(z >> 32);
  }
#endif
  const bool u16 = (lane & 16) != 0, u8 = (lane & 8) != 0, u4 = (lane & 4) != 0, u2 = (lane & 2) != 0, u1 = (lane & 1) != 0;
  lo = t64_stage16(lo, u16);
  hi = t64_stage16(hi, u16);
  lo = t64_stage8(lo, u8);
  hi = t64_stage8(hi, u8);
  lo = t64_stage4(lo, u4);
  hi = t64_stage4(hi, u4);
  lo = t64_stage2(lo, u2);
  hi = t64_stage2(hi, u2);
  lo = t64_stage1(lo, u1);
  hi = t64_stage1(hi, u1);
  return ((unsigned long long)hi << 32) | lo;
}
// the same through __shfl_xor (the LDS crossbar): the definition the fast form is tested against (kamd_debug_transpose64)
__device__ __forceinline__ unsigned long long wave_transpose64_reference(unsigned long long x) {
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

// ---- inclusive scan over a wavefront without the LDS crossbar: DPP row shifts inside the four rows of 16 lanes (a lane without a
// source reads 0), then the row totals broadcast with v_readlane
__device__ __forceinline__ int wave_inclusive_scan_dpp(int v) {
  const int lane = threadIdx.x & 63;
  v += row_shr<1>(v);
  v += row_shr<2>(v);
  v += row_shr<4>(v);
  v += row_shr<8>(v);
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
  return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
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
