// Screen-tile face binning shared by the DIB-R rasterizer and the soft-mask kernels (gfx950).
//
// The reference tests every pixel against every face (rasterization_cuda.cu:88-170,
// dibr_soft_mask_cuda.cu:80-172).  Both kernels only ever let a face act on a pixel whose centre lies
// inside the face's (possibly enlarged) bounding box, and both visit faces in ascending index.  We keep
// exactly that semantics but make the search sub-linear:
//
//   1. bin kernel: one thread per face sets the face's bit in the bitmask of every 32x32-pixel tile its
//      bbox can touch (conservative: +-1 pixel, NaN boxes go everywhere).  A bitmask -- not an append
//      list -- so that "ascending face index" is free and the workspace size does not depend on the data.
//   2. tile kernel: one 1024-thread workgroup per tile expands the tile's bitmask (popcount + block scan)
//      into an ascending face list, stages the face records in LDS, and each of its 16 wavefronts culls
//      the list against its own 16x4-pixel sub-tile with one ballot per 64 faces before any per-pixel
//      work happens.
//
// Workspace layout (all offsets 256-B aligned):
//   records : total_faces x 16 x sizeof(T)   {bbox[4], a.xy, b.xy, c.xy, z[3], pad[3]}
//   masks   : ntiles x (total_faces/32 + B + 1) 32-bit words; mesh b owns the word range
//             ntiles*(first[b]/32 + b) .. , tile t of mesh b starts at  + t*stride_b,
//             stride_b = ceil(n_b/32)  (regions provably do not overlap);
//             then B x ntiles flag words (tile touched by any face of the mesh).
#pragma once
#include "common.h"

namespace kamd {

constexpr int TILE_W = 32, TILE_H = 32;  // pixels per workgroup tile
constexpr int SUB_W = 16, SUB_H = 4;     // pixels per wavefront sub-tile (64 lanes)
constexpr int TILE_THREADS = 1024;       // 16 wavefronts: 2 x 8 sub-tiles
constexpr int REC_STRIDE = 16;           // scalars per face record

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

struct TileGeom {
  int H, W, tiles_x, tiles_y, ntiles;
};
__host__ __device__ inline TileGeom tile_geom(int H, int W) {
  TileGeom g;
  g.H = H;
  g.W = W;
  g.tiles_x = (W + TILE_W - 1) / TILE_W;
  g.tiles_y = (H + TILE_H - 1) / TILE_H;
  g.ntiles = g.tiles_x * g.tiles_y;
  return g;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline size_t mask_words(int ntiles, int B, long long total_faces) {
  return (size_t)ntiles * (size_t)(total_faces / 32 + B + 1);
}
// the mask area is followed by one word per (mesh, tile): non-zero when any face touches the tile
inline size_t flag_words(int ntiles, int B) { return (size_t)ntiles * B; }
inline size_t bins_workspace_bytes(int B, int H, int W, long long total_faces, int elem_size) {
  TileGeom g = tile_geom(H, W);
  return align256((size_t)total_faces * REC_STRIDE * elem_size) + align256((mask_words(g.ntiles, B, total_faces) + flag_words(g.ntiles, B)) * 4);
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

// word offset of (mesh b, tile t) in the mask area
__device__ __forceinline__ size_t mask_base(int ntiles, long long first_b, int b, int t, int stride_b) {
  return (size_t)ntiles * (size_t)(first_b / 32 + b) + (size_t)t * stride_b;
}

// marks every 16x4-pixel sub-tile the pixel range [c_lo,c_hi] x [r_lo,r_hi] touches (one byte per
// (mesh, tile, sub-tile), layout = workgroup item order of the soft-mask kernels; all writers store 1)
__device__ __forceinline__ void mark_sub_tiles(uint8_t* __restrict__ sub_flags, const TileGeom& g, int B, int b, int c_lo,
                                               int c_hi, int r_lo, int r_hi) {
  constexpr int SUBS_X = TILE_W / SUB_W, SUBS = SUBS_X * (TILE_H / SUB_H);
  for (int sy = r_lo / SUB_H; sy <= r_hi / SUB_H; ++sy)
    for (int sx = c_lo / SUB_W; sx <= c_hi / SUB_W; ++sx) {
      const int tile = (sy * SUB_H / TILE_H) * g.tiles_x + (sx * SUB_W / TILE_W);
      const int sub = (sy % (TILE_H / SUB_H)) * SUBS_X + (sx % SUBS_X);
      sub_flags[((size_t)tile * B + b) * SUBS + sub] = 1;
    }
}

// ---- bin kernel -----------------------------------------------------------------------------------
// One thread per face of the (packed) face list.  `first` (B+1, device) gives each mesh's face range;
// first == nullptr means a dense batch: mesh b owns faces [b*F, (b+1)*F).  Copies bbox / vertices / z
// into the record array and sets the face's bit in every tile its bbox can touch.
template <typename T>
__global__ __launch_bounds__(256) void bin_faces_kernel(
    int B, int F_dense, long long total_faces, const int64_t* __restrict__ first,
    const T* __restrict__ bbox, const T* __restrict__ img, const T* __restrict__ z,
    TileGeom g, float multiplier, T* __restrict__ rec, unsigned int* __restrict__ masks,
    unsigned int* __restrict__ tile_flags, uint8_t* __restrict__ sub_flags) {
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  if (f >= total_faces) return;
  int b;
  long long first_b, n_b;
  if (first == nullptr) {
    b = (int)(f / F_dense);
    first_b = (long long)b * F_dense;
    n_b = F_dense;
  } else {
    b = 0;
    while (b + 1 < B && first[b + 1] <= f) ++b;
    first_b = first[b];
    n_b = first[b + 1] - first_b;
    if (f >= first[B]) return;
  }
  const T xmin = bbox[f * 4 + 0], ymin = bbox[f * 4 + 1], xmax = bbox[f * 4 + 2], ymax = bbox[f * 4 + 3];
  T* r = rec + (size_t)f * REC_STRIDE;
  r[0] = xmin;
  r[1] = ymin;
  r[2] = xmax;
  r[3] = ymax;
#pragma unroll
  for (int i = 0; i < 6; ++i) r[4 + i] = img[f * 6 + i];
  if (z != nullptr) {
    r[10] = z[f * 3 + 0];
    r[11] = z[f * 3 + 1];
    r[12] = z[f * 3 + 2];
  }
  // conservative pixel range of the half-open box [xmin,xmax) x [ymin,ymax):
  //   col(x) = (x*W/mult + W - 1)/2 increasing in x, row(y) = (H - 1 - y*H/mult)/2 decreasing in y
  int c_lo = 0, c_hi = g.W - 1, r_lo = 0, r_hi = g.H - 1;
  const double dxmin = (double)xmin, dxmax = (double)xmax, dymin = (double)ymin, dymax = (double)ymax;
  const bool has_nan = !(multiplier > 0.f) || !(dxmin == dxmin) || !(dxmax == dxmax) || !(dymin == dymin) || !(dymax == dymax);
  if (!has_nan) {
    const double sx = (double)g.W / (double)multiplier, sy = (double)g.H / (double)multiplier;
    const double cl = floor((dxmin * sx + g.W - 1) * 0.5) - 1.0, ch = ceil((dxmax * sx + g.W - 1) * 0.5) + 1.0;
    const double rl = floor((g.H - 1 - dymax * sy) * 0.5) - 1.0, rh = ceil((g.H - 1 - dymin * sy) * 0.5) + 1.0;
    if (ch < 0.0 || cl > (double)(g.W - 1) || rh < 0.0 || rl > (double)(g.H - 1)) return;
    c_lo = (int)fmax(cl, 0.0);
    c_hi = (int)fmin(ch, (double)(g.W - 1));
    r_lo = (int)fmax(rl, 0.0);
    r_hi = (int)fmin(rh, (double)(g.H - 1));
  }
  const int tx0 = c_lo / TILE_W, tx1 = c_hi / TILE_W, ty0 = r_lo / TILE_H, ty1 = r_hi / TILE_H;
  const long long j = f - first_b;
  const int stride_b = (int)((n_b + 31) / 32);
  const unsigned int bit = 1u << (unsigned)(j & 31);
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) {
      const int t = ty * g.tiles_x + tx;
      atomicOr(masks + mask_base(g.ntiles, first_b, b, t, stride_b) + (size_t)(j >> 5), bit);
      if (tile_flags[(size_t)b * g.ntiles + t] == 0u) tile_flags[(size_t)b * g.ntiles + t] = 1u;  // benign race: all writers store 1
    }
  if (sub_flags != nullptr) mark_sub_tiles(sub_flags, g, B, b, c_lo, c_hi, r_lo, r_hi);
}

// ---- bin kernel, fused form ------------------------------------------------------------------------------
// Same as bin_faces_kernel for a dense batch (mesh b owns faces [b*F, (b+1)*F)), but starting from the operator's
// RAW inputs, i.e. fusing the torch glue of the reference's Python layer (rasterization.py:292-327, dibr.py:31-39):
//   scaled = face_vertices_image * multiplier          (one rounding, as torch's tensor * scalar)
//   bbox   = [min over the 3 vertices - margin, max + margin]   (margin = boxlen*multiplier, 0 for rasterize)
//   faces with valid[b,f] == 0 are skipped (they are what the reference's packing removes)
// No torch.where (a host sync), no gathers, no packed copies.
// ---- lane = face  ->  lane = pixel -------------------------------------------------------------------------------
// Which of the 64 pixels of a 16x4 sub-tile have their centre inside a face's box?  The pixel grid is regular and
// pixel_x / pixel_y are monotone in col / row, so the answer is (a run of columns) x (a run of rows): the lane that holds
// the face evaluates the reference's reject test (x0 < xmin || x0 >= xmax || y0 < ymin || y0 >= ymax; NaN limits never
// reject) on the 16 column and 4 row coordinates -- the very float expressions every pixel would use -- and forms the
// 64-bit pixel mask as their outer product.  A 64 x 64 bit transpose across the wavefront (6 butterfly stages) then hands
// every lane = pixel the mask of the faces that contain it: ~150 instructions per 64 faces instead of 64 x 10.
template <typename T>
__device__ __forceinline__ unsigned long long sub_tile_pixels_in_box(const Box4<T>& bb, float multiplier, const TileGeom& g,
                                                                     int sub_x, int sub_y) {
  unsigned cols = 0, rows = 0;
#pragma unroll
  for (int c = 0; c < SUB_W; ++c) {
    const T x = pixel_x(multiplier, g.W, sub_x + c);
    cols |= ((x < bb.x0) | (x >= bb.x1)) ? 0u : (1u << c);
  }
#pragma unroll
  for (int r = 0; r < SUB_H; ++r) {
    const T y = pixel_y(multiplier, g.H, sub_y + r);
    rows |= ((y < bb.y0) | (y >= bb.y1)) ? 0u : (1u << r);
  }
  unsigned long long m = 0;
#pragma unroll
  for (int r = 0; r < SUB_H; ++r) m |= ((rows >> r) & 1u) ? ((unsigned long long)cols << (SUB_W * r)) : 0ull;
  return m;
}
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

// sets, for every tile some lane's box touches, the bits of the touching lanes' faces in that tile's mask.
// Consecutive lanes hold consecutive faces of (usually) one mesh, so the 32 faces of a mask word live in one wavefront:
// a per-lane atomicOr would send up to 32 same-address atomics to L2 for every word.  Here the wavefront walks the tiles
// of the union of its boxes, takes ONE ballot per tile and issues at most three atomicOr per (tile, mesh) -- the words
// its 64 faces straddle.  A wavefront whose union is large (a huge face, or an incoherent face order) keeps the per-lane path.
__device__ __forceinline__ void bin_emit_wave(bool active, int b, long long j, int tx0, int tx1, int ty0, int ty1, int F,
                                              const TileGeom& g, unsigned int* __restrict__ masks,
                                              unsigned int* __restrict__ tile_flags) {
  const int lane = threadIdx.x & 63;
  const int stride_b = (F + 31) / 32;
  unsigned long long remaining = __ballot(active);
  while (remaining != 0ull) {
    const int leader = __ffsll((long long)remaining) - 1;
    const int bL = __builtin_amdgcn_readlane(b, leader);
    const bool mine = active && b == bL;
    remaining &= ~__ballot(mine);
    // union of the group's tile rectangles
    int ux0 = mine ? tx0 : 0x7fffffff, uy0 = mine ? ty0 : 0x7fffffff, ux1 = mine ? tx1 : -1, uy1 = mine ? ty1 : -1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      ux0 = min(ux0, __shfl_xor(ux0, d, 64));
      uy0 = min(uy0, __shfl_xor(uy0, d, 64));
      ux1 = max(ux1, __shfl_xor(ux1, d, 64));
      uy1 = max(uy1, __shfl_xor(uy1, d, 64));
    }
    const long long first_b = (long long)bL * F;
    if ((ux1 - ux0 + 1) * (uy1 - uy0 + 1) > 36) {  // not a compact patch: per-lane atomics
      if (mine) {
        const unsigned int bit = 1u << (unsigned)(j & 31);
        for (int ty = ty0; ty <= ty1; ++ty)
          for (int tx = tx0; tx <= tx1; ++tx) {
            const int t = ty * g.tiles_x + tx;
            atomicOr(masks + mask_base(g.ntiles, first_b, bL, t, stride_b) + (size_t)(j >> 5), bit);
            if (tile_flags[(size_t)bL * g.ntiles + t] == 0u) tile_flags[(size_t)bL * g.ntiles + t] = 1u;
          }
      }
      continue;
    }
    // lanes of the group hold consecutive faces: lane l has face j0 + l, j0 = (face of lane 0, possibly virtual)
    const long long j0 = __shfl(j, leader, 64) - leader;
    const int s = (int)(j0 & 31);                // lane l's bit = (s + l) & 31, word = (j0 >> 5) + ((s + l) >> 5)
    for (int ty = uy0; ty <= uy1; ++ty)
      for (int tx = ux0; tx <= ux1; ++tx) {
        const unsigned long long bal = __ballot(mine && tx >= tx0 && tx <= tx1 && ty >= ty0 && ty <= ty1);
        if (bal == 0ull) continue;
        const int t = ty * g.tiles_x + tx;
        if (lane < 3) {
          // word k (k = 0,1,2) collects lanes [32k - s, 32k - s + 32) at bit (lane - (32k - s))
          const int lo = 32 * lane - s;
          const unsigned long long part = lo >= 0 ? (lo < 64 ? bal >> lo : 0ull) : bal << (-lo);
          const unsigned int bits = (unsigned int)(part & 0xffffffffull);
          if (bits != 0u)
            atomicOr(masks + mask_base(g.ntiles, first_b, bL, t, stride_b) + (size_t)((j0 >> 5) + lane), bits);
        }
        if (lane == 0) tile_flags[(size_t)bL * g.ntiles + t] = 1u;
      }
  }
}

// element strides of the optional per-face inputs of the fused front doors (dense: z_face 3, z_vertex 1, front_stride 1)
struct FaceLayout {
  long long z_face, z_vertex, front_stride;
};

template <typename T>
struct alignas(16) Rec4 {
  T a, b, c, d;
};

template <typename T>
__global__ __launch_bounds__(256) void bin_faces_raw_kernel(
    int B, int F, const T* __restrict__ img, const T* __restrict__ z, FaceLayout lay, const uint8_t* __restrict__ valid,
    const T* __restrict__ front, T mult, T margin, TileGeom g, float multiplier, T* __restrict__ rec,
    unsigned int* __restrict__ masks, unsigned int* __restrict__ tile_flags, uint8_t* __restrict__ sub_flags) {
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  bool active = f < (long long)B * F;
  if (active && valid != nullptr && valid[f] == 0) active = false;
  // `front`: a per-face scalar (the z of the face normal) read in place; the face is kept when it is >= 0, which is the
  // mask `face_normals_z >= 0` of the reference's dibr_rasterization (dibr.py:188) without the compare kernel
  if (active && front != nullptr && !(front[f * lay.front_stride] >= (T)0)) active = false;
  int b = 0, tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  long long j = 0;
  if (active) {
    b = (int)(f / F);
    j = f - (long long)b * F;
    T v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = img[f * 6 + i] * mult;
    T xmin = fmin(fmin(v[0], v[2]), v[4]), xmax = fmax(fmax(v[0], v[2]), v[4]);
    T ymin = fmin(fmin(v[1], v[3]), v[5]), ymax = fmax(fmax(v[1], v[3]), v[5]);
    if (margin != (T)0) {
      xmin = xmin - margin;
      ymin = ymin - margin;
      xmax = xmax + margin;
      ymax = ymax + margin;
    }
    // the 16-scalar record as four 16-byte stores: box | a.xy b.xy | c.xy z.ab | z.c pad
    Rec4<T>* r = reinterpret_cast<Rec4<T>*>(rec + (size_t)f * REC_STRIDE);
    T z0 = 0, z1 = 0, z2 = 0;
    if (z != nullptr) {  // read in place: z may be the [..., 2] view of the (B, F, 3, 3) camera-space vertices
      z0 = z[f * lay.z_face + 0 * lay.z_vertex];
      z1 = z[f * lay.z_face + 1 * lay.z_vertex];
      z2 = z[f * lay.z_face + 2 * lay.z_vertex];
    }
    r[0] = Rec4<T>{xmin, ymin, xmax, ymax};
    r[1] = Rec4<T>{v[0], v[1], v[2], v[3]};
    r[2] = Rec4<T>{v[4], v[5], z0, z1};
    r[3] = Rec4<T>{z2, 0, 0, 0};
    int c_lo = 0, c_hi = g.W - 1, r_lo = 0, r_hi = g.H - 1;
    const double dxmin = (double)xmin, dxmax = (double)xmax, dymin = (double)ymin, dymax = (double)ymax;
    const bool has_nan = !(multiplier > 0.f) || !(dxmin == dxmin) || !(dxmax == dxmax) || !(dymin == dymin) || !(dymax == dymax);
    if (!has_nan) {
      const double sx = (double)g.W / (double)multiplier, sy = (double)g.H / (double)multiplier;
      const double cl = floor((dxmin * sx + g.W - 1) * 0.5) - 1.0, ch = ceil((dxmax * sx + g.W - 1) * 0.5) + 1.0;
      const double rl = floor((g.H - 1 - dymax * sy) * 0.5) - 1.0, rh = ceil((g.H - 1 - dymin * sy) * 0.5) + 1.0;
      if (ch < 0.0 || cl > (double)(g.W - 1) || rh < 0.0 || rl > (double)(g.H - 1)) {
        active = false;  // off screen (the record is still written: nothing reads it)
      } else {
        c_lo = (int)fmax(cl, 0.0);
        c_hi = (int)fmin(ch, (double)(g.W - 1));
        r_lo = (int)fmax(rl, 0.0);
        r_hi = (int)fmin(rh, (double)(g.H - 1));
      }
    }
    if (active) {
      tx0 = c_lo / TILE_W;
      tx1 = c_hi / TILE_W;
      ty0 = r_lo / TILE_H;
      ty1 = r_hi / TILE_H;
      if (sub_flags != nullptr) mark_sub_tiles(sub_flags, g, B, b, c_lo, c_hi, r_lo, r_hi);
    }
  }
  bin_emit_wave(active, b, j, tx0, tx1, ty0, ty1, F, g, masks, tile_flags);
}

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

// ---- block-wide exclusive scan over 1024 threads (16 wavefronts) -------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
// returns the exclusive prefix of v over the block; *total = block sum.  `scratch` holds 17 ints.
__device__ __forceinline__ int block_exclusive_scan(int v, int* scratch, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inc = wave_inclusive_scan(v);
  __syncthreads();  // scratch may still be read by the previous round
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
#pragma unroll
    for (int w = 0; w < TILE_THREADS / 64; ++w) {
      const int s = scratch[w];
      scratch[w] = run;
      run += s;
    }
    scratch[TILE_THREADS / 64] = run;
  }
  __syncthreads();
  *total = scratch[TILE_THREADS / 64];
  return scratch[wave] + inc - v;
}

}  // namespace kamd
