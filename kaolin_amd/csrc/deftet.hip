// DefTet sparse renderer: multi-hit rasterization of FREE pixel coordinates.
//
// Replaces kaolin/csrc/render/mesh/deftet_cuda.cu (forward :31-186, backward :240-404) and the torch chain that
// follows the forward operator in kaolin/render/mesh/deftet.py:297-311 (argsort by depth, 3 gathers, pad, stack, sum).
//
// The reference visits every face for every pixel (P x F box tests, a warp per pixel).  Here the work is the number of
// (pixel, face) pairs that are CLOSE, whichever of P and F is large:
//   1. pixels are counting-sorted by the Z-order number of their cell in a G x G grid over their own extent, G chosen
//      so that a cell holds ~8 pixels (16 <= G <= 256; one global atomic per pixel, little contention by construction);
//      the sorted copy (x, y, range) makes a cell's pixels contiguous;
//   2. thread = face walks the pixels of the few cells its box overlaps, runs the reference's arithmetic on the ones
//      inside the box and appends every hit to that pixel's row through a per-pixel counter (any order);
//      a face whose cells hold more than 256 pixels is handed to a wavefront (or, above 1024 cells, a workgroup) of its
//      own, 4 lanes per cell;
//   3. rows are put in mesh order afterwards (they hold a handful of entries).  A pixel that collected MORE than knum
//      hits needs "the first knum in mesh order": those (rare) pixels are redone the reference's way, one wavefront per
//      pixel streaming the faces in order with ballot-prefix appends.
// Arithmetic follows the reference expression by expression (-ffp-contract=off): integer outputs are bit-exact vs
// oracle/deftet_oracle.inc, floats equal; results do not depend on the order in which hits arrive.
#include "common.h"
#include "profile.h"
#include "tile_bins.h"
#include "../../include/kaolin_amd.h"

namespace {
using kamd::Box4;

constexpr int DT_THREADS = 256;
constexpr int DT_NB = 64;              // extent partials per batch item
constexpr int DT_EXT_THREADS = 1024;
constexpr int DT_MAX_GSHIFT = 8;       // at most 256 x 256 cells
constexpr int DT_SMALL_CELLS = 16;     // thread = face handles a box over at most this many cells ...
constexpr int DT_SMALL_CAND = 256;     // ... holding at most this many pixels
constexpr int DT_WAVE_CELLS = 1024;    // up to here a wavefront per face, above a workgroup per face

// grid resolution: G = 2^gshift cells per axis so that a cell holds about 8 pixels (16 <= G <= 256)
__host__ __device__ inline int dt_gshift(int P) {
  int g = 4;
  while (g < DT_MAX_GSHIFT && ((size_t)1 << (2 * g)) * 8 < (size_t)P) ++g;
  return g;
}

// workspace, 4-byte words (B batch items, P pixels, F faces, NC = G*G cells), ST = sizeof(T) / 4:
//   [zeroed every call]  counters (4) | cell starts B*(NC+4) | per-pixel hit counters B*P (padded to 4)
//   [fully written]      extent partials B*NB*4 | cell cursors B*NC | order B*P | wave-face list B*F | workgroup-face
//                        list B*F | overflow list B*P | sorted pixels B*P*4*ST
struct DtWs {
  int* counters;   // [0] = faces for the wave kernel, [1] = faces for the workgroup kernel, [2] = overflowing pixels
  int* start;      // + b * (NC + 4)
  int* cnt;        // + b * P
  float* part;     // + b * NB * 4
  int* cursor;     // + b * NC
  int* order;      // + b * P
  int* wlist;      // (b * F + f) entries
  int* glist;      // (b * F + f) entries
  int* over;       // (b * P + p) entries
  void* spix;      // 4 T per sorted pixel: x, y, min depth, max depth
  size_t zero_words, total_words;
  int gshift, nc;
};
__host__ __device__ inline size_t dt_pad4(size_t n) { return (n + 3) & ~(size_t)3; }
__host__ __device__ inline DtWs dt_ws(void* base, int B, int F, int P, int st) {
  DtWs r;
  r.gshift = dt_gshift(P);
  r.nc = 1 << (2 * r.gshift);
  int* w = (int*)base;
  r.counters = w;
  w += 4;
  r.start = w;
  w += (size_t)B * (r.nc + 4);
  r.cnt = w;
  w += dt_pad4((size_t)B * P);
  r.zero_words = (size_t)(w - (int*)base);
  r.part = (float*)w;
  w += (size_t)B * DT_NB * 4;
  r.cursor = w;
  w += (size_t)B * r.nc;
  r.order = w;
  w += dt_pad4((size_t)B * P);
  r.wlist = w;
  w += dt_pad4((size_t)B * F);
  r.glist = w;
  w += dt_pad4((size_t)B * F);
  r.over = w;
  w += dt_pad4((size_t)B * P);
  r.spix = (void*)w;
  w += (size_t)B * P * 4 * st;
  r.total_words = (size_t)(w - (int*)base);
  return r;
}

// ---- 1. extent of the finite pixel coordinates (partials), as floats rounded OUTWARDS ---------------------------
template <typename T>
__device__ __forceinline__ float dt_round_down(T v) {
  float f = (float)v;
  if ((T)f > v) f = nextafterf(f, -INFINITY);
  return f;
}
template <typename T>
__device__ __forceinline__ float dt_round_up(T v) {
  float f = (float)v;
  if ((T)f < v) f = nextafterf(f, INFINITY);
  return f;
}
template <typename T>
__global__ __launch_bounds__(DT_EXT_THREADS) void dt_extent_kernel(int P, const T* __restrict__ pix, float* __restrict__ part) {
  __shared__ float s[4][DT_EXT_THREADS];
  const int b = blockIdx.y;
  const T* X = pix + (size_t)b * P * 2;
  float lo0 = INFINITY, lo1 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY;
  for (int i = blockIdx.x * DT_EXT_THREADS + threadIdx.x; i < P; i += gridDim.x * DT_EXT_THREADS) {
    const T x = X[(size_t)i * 2], y = X[(size_t)i * 2 + 1];
    if (isfinite(x) && isfinite(y)) {  // a pixel with a non-finite coordinate is inside no face (see dt_face_kernel)
      lo0 = fminf(lo0, dt_round_down<T>(x));
      hi0 = fmaxf(hi0, dt_round_up<T>(x));
      lo1 = fminf(lo1, dt_round_down<T>(y));
      hi1 = fmaxf(hi1, dt_round_up<T>(y));
    }
  }
  s[0][threadIdx.x] = lo0;
  s[1][threadIdx.x] = lo1;
  s[2][threadIdx.x] = hi0;
  s[3][threadIdx.x] = hi1;
  __syncthreads();
  for (int d = DT_EXT_THREADS / 2; d >= 1; d >>= 1) {
    if (threadIdx.x < d) {
      s[0][threadIdx.x] = fminf(s[0][threadIdx.x], s[0][threadIdx.x + d]);
      s[1][threadIdx.x] = fminf(s[1][threadIdx.x], s[1][threadIdx.x + d]);
      s[2][threadIdx.x] = fmaxf(s[2][threadIdx.x], s[2][threadIdx.x + d]);
      s[3][threadIdx.x] = fmaxf(s[3][threadIdx.x], s[3][threadIdx.x + d]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) part[((size_t)b * DT_NB + blockIdx.x) * 4 + threadIdx.x] = s[threadIdx.x][0];
}

struct DtGrid {
  float lo[2], hi[2], inv[2];
  int G;
};
// every consumer re-reduces the 64 partials of its batch item (1 KB from L2); first wave, broadcast through LDS
__device__ __forceinline__ void dt_grid_setup(const float* __restrict__ part, int gshift, DtGrid* g) {
  if (threadIdx.x < 64) {
    float lo0 = part[threadIdx.x * 4], lo1 = part[threadIdx.x * 4 + 1];
    float hi0 = part[threadIdx.x * 4 + 2], hi1 = part[threadIdx.x * 4 + 3];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lo0 = fminf(lo0, __shfl_xor(lo0, d, 64));
      lo1 = fminf(lo1, __shfl_xor(lo1, d, 64));
      hi0 = fmaxf(hi0, __shfl_xor(hi0, d, 64));
      hi1 = fmaxf(hi1, __shfl_xor(hi1, d, 64));
    }
    if (threadIdx.x == 0) {
      const float lo[2] = {lo0, lo1}, hi[2] = {hi0, hi1};
      g->G = 1 << gshift;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float size = (hi[a] - lo[a]) / (float)(1 << gshift);
        if (!(size > 0.f) || !isfinite(size)) size = 1.f;  // degenerate extent: one slab holds everything
        g->lo[a] = lo[a];  // +inf / -inf when no pixel is finite: every face is then rejected
        g->hi[a] = hi[a];
        g->inv[a] = 1.f / size;
      }
    }
  }
  __syncthreads();
}
// monotone non-decreasing in v (float subtract, multiply, truncate, clamp): a <= b  =>  cell(a) <= cell(b)
__device__ __forceinline__ int dt_axis_cell(float v, float lo, float inv, int G) {
  const float t = (v - lo) * inv;
  return (t >= 0.f) ? (t < (float)G ? (int)t : G - 1) : 0;  // NaN -> 0, +-inf clamped
}
__device__ __forceinline__ unsigned dt_spread8(unsigned v) {  // abcdefgh -> 0a0b0c0d0e0f0g0h
  v = (v | (v << 4)) & 0x0F0Fu;
  v = (v | (v << 2)) & 0x3333u;
  v = (v | (v << 1)) & 0x5555u;
  return v;
}
// Z-order cell number: neighbouring cells are neighbours in memory, which keeps a face's pixel reads in few lines
__device__ __forceinline__ int dt_code(int cx, int cy) { return (int)(dt_spread8((unsigned)cx) | (dt_spread8((unsigned)cy) << 1)); }
template <typename T>
__device__ __forceinline__ int dt_cell(const DtGrid& g, const T* __restrict__ X, int i) {
  return dt_code(dt_axis_cell((float)X[(size_t)i * 2], g.lo[0], g.inv[0], g.G),
                 dt_axis_cell((float)X[(size_t)i * 2 + 1], g.lo[1], g.inv[1], g.G));
}

// ---- 2. cell histogram.  A cell holds ~8 pixels by construction, so plain global atomics see little contention --------
template <typename T>
__global__ __launch_bounds__(DT_THREADS) void dt_count_kernel(int P, const T* __restrict__ pix, DtWs w) {
  __shared__ DtGrid s_g;
  const int b = blockIdx.y;
  dt_grid_setup(w.part + (size_t)b * DT_NB * 4, w.gshift, &s_g);
  const int i = blockIdx.x * DT_THREADS + threadIdx.x;
  if (i < P) atomicAdd(&w.start[(size_t)b * (w.nc + 4) + dt_cell<T>(s_g, pix + (size_t)b * P * 2, i)], 1);
}

// ---- 3. exclusive scan of the NC counts in two launches: sums of 1024-entry blocks, then every block adds the sums
//         before it and scans itself (entry NC receives the total); cursors start at the cell starts -----------------
__device__ __forceinline__ int dt_block_inclusive(int v, int* s_wave) {
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
  for (int k = 0; k < wave; ++k) woff += s_wave[k];
  return woff + inc;
}
__global__ __launch_bounds__(1024) void dt_scan_sums_kernel(DtWs w, int* __restrict__ sums) {
  __shared__ int s_wave[16];
  const int* start = w.start + (size_t)blockIdx.y * (w.nc + 4);
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const int tot = dt_block_inclusive(i < w.nc ? start[i] : 0, s_wave);
  if (threadIdx.x == 1023) sums[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void dt_scan_apply_kernel(DtWs w, const int* __restrict__ sums) {
  __shared__ int s_wave[16];
  __shared__ int s_off;
  int* start = w.start + (size_t)blockIdx.y * (w.nc + 4);
  int* cursor = w.cursor + (size_t)blockIdx.y * w.nc;
  const int* my = sums + blockIdx.y * gridDim.x;
  int part = 0;
  for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) part += my[k];
  const int before = dt_block_inclusive(part, s_wave);
  if (threadIdx.x == 1023) s_off = before;
  __syncthreads();
  const int off = s_off;
  __syncthreads();
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const int v = i < w.nc ? start[i] : 0;
  const int inc = dt_block_inclusive(v, s_wave);
  if (i < w.nc) {
    start[i] = off + inc - v;
    cursor[i] = off + inc - v;
  }
  if (i == w.nc - 1) start[w.nc] = off + inc;
}

// ---- 4. scatter into cell order, with a contiguous copy of (x, y, min depth, max depth) -----------------------------
template <typename T>
__global__ __launch_bounds__(DT_THREADS) void dt_scatter_kernel(int P, const T* __restrict__ pix,
                                                                const T* __restrict__ range, DtWs w) {
  __shared__ DtGrid s_g;
  const int b = blockIdx.y;
  dt_grid_setup(w.part + (size_t)b * DT_NB * 4, w.gshift, &s_g);
  const int i = blockIdx.x * DT_THREADS + threadIdx.x;
  if (i >= P) return;
  const T* X = pix + (size_t)b * P * 2;
  const int pos = atomicAdd(&w.cursor[(size_t)b * w.nc + dt_cell<T>(s_g, X, i)], 1);
  w.order[(size_t)b * P + pos] = i;
  Box4<T> v;
  v.x0 = X[(size_t)i * 2];
  v.y0 = X[(size_t)i * 2 + 1];
  v.x1 = range[((size_t)b * P + i) * 2];
  v.y1 = range[((size_t)b * P + i) * 2 + 1];
  reinterpret_cast<Box4<T>*>(w.spix)[(size_t)b * P + pos] = v;
}

// ---- outputs of the reference wrapper start as (-1, -inf, 0, 0) (deftet.cpp:90-96) ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dt_fill_kernel(size_t n, int64_t* __restrict__ face_idx, T* __restrict__ depth,
                                                       T* __restrict__ w0, T* __restrict__ w1) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    face_idx[i] = -1;
    depth[i] = (T)(-INFINITY);
    w0[i] = 0;
    w1[i] = 0;
  }
}

// ---- 5. thread = face -----------------------------------------------------------------------------------------------
template <typename T>
struct DtFace {
  T ax, ay, bx, by, cx, cy, az, bz, cz;
};
template <typename T>
__device__ __forceinline__ DtFace<T> dt_load_face(const T* __restrict__ fimg, const T* __restrict__ fz, size_t g) {
  DtFace<T> fc;
  fc.ax = fimg[g * 6 + 0];
  fc.ay = fimg[g * 6 + 1];
  fc.bx = fimg[g * 6 + 2];
  fc.by = fimg[g * 6 + 3];
  fc.cx = fimg[g * 6 + 4];
  fc.cy = fimg[g * 6 + 5];
  fc.az = fz[g * 3 + 0];
  fc.bz = fz[g * 3 + 1];
  fc.cz = fz[g * 3 + 2];
  return fc;
}
// deftet_cuda.cu:117-152 for one (pixel, face): box [min,max), normalised edge functions >= 0, depth in [min,max)
template <typename T>
__device__ __forceinline__ bool dt_hit(const Box4<T>& bb, const DtFace<T>& fc, T x0, T y0, T dmin, T dmax, T eps_f, T* w0o,
                                       T* w1o, T* dout) {
  if (!((x0 >= bb.x0) & (x0 < bb.x1) & (y0 >= bb.y0) & (y0 < bb.y1))) return false;
  const T aex = fc.ax - x0, aey = fc.ay - y0, bex = fc.bx - x0, bey = fc.by - y0, cex = fc.cx - x0, cey = fc.cy - y0;
  const T u0 = bex * cey - bey * cex;
  const T u1 = cex * aey - cey * aex;
  const T u2 = aex * bey - aey * bex;
  const T norm = u0 + u1 + u2;
  const T norm_eps = (T)copysignf((float)eps_f, (float)norm);  // both arguments as FLOAT (:135-136)
  const T w0 = u0 / (norm + norm_eps), w1 = u1 / (norm + norm_eps), w2 = u2 / (norm + norm_eps);
  if (!(w0 >= 0. && w1 >= 0. && w2 >= 0.)) return false;
  const T d = w0 * fc.az + w1 * fc.bz + w2 * fc.cz;
  if (!(d < dmax && d >= dmin)) return false;
  *w0o = w0;
  *w1o = w1;
  *dout = d;
  return true;
}
template <typename T>
struct DtOut {
  int64_t* face_idx;
  T *depth, *w0, *w1;
};
// appends to the pixel's row in arrival order; the counter keeps counting past K so that overflow is detectable
template <typename T>
__device__ __forceinline__ void dt_append(const DtOut<T>& o, int* __restrict__ cnt, size_t bp, int K, int f, T d, T w0, T w1) {
  const int slot = atomicAdd(&cnt[bp], 1);
  if (slot < K) {
    const size_t i = bp * K + slot;
    o.face_idx[i] = f;
    o.depth[i] = d;
    o.w0[i] = w0;
    o.w1[i] = w1;
  }
}
// the pixels of one cell against one face
template <typename T>
__device__ __forceinline__ void dt_face_cell(const Box4<T>& bb, const DtFace<T>& fc, int f, int b, int P, int K, T eps_f,
                                             const Box4<T>* __restrict__ spix, const int* __restrict__ order,
                                             int i0, int i1, int step, int first, const DtOut<T>& o, int* __restrict__ cnt) {
  for (int i = i0 + first; i < i1; i += step) {
    const Box4<T> px = spix[i];  // x, y, min depth, max depth
    T w0, w1, d;
    if (dt_hit<T>(bb, fc, px.x0, px.y0, px.x1, px.y1, eps_f, &w0, &w1, &d))
      dt_append<T>(o, cnt, (size_t)b * P + order[i], K, f, d, w0, w1);
  }
}

struct DtCells {
  int cx0, cx1, cy0, cy1;
};
template <typename T>
__device__ __forceinline__ DtCells dt_cells_of(const DtGrid& g, const Box4<T>& bb) {
  // xmin <= x < xmax and the cell mapping is monotone: the pixel's cell lies in [cell(xmin), cell(xmax)]
  DtCells c;
  c.cx0 = dt_axis_cell((float)bb.x0, g.lo[0], g.inv[0], g.G);
  c.cx1 = dt_axis_cell((float)bb.x1, g.lo[0], g.inv[0], g.G);
  c.cy0 = dt_axis_cell((float)bb.y0, g.lo[1], g.inv[1], g.G);
  c.cy1 = dt_axis_cell((float)bb.y1, g.lo[1], g.inv[1], g.G);
  return c;
}

template <typename T>
__global__ __launch_bounds__(DT_THREADS) void dt_face_kernel(int F, int P, int K, const T* __restrict__ fz,
                                                             const T* __restrict__ fimg, const T* __restrict__ fbb,
                                                             float eps, DtWs w, DtOut<T> o) {
  __shared__ DtGrid s_g;
  const int b = blockIdx.y;
  dt_grid_setup(w.part + (size_t)b * DT_NB * 4, w.gshift, &s_g);
  const int f = blockIdx.x * DT_THREADS + threadIdx.x;
  if (f >= F) return;
  const size_t g = (size_t)b * F + f;
  const Box4<T> bb = reinterpret_cast<const Box4<T>*>(fbb)[g];
  // an empty box, a NaN limit, or a box that misses the extent of the finite pixels contains no pixel.
  // (a pixel with a non-finite coordinate is in no face either: inside [min,max) it needs a vertex at -inf on that axis,
  //  which turns the edge functions and all three weights into NaN)
  if (!((bb.x0 < bb.x1) & (bb.y0 < bb.y1))) return;
  if ((bb.x1 <= (T)s_g.lo[0]) | (bb.x0 > (T)s_g.hi[0]) | (bb.y1 <= (T)s_g.lo[1]) | (bb.y0 > (T)s_g.hi[1])) return;
  const DtCells c = dt_cells_of<T>(s_g, bb);
  const int* start = w.start + (size_t)b * (w.nc + 4);
  const int ncell = (c.cx1 - c.cx0 + 1) * (c.cy1 - c.cy0 + 1);
  int cand = DT_SMALL_CAND + 1;
  if (ncell <= DT_SMALL_CELLS) {
    cand = 0;
    for (int cy = c.cy0; cy <= c.cy1; ++cy)
      for (int cx = c.cx0; cx <= c.cx1; ++cx) {
        const int z = dt_code(cx, cy);
        cand += start[z + 1] - start[z];
      }
  }
  if (cand == 0) return;
  if (cand > DT_SMALL_CAND) {  // hand over: a wavefront per face, or a workgroup when the box spans very many cells
    if (ncell <= DT_WAVE_CELLS)
      w.wlist[atomicAdd(&w.counters[0], 1)] = (int)g;
    else
      w.glist[atomicAdd(&w.counters[1], 1)] = (int)g;
    return;
  }
  const DtFace<T> fc = dt_load_face<T>(fimg, fz, g);
  const T eps_f = (T)(float)(double)eps;
  const Box4<T>* spix = reinterpret_cast<const Box4<T>*>(w.spix) + (size_t)b * P;
  const int* order = w.order + (size_t)b * P;
  for (int cy = c.cy0; cy <= c.cy1; ++cy)
    for (int cx = c.cx0; cx <= c.cx1; ++cx) {
      const int z = dt_code(cx, cy);
      dt_face_cell<T>(bb, fc, f, b, P, K, eps_f, spix, order, start[z], start[z + 1], 1, 0, o, w.cnt);
    }
}

// a group of GROUP threads (a wavefront, or the whole workgroup) per handed-over face: 4 lanes per cell
template <typename T, int GROUP>
__global__ __launch_bounds__(DT_THREADS) void dt_big_face_kernel(int F, int P, int K, const T* __restrict__ fz,
                                                                 const T* __restrict__ fimg, const T* __restrict__ fbb,
                                                                 float eps, DtWs w, DtOut<T> o) {
  const int n = GROUP == 64 ? w.counters[0] : w.counters[1];
  const int* list = GROUP == 64 ? w.wlist : w.glist;
  const T eps_f = (T)(float)(double)eps;
  const int groups = DT_THREADS / GROUP, gid = threadIdx.x / GROUP, t = threadIdx.x % GROUP;
  for (int q = blockIdx.x * groups + gid; q < n; q += gridDim.x * groups) {
    const size_t g = (size_t)list[q];
    const int b = (int)(g / (size_t)F), f = (int)(g - (size_t)b * F);
    // grid geometry of batch item b, re-reduced by every group (64 partials)
    DtGrid gr;
    {
      const float* part = w.part + (size_t)b * DT_NB * 4;
      const int l = threadIdx.x & 63;
      float lo0 = part[l * 4], lo1 = part[l * 4 + 1], hi0 = part[l * 4 + 2], hi1 = part[l * 4 + 3];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        lo0 = fminf(lo0, __shfl_xor(lo0, d, 64));
        lo1 = fminf(lo1, __shfl_xor(lo1, d, 64));
        hi0 = fmaxf(hi0, __shfl_xor(hi0, d, 64));
        hi1 = fmaxf(hi1, __shfl_xor(hi1, d, 64));
      }
      const float lo[2] = {lo0, lo1}, hi[2] = {hi0, hi1};
      gr.G = 1 << w.gshift;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float size = (hi[a] - lo[a]) / (float)gr.G;
        if (!(size > 0.f) || !isfinite(size)) size = 1.f;
        gr.lo[a] = lo[a];
        gr.hi[a] = hi[a];
        gr.inv[a] = 1.f / size;
      }
    }
    const Box4<T> bb = reinterpret_cast<const Box4<T>*>(fbb)[g];
    const DtFace<T> fc = dt_load_face<T>(fimg, fz, g);
    const DtCells c = dt_cells_of<T>(gr, bb);
    const int nx = c.cx1 - c.cx0 + 1, ncell = nx * (c.cy1 - c.cy0 + 1);
    const int* start = w.start + (size_t)b * (w.nc + 4);
    const Box4<T>* spix = reinterpret_cast<const Box4<T>*>(w.spix) + (size_t)b * P;
    const int* order = w.order + (size_t)b * P;
    for (int ci = t >> 2; ci < ncell; ci += GROUP / 4) {
      const int z = dt_code(c.cx0 + ci % nx, c.cy0 + ci / nx);
      dt_face_cell<T>(bb, fc, f, b, P, K, eps_f, spix, order, start[z], start[z + 1], 4, t & 3, o, w.cnt);
    }
  }
}

// ---- 6. rows in mesh order; overflowing pixels redone exactly ----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dt_order_rows_kernel(size_t BP, int K, DtOut<T> o, DtWs w, int* __restrict__ hit_count,
                                                            int sort_rows) {
  const size_t bp = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (bp >= BP) return;
  const int n = w.cnt[bp];
  if (n > K) {
    w.over[atomicAdd(&w.counters[2], 1)] = (int)bp;  // B * P < 2^31 (checked on the host)
    return;                                          // dt_overflow_kernel writes the row and its count
  }
  if (hit_count) hit_count[bp] = n;
  if (!sort_rows) return;
  const size_t row = bp * K;
  for (int i = 1; i < n; ++i) {  // insertion sort by face number: rows hold a handful of entries
    const int64_t f = o.face_idx[row + i];
    const T d = o.depth[row + i], a = o.w0[row + i], c = o.w1[row + i];
    int j = i - 1;
    while (j >= 0 && o.face_idx[row + j] > f) {
      o.face_idx[row + j + 1] = o.face_idx[row + j];
      o.depth[row + j + 1] = o.depth[row + j];
      o.w0[row + j + 1] = o.w0[row + j];
      o.w1[row + j + 1] = o.w1[row + j];
      --j;
    }
    o.face_idx[row + j + 1] = f;
    o.depth[row + j + 1] = d;
    o.w0[row + j + 1] = a;
    o.w1[row + j + 1] = c;
  }
}

// one wavefront per overflowing pixel: faces streamed in mesh order, 64 at a time, hits appended by ballot prefix
template <typename T>
__global__ __launch_bounds__(DT_THREADS) void dt_overflow_kernel(int F, int P, int K, const T* __restrict__ fz,
                                                                 const T* __restrict__ fimg, const T* __restrict__ fbb,
                                                                 const T* __restrict__ pix, const T* __restrict__ range,
                                                                 float eps, DtWs w, DtOut<T> o, int* __restrict__ hit_count) {
  const int nover = w.counters[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T eps_f = (T)(float)(double)eps;
  for (int q = blockIdx.x * (DT_THREADS / 64) + wave; q < nover; q += gridDim.x * (DT_THREADS / 64)) {
    const size_t bp = (size_t)w.over[q];
    const int b = (int)(bp / (size_t)P);
    const T x0 = pix[bp * 2], y0 = pix[bp * 2 + 1], dmin = range[bp * 2], dmax = range[bp * 2 + 1];
    const size_t row = bp * K;
    int n = 0;
    for (int base = 0; base < F && n < K; base += 64) {
      const int f = base + lane;
      bool hit = false;
      T w0 = 0, w1 = 0, d = 0;
      if (f < F) {
        const size_t g = (size_t)b * F + f;
        const Box4<T> bb = reinterpret_cast<const Box4<T>*>(fbb)[g];
        if ((x0 >= bb.x0) & (x0 < bb.x1) & (y0 >= bb.y0) & (y0 < bb.y1))
          hit = dt_hit<T>(bb, dt_load_face<T>(fimg, fz, g), x0, y0, dmin, dmax, eps_f, &w0, &w1, &d);
      }
      const unsigned long long m = __ballot(hit);
      const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && pos < K) {
        o.face_idx[row + pos] = f;
        o.depth[row + pos] = d;
        o.w0[row + pos] = w0;
        o.w1[row + pos] = w1;
      }
      n += __popcll(m);
    }
    if (lane == 0 && hit_count) hit_count[bp] = K;
  }
}

// ---- 7. depth sort + weights + interpolation (deftet.py:297-311) ------------------------------------------------
// rank of entry k among the pixel's n hits by (depth descending, then face number) = its output slot; slots >= n keep
// the defaults.  Rows arrive in any order; equal depths are ordered by face number = mesh order
// (torch.argsort in the reference leaves that case open).
// the sorted result is mostly defaults (knum slots, a handful of hits): they go out as flat 16-byte stores ...
inline int dt_fill_bytes(void* ptr, size_t bytes, unsigned char byte, hipStream_t st) {  // byte pattern, any size
  return kamd_fill_async(ptr, bytes, byte, st);
}
// ... and the hits are then ranked and written over them: KH lanes per pixel, lane k takes entries k, k + KH, ...
template <typename T, int KH>
__global__ __launch_bounds__(256) void dt_sort_interp_kernel(int B, int F, int P, int K, int D,
                                                             const int64_t* __restrict__ face_idx,
                                                             const T* __restrict__ depth, const T* __restrict__ w0a,
                                                             const T* __restrict__ w1a, const int* __restrict__ hit_count,
                                                             const T* __restrict__ feat, int64_t* __restrict__ out_idx,
                                                             T* __restrict__ out_w, T* __restrict__ out_feat) {
  // KH == 0: small results, one pass -- a thread per (pixel, slot) that also writes the defaults of the unused slots
  const int kh = KH > 0 ? KH : K;
  const size_t total = (size_t)B * P * kh;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t bp = i / kh;
    const int n = min(hit_count[bp], K);
    const size_t row = bp * K;
    if (KH == 0 && (int)(i - bp * kh) >= n) {
      out_idx[i] = -1;
      out_w[i * 3 + 0] = 0;
      out_w[i * 3 + 1] = 0;
      out_w[i * 3 + 2] = 0;
      for (int d = 0; d < D; ++d) out_feat[i * D + d] = 0;
      continue;
    }
    for (int k = (int)(i - bp * kh); k < n; k += kh) {
      const T dk = depth[row + k];
      const int64_t f = face_idx[row + k];
      int r = 0;
      for (int j = 0; j < n; ++j) {
        const T dj = depth[row + j];
        r += (dj > dk) | ((dj == dk) & (face_idx[row + j] < f));
      }
      const T w0 = w0a[row + k], w1 = w1a[row + k];
      const T w2 = (T)1 - (w0 + w1);
      const size_t o = row + r;
      out_idx[o] = f;
      out_w[o * 3 + 0] = w0;
      out_w[o * 3 + 1] = w1;
      out_w[o * 3 + 2] = w2;
      const int b = (int)(bp / P);
      const T* c = feat + ((size_t)b * F + (size_t)f) * 3 * D;
      for (int d = 0; d < D; ++d) out_feat[o * D + d] = (w0 * c[d] + w1 * c[D + d]) + w2 * c[2 * D + d];
    }
  }
}

// ---- 8. backward -----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dt_backward_kernel(int B, int F, int P, int K, int D, const T* __restrict__ grad,
                                                          const int64_t* __restrict__ face_idx,
                                                          const T* __restrict__ weights, const T* __restrict__ fimg,
                                                          const T* __restrict__ feat, float eps, T* __restrict__ g_img,
                                                          T* __restrict__ g_feat) {
  const size_t total = (size_t)B * P * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int64_t f = face_idx[i];
    if (f < 0) continue;
    const int b = (int)(i / ((size_t)P * K));
    const size_t g = (size_t)b * F + (size_t)f;
    const T wa = weights[i * 3], wb = weights[i * 3 + 1], wc = weights[i * 3 + 2];
    const T* go = grad + i * D;
    T* gf = g_feat + g * 3 * D;
    for (int d = 0; d < D; ++d) {
      const T v = go[d];
      kamd_atomic_add(gf + d, v * wa);
      kamd_atomic_add(gf + D + d, v * wb);
      kamd_atomic_add(gf + 2 * D + d, v * wc);
    }
    const T* v = fimg + g * 6;
    const T ax = v[0], ay = v[1], bx = v[2], by = v[3], cx = v[4], cy = v[5];
    const T x0 = wa * ax + wb * bx + wc * cx;
    const T y0 = wa * ay + wb * by + wc * cy;
    const T m = bx - ax, p = by - ay, n = cx - ax, q = cy - ay, s = x0 - ax, t = y0 - ay;
    const T k1 = s * q - n * t;
    const T k2 = m * t - s * p;
    T k3 = m * q - n * p;
    k3 = (T)((double)k3 + copysign((double)eps, (double)k3));
    // d(w1 k3^2), d(w2 k3^2) w.r.t. (m, n, p, q, s, t) with dk1 = (0,-t,0,s,q,-n), dk2 = (t,0,-s,0,-p,m),
    // dk3 = (q,-p,-n,m,0,0), written term by term as deftet_cuda.cu:331-363 (the zero products are kept: 0 * inf)
    const T z = 0;
    const T dw1dm = z * k3 - q * k1, dw1dn = (-t) * k3 - (-p) * k1, dw1dp = z * k3 - (-n) * k1;
    const T dw1dq = s * k3 - m * k1, dw1ds = q * k3 - z * k1, dw1dt = (-n) * k3 - z * k1;
    const T dw2dm = t * k3 - q * k2, dw2dn = z * k3 - (-p) * k2, dw2dp = (-s) * k3 - (-n) * k2;
    const T dw2dq = z * k3 - m * k2, dw2ds = (-p) * k3 - z * k2, dw2dt = m * k3 - z * k2;
    const T dw1[6] = {-(dw1dm + dw1dn + dw1ds), -(dw1dp + dw1dq + dw1dt), dw1dm, dw1dp, dw1dn, dw1dq};
    const T dw2[6] = {-(dw2dm + dw2dn + dw2ds), -(dw2dp + dw2dq + dw2dt), dw2dm, dw2dp, dw2dn, dw2dq};
    T acc[6] = {0, 0, 0, 0, 0, 0};
    const T* c = feat + g * 3 * D;
    for (int d = 0; d < D; ++d) {
      const T c0 = c[d], c1 = c[D + d], c2 = c[2 * D + d];
      const T dldI = go[d] / (k3 * k3);
#pragma unroll
      for (int e = 0; e < 6; ++e) acc[e] += dldI * ((c1 - c0) * dw1[e] + (c2 - c0) * dw2[e]);
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) kamd_atomic_add(g_img + g * 6 + e, acc[e]);
  }
}

// ---- host side --------------------------------------------------------------------------------------------------
template <typename T>
int dt_search(hipStream_t st, int B, int F, int P, int K, const T* fz, const T* fimg, const T* fbb, const T* pix,
              const T* range, float eps, int64_t* face_idx, T* depth, T* w0, T* w1, int* hit_count, int sort_rows, void* ws,
              size_t ws_bytes) {
  if (B <= 0 || P <= 0) return 0;
  if ((size_t)B * P >= (1ull << 31) || (size_t)B * (F > 0 ? F : 0) >= (1ull << 31)) return (int)hipErrorInvalidValue;
  const DtWs w = dt_ws(ws, B, F > 0 ? F : 0, P, (int)(sizeof(T) / 4));
  if (ws == nullptr || ws_bytes < w.total_words * 4) return (int)hipErrorInvalidValue;
  kamd::ProfScope prof_(kamd::K_DEFTET_FORWARD, st);
  KAMD_CHECK(kamd_zero_async(ws, w.zero_words * 4, st));
  DtOut<T> o;
  o.face_idx = face_idx;
  o.depth = depth;
  o.w0 = w0;
  o.w1 = w1;
  const int nblk = kamd_cdiv(P, DT_THREADS);
  hipLaunchKernelGGL(dt_extent_kernel<T>, dim3(DT_NB, B), dim3(DT_EXT_THREADS), 0, st, P, pix, w.part);
  hipLaunchKernelGGL(dt_count_kernel<T>, dim3(nblk, B), dim3(DT_THREADS), 0, st, P, pix, w);
  {  // the block sums live in the (not yet used) overflow list
    const int nsb = kamd_cdiv(w.nc, 1024);
    hipLaunchKernelGGL(dt_scan_sums_kernel, dim3(nsb, B), dim3(1024), 0, st, w, w.over);
    hipLaunchKernelGGL(dt_scan_apply_kernel, dim3(nsb, B), dim3(1024), 0, st, w, (const int*)w.over);
  }
  hipLaunchKernelGGL(dt_scatter_kernel<T>, dim3(nblk, B), dim3(DT_THREADS), 0, st, P, pix, range, w);
  if (F > 0) {
    hipLaunchKernelGGL(dt_face_kernel<T>, dim3(kamd_cdiv(F, DT_THREADS), B), dim3(DT_THREADS), 0, st, F, P, K, fz, fimg, fbb,
                       eps, w, o);
    hipLaunchKernelGGL((dt_big_face_kernel<T, 64>), dim3(KAMD_NUM_CU * 8), dim3(DT_THREADS), 0, st, F, P, K, fz, fimg, fbb,
                       eps, w, o);
    hipLaunchKernelGGL((dt_big_face_kernel<T, DT_THREADS>), dim3(KAMD_NUM_CU * 4), dim3(DT_THREADS), 0, st, F, P, K, fz,
                       fimg, fbb, eps, w, o);
  }
  const size_t BP = (size_t)B * P;
  hipLaunchKernelGGL(dt_order_rows_kernel<T>, dim3(kamd_cdiv(BP, 256)), dim3(256), 0, st, BP, K, o, w, hit_count, sort_rows);
  if (F > 0)
    hipLaunchKernelGGL(dt_overflow_kernel<T>, dim3(KAMD_NUM_CU * 2), dim3(DT_THREADS), 0, st, F, P, K, fz, fimg, fbb, pix,
                       range, eps, w, o, hit_count);
  KAMD_RETURN_LAST_ERROR();
}

inline unsigned dt_grid_for(size_t n) {
  size_t blocks = (n + 255) / 256;
  if (blocks > (size_t)KAMD_NUM_CU * 32) blocks = (size_t)KAMD_NUM_CU * 32;
  return (unsigned)(blocks ? blocks : 1);
}

template <typename T>
int dt_forward(hipStream_t st, int B, int F, int P, int K, const T* fz, const T* fimg, const T* fbb, const T* pix,
               const T* range, float eps, int64_t* face_idx, T* depth, T* w0, T* w1, void* ws, size_t ws_bytes) {
  const size_t n = (size_t)B * P * K;
  if (n == 0) return 0;
  hipLaunchKernelGGL(dt_fill_kernel<T>, dim3(dt_grid_for(n)), dim3(256), 0, st, n, face_idx, depth, w0, w1);
  return dt_search<T>(st, B, F, P, K, fz, fimg, fbb, pix, range, eps, face_idx, depth, w0, w1, nullptr, 1, ws, ws_bytes);
}

template <typename T>
int dt_forward_fused(hipStream_t st, int B, int F, int P, int K, int D, const T* fz, const T* fimg, const T* fbb,
                     const T* pix, const T* range, const T* feat, float eps, int64_t* tmp_idx, T* tmp_depth, T* tmp_w0,
                     T* tmp_w1, int* hit_count, int64_t* out_idx, T* out_w, T* out_feat, void* ws, size_t ws_bytes) {
  const size_t n = (size_t)B * P * K;
  if (n == 0) return 0;
  // rows stay in arrival order: the depth sort below ranks by (depth, face number) and does not need mesh order
  KAMD_CHECK(dt_search<T>(st, B, F, P, K, fz, fimg, fbb, pix, range, eps, tmp_idx, tmp_depth, tmp_w0, tmp_w1, hit_count, 0,
                          ws, ws_bytes));
  kamd::ProfScope prof_(kamd::K_DEFTET_SORT, st);
  if (n < ((size_t)1 << 22)) {  // a few launches cost more than the strided default stores they avoid
    hipLaunchKernelGGL((dt_sort_interp_kernel<T, 0>), dim3(dt_grid_for(n)), dim3(256), 0, st, B, F, P, K, D, tmp_idx,
                       tmp_depth, tmp_w0, tmp_w1, hit_count, feat, out_idx, out_w, out_feat);
  } else {
    KAMD_CHECK(dt_fill_bytes(out_idx, n * 8, 0xFF, st));  // -1
    KAMD_CHECK(dt_fill_bytes(out_w, n * 3 * sizeof(T), 0, st));
    KAMD_CHECK(dt_fill_bytes(out_feat, n * (size_t)D * sizeof(T), 0, st));
    hipLaunchKernelGGL((dt_sort_interp_kernel<T, 4>), dim3(dt_grid_for((size_t)B * P * 4)), dim3(256), 0, st, B, F, P, K, D,
                       tmp_idx, tmp_depth, tmp_w0, tmp_w1, hit_count, feat, out_idx, out_w, out_feat);
  }
  KAMD_RETURN_LAST_ERROR();
}

// The same arithmetic with the number of features known at compile time, and the results leaving through per-wavefront LDS
// rows: a hit's 3 DT feature gradients and its 6 image-coordinate gradients then sit in consecutive LANES of the atomic
// instructions -- two requests per hit (one line of g_feat, one of g_img) instead of 3 DT + 6.  Global float atomics cost
// per request (a line touched by an instruction), ~60 ps chip-wide on MI355X, whatever the lanes in it (measured
// on the rasterizer's backward): the lane-per-value form above made 15 requests per hit at D = 3.
template <typename T, int DT>
__global__ __launch_bounds__(256) void dt_backward_staged_kernel(int B, int F, int P, int K, const T* __restrict__ grad,
                                                                 const int64_t* __restrict__ face_idx,
                                                                 const T* __restrict__ weights, const T* __restrict__ fimg,
                                                                 const T* __restrict__ feat, float eps, T* __restrict__ g_img,
                                                                 T* __restrict__ g_feat) {
  constexpr int D = DT, NF = 3 * DT, NV = NF + 6;
  __shared__ T s_val[4][64 * NV];
  __shared__ long long s_g[4][64];  // b * F + face of the lane's hit, -1: none
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t total = (size_t)B * P * K;
  const size_t rounds = (total + 255) / 256;
  for (size_t r = blockIdx.x; r < rounds; r += gridDim.x) {  // (uniform per workgroup; a wavefront never waits for another)
    const size_t i = r * 256 + threadIdx.x;
    const int64_t f = i < total ? face_idx[i] : -1;
    long long g = -1;
    T vals[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) vals[e] = 0;
    if (f >= 0) {
      const int b = (int)(i / ((size_t)P * K));
      g = (long long)b * F + (long long)f;
      const T wa = weights[i * 3], wb = weights[i * 3 + 1], wc = weights[i * 3 + 2];
      const T* go = grad + i * D;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T v = go[d];
        vals[d] = v * wa;
        vals[D + d] = v * wb;
        vals[2 * D + d] = v * wc;
      }
      const T* v = fimg + (size_t)g * 6;
      const T ax = v[0], ay = v[1], bx = v[2], by = v[3], cx = v[4], cy = v[5];
      const T x0 = wa * ax + wb * bx + wc * cx;
      const T y0 = wa * ay + wb * by + wc * cy;
      const T m = bx - ax, p = by - ay, n = cx - ax, q = cy - ay, s = x0 - ax, t = y0 - ay;
      const T k1 = s * q - n * t;
      const T k2 = m * t - s * p;
      T k3 = m * q - n * p;
      k3 = (T)((double)k3 + copysign((double)eps, (double)k3));
      const T z = 0;
      const T dw1dm = z * k3 - q * k1, dw1dn = (-t) * k3 - (-p) * k1, dw1dp = z * k3 - (-n) * k1;
      const T dw1dq = s * k3 - m * k1, dw1ds = q * k3 - z * k1, dw1dt = (-n) * k3 - z * k1;
      const T dw2dm = t * k3 - q * k2, dw2dn = z * k3 - (-p) * k2, dw2dp = (-s) * k3 - (-n) * k2;
      const T dw2dq = z * k3 - m * k2, dw2ds = (-p) * k3 - z * k2, dw2dt = m * k3 - z * k2;
      const T dw1[6] = {-(dw1dm + dw1dn + dw1ds), -(dw1dp + dw1dq + dw1dt), dw1dm, dw1dp, dw1dn, dw1dq};
      const T dw2[6] = {-(dw2dm + dw2dn + dw2ds), -(dw2dp + dw2dq + dw2dt), dw2dm, dw2dp, dw2dn, dw2dq};
      const T* c = feat + (size_t)g * 3 * D;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const T c0 = c[d], c1 = c[D + d], c2 = c[2 * D + d];
        const T dldI = go[d] / (k3 * k3);
#pragma unroll
        for (int e = 0; e < 6; ++e) vals[NF + e] += dldI * ((c1 - c0) * dw1[e] + (c2 - c0) * dw2[e]);
      }
    }
    // (wavefront-level ordering is all that is needed: every wavefront owns its rows)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the wavefront's hits, compacted (a pixel's K slots are mostly empty: a wavefront holds a handful of hits)
    const unsigned long long act = __ballot(g >= 0);
    const int n_act = __popcll(act);
    if (g >= 0) {
      const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u));
      s_g[wave][pos] = g;
#pragma unroll
      for (int e = 0; e < NV; ++e) s_val[wave][pos * NV + e] = vals[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = lane; j < n_act * NV; j += 64) {
      const int q = j / NV, e = j - q * NV;
      const long long gq = s_g[wave][q];
      const T v = s_val[wave][j];
      if (e < NF)
        kamd_atomic_add(g_feat + (size_t)gq * NF + e, v);
      else
        kamd_atomic_add(g_img + (size_t)gq * 6 + (e - NF), v);
    }
  }
}

template <typename T>
int dt_backward(hipStream_t st, int B, int F, int P, int K, int D, const T* grad, const int64_t* face_idx,
                const T* weights, const T* fimg, const T* feat, float eps, T* g_img, T* g_feat) {
  const size_t n = (size_t)B * P * K;
  if (n == 0 || F <= 0) return 0;
  kamd::ProfScope prof_(kamd::K_DEFTET_BACKWARD, st);
  static const bool staged = kamd_env_int("KAMD_DEFTET_BWD_STAGED", 1) == 1;  // (2: a lane per value, for A/B runs)
#define KAMD_DTB(DT)                                                                                                      \
  hipLaunchKernelGGL((dt_backward_staged_kernel<T, DT>), dim3(dt_grid_for(n)), dim3(256), 0, st, B, F, P, K, grad, face_idx, \
                     weights, fimg, feat, eps, g_img, g_feat)
  if (staged && D >= 1 && D <= 4 && sizeof(T) == 4) {
    switch (D) {
      case 1: KAMD_DTB(1); break;
      case 2: KAMD_DTB(2); break;
      case 3: KAMD_DTB(3); break;
      default: KAMD_DTB(4); break;
    }
  } else {
    hipLaunchKernelGGL(dt_backward_kernel<T>, dim3(dt_grid_for(n)), dim3(256), 0, st, B, F, P, K, D, grad, face_idx,
                       weights, fimg, feat, eps, g_img, g_feat);
  }
#undef KAMD_DTB
  KAMD_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {
size_t kamd_deftet_forward_workspace(int B, int F, int P, int elem_size) {
  if (B <= 0 || P <= 0) return 0;
  return dt_ws(nullptr, B, F > 0 ? F : 0, P, elem_size / 4).total_words * 4;
}

#define KAMD_DEFTET_ENTRY(SFX, T)                                                                                       \
  int kamd_deftet_sparse_render_forward_##SFX(void* stream, int B, int F, int P, int K, const T* face_vertices_z,       \
                                              const T* face_vertices_image, const T* face_bboxes, const T* pixel_coords, \
                                              const T* pixel_depth_ranges, float eps, int64_t* face_idx,                 \
                                              T* pixel_depths, T* w0, T* w1, void* workspace, size_t workspace_bytes) {  \
    return dt_forward<T>((hipStream_t)stream, B, F, P, K, face_vertices_z, face_vertices_image, face_bboxes,            \
                         pixel_coords, pixel_depth_ranges, eps, face_idx, pixel_depths, w0, w1, workspace,              \
                         workspace_bytes);                                                                              \
  }                                                                                                                     \
  int kamd_deftet_sparse_render_forward_fused_##SFX(                                                                    \
      void* stream, int B, int F, int P, int K, int D, const T* face_vertices_z, const T* face_vertices_image,          \
      const T* face_bboxes, const T* pixel_coords, const T* pixel_depth_ranges, const T* face_features, float eps,      \
      int64_t* tmp_face_idx, T* tmp_depths, T* tmp_w0, T* tmp_w1, int32_t* hit_count, int64_t* sorted_face_idx,         \
      T* weights, T* interpolated_features, void* workspace, size_t workspace_bytes) {                                  \
    return dt_forward_fused<T>((hipStream_t)stream, B, F, P, K, D, face_vertices_z, face_vertices_image, face_bboxes,   \
                               pixel_coords, pixel_depth_ranges, face_features, eps, tmp_face_idx, tmp_depths, tmp_w0,  \
                               tmp_w1, hit_count, sorted_face_idx, weights, interpolated_features, workspace,           \
                               workspace_bytes);                                                                        \
  }                                                                                                                     \
  int kamd_deftet_sparse_render_backward_##SFX(void* stream, int B, int F, int P, int K, int D,                         \
                                               const T* grad_interpolated_features, const int64_t* face_idx,            \
                                               const T* weights, const T* face_vertices_image,                          \
                                               const T* face_features, float eps, T* grad_face_vertices_image,          \
                                               T* grad_face_features) {                                                 \
    return dt_backward<T>((hipStream_t)stream, B, F, P, K, D, grad_interpolated_features, face_idx, weights,            \
                          face_vertices_image, face_features, eps, grad_face_vertices_image, grad_face_features);       \
  }
KAMD_DEFTET_ENTRY(f32, float)
KAMD_DEFTET_ENTRY(f64, double)
#undef KAMD_DEFTET_ENTRY
}  // extern "C"
