// DefTet sparse renderer: multi-hit rasterization of FREE pixel coordinates.
//
// Replaces kaolin/csrc/render/mesh/deftet_cuda.cu (forward :31-186, backward :240-404) and the torch chain that
// follows the forward operator in kaolin/render/mesh/deftet.py:297-311 (argsort by depth, 3 gathers, pad, stack, sum).
//
// The reference visits every face for every pixel (P x F box tests).  Here:
//   1. pixels are counting-sorted into a 64 x 64 grid over their own extent (two LDS-histogram passes, one global
//      atomic per (workgroup, non-empty cell)), so a workgroup owns 256 pixels that lie close together;
//   2. a workgroup streams the face boxes once (16-byte loads, thread = face), keeps the faces whose box can contain
//      one of ITS pixels, and compacts them IN MESH ORDER (ballot + prefix) into an LDS list together with their
//      vertices;
//   3. lane = pixel walks that short list in order: one ds_read_b128 per box, a wave-uniform skip when no lane is
//      inside, the reference's arithmetic for the survivors, hits appended to the pixel's row while fewer than knum.
// The order of the pixels never changes a result (every pixel is independent), only which faces share a workgroup.
// Arithmetic follows the reference expression by expression (-ffp-contract=off): integer outputs are bit-exact vs
// oracle/deftet_oracle.inc, floats equal.
#include "common.h"
#include "profile.h"
#include "tile_bins.h"
#include "../../include/kaolin_amd.h"

namespace {
using kamd::Box4;

constexpr int DT_THREADS = 256;
constexpr int DT_WAVES = DT_THREADS / 64;
constexpr int DT_G = 64;               // pixel-sort cells per axis
constexpr int DT_NC = DT_G * DT_G;     // 4096 cells
constexpr int DT_NB = 64;              // extent partials per batch item
constexpr int DT_PPB = 4096;           // pixels per workgroup in the two histogram passes
constexpr int DT_CAP = 512;            // LDS face list capacity (flushed before a 256-face chunk could overflow it)

// workspace (4-byte words) per batch item: extent partials | cell starts (NC + 1) | cell cursors (NC) | order (P)
__host__ __device__ inline size_t dt_ws_words(int P) { return (size_t)DT_NB * 4 + (DT_NC + 1) + DT_NC + (size_t)P; }
struct DtWs {
  float* part;
  int* start;
  int* cursor;
  int* order;
};
__host__ __device__ inline DtWs dt_ws(void* base, int P, int b) {
  int* w = (int*)base + (size_t)b * dt_ws_words(P);
  DtWs r;
  r.part = (float*)w;
  r.start = w + DT_NB * 4;
  r.cursor = r.start + DT_NC + 1;
  r.order = r.cursor + DT_NC;
  return r;
}

// ---- 1. extent of the finite pixel coordinates (partials) ------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(DT_THREADS) void dt_extent_kernel(int P, const T* __restrict__ pix, void* ws) {
  __shared__ float s[4][DT_THREADS];
  const int b = blockIdx.y;
  const T* X = pix + (size_t)b * P * 2;
  float lo0 = INFINITY, lo1 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY;
  for (int i = blockIdx.x * DT_THREADS + threadIdx.x; i < P; i += gridDim.x * DT_THREADS) {
    const float x = (float)X[(size_t)i * 2], y = (float)X[(size_t)i * 2 + 1];
    if (isfinite(x)) {
      lo0 = fminf(lo0, x);
      hi0 = fmaxf(hi0, x);
    }
    if (isfinite(y)) {
      lo1 = fminf(lo1, y);
      hi1 = fmaxf(hi1, y);
    }
  }
  s[0][threadIdx.x] = lo0;
  s[1][threadIdx.x] = lo1;
  s[2][threadIdx.x] = hi0;
  s[3][threadIdx.x] = hi1;
  __syncthreads();
  for (int d = DT_THREADS / 2; d >= 1; d >>= 1) {
    if (threadIdx.x < d) {
      s[0][threadIdx.x] = fminf(s[0][threadIdx.x], s[0][threadIdx.x + d]);
      s[1][threadIdx.x] = fminf(s[1][threadIdx.x], s[1][threadIdx.x + d]);
      s[2][threadIdx.x] = fmaxf(s[2][threadIdx.x], s[2][threadIdx.x + d]);
      s[3][threadIdx.x] = fmaxf(s[3][threadIdx.x], s[3][threadIdx.x + d]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) dt_ws(ws, P, b).part[blockIdx.x * 4 + threadIdx.x] = s[threadIdx.x][0];
}

struct DtGrid {
  float lo[2], inv[2];
};
// every consumer re-reduces the 64 partials (1 KB from L2); called by the first wave, result broadcast through LDS
__device__ __forceinline__ void dt_grid_setup(const float* __restrict__ part, DtGrid* g) {
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
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float l = lo[a], size = (hi[a] - lo[a]) / (float)DT_G;
        if (!(lo[a] <= hi[a])) l = 0.f;
        if (!(size > 0.f) || !isfinite(size)) size = 1.f;  // degenerate extent: one slab holds everything
        g->lo[a] = l;
        g->inv[a] = 1.f / size;
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ int dt_axis_cell(float v, float lo, float inv) {
  const float t = (v - lo) * inv;
  return (t >= 0.f) ? (t < (float)DT_G ? (int)t : DT_G - 1) : 0;  // NaN -> 0, +-inf clamped
}
template <typename T>
__device__ __forceinline__ int dt_cell(const DtGrid& g, const T* __restrict__ X, int i) {
  return dt_axis_cell((float)X[(size_t)i * 2 + 1], g.lo[1], g.inv[1]) * DT_G +
         dt_axis_cell((float)X[(size_t)i * 2], g.lo[0], g.inv[0]);
}

// ---- 2. cell histogram: LDS counts per workgroup, one global atomic per (workgroup, non-empty cell) ----------
template <typename T>
__global__ __launch_bounds__(DT_THREADS) void dt_count_kernel(int P, const T* __restrict__ pix, void* ws) {
  __shared__ int s_hist[DT_NC];
  __shared__ DtGrid s_g;
  const int b = blockIdx.y;
  const DtWs w = dt_ws(ws, P, b);
  for (int c = threadIdx.x; c < DT_NC; c += DT_THREADS) s_hist[c] = 0;
  dt_grid_setup(w.part, &s_g);
  const T* X = pix + (size_t)b * P * 2;
  const int i0 = blockIdx.x * DT_PPB, i1 = min(P, i0 + DT_PPB);
  for (int i = i0 + threadIdx.x; i < i1; i += DT_THREADS) atomicAdd(&s_hist[dt_cell<T>(s_g, X, i)], 1);
  __syncthreads();
  for (int c = threadIdx.x; c < DT_NC; c += DT_THREADS)
    if (s_hist[c]) atomicAdd(&w.start[c], s_hist[c]);
}

// ---- 3. exclusive scan of the 4096 counts (one workgroup per batch item); cursors start at the cell starts --------
__global__ __launch_bounds__(1024) void dt_scan_kernel(int P, void* ws) {
  __shared__ int s_wave[16];
  const DtWs w = dt_ws(ws, P, blockIdx.x);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int v[4], sum = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = w.start[t * 4 + k];
    sum += v[k];
  }
  int inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int off = inc - sum;
  for (int k = 0; k < wave; ++k) off += s_wave[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    w.start[t * 4 + k] = off;
    w.cursor[t * 4 + k] = off;
    off += v[k];
  }
  if (t == 1023) w.start[DT_NC] = off;
}

// ---- 4. scatter: rank inside the workgroup from the LDS atomic, one global reservation per non-empty cell ------
template <typename T>
__global__ __launch_bounds__(DT_THREADS) void dt_scatter_kernel(int P, const T* __restrict__ pix, void* ws) {
  __shared__ int s_hist[DT_NC];
  __shared__ DtGrid s_g;
  const int b = blockIdx.y;
  const DtWs w = dt_ws(ws, P, b);
  for (int c = threadIdx.x; c < DT_NC; c += DT_THREADS) s_hist[c] = 0;
  dt_grid_setup(w.part, &s_g);
  const T* X = pix + (size_t)b * P * 2;
  const int i0 = blockIdx.x * DT_PPB, i1 = min(P, i0 + DT_PPB);
  int cell[DT_PPB / DT_THREADS], rank[DT_PPB / DT_THREADS];
#pragma unroll
  for (int k = 0; k < DT_PPB / DT_THREADS; ++k) {
    const int i = i0 + k * DT_THREADS + threadIdx.x;
    cell[k] = -1;
    if (i < i1) {
      cell[k] = dt_cell<T>(s_g, X, i);
      rank[k] = atomicAdd(&s_hist[cell[k]], 1);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < DT_NC; c += DT_THREADS) {
    const int n = s_hist[c];
    if (n) s_hist[c] = atomicAdd(&w.cursor[c], n);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < DT_PPB / DT_THREADS; ++k)
    if (cell[k] >= 0) w.order[s_hist[cell[k]] + rank[k]] = i0 + k * DT_THREADS + threadIdx.x;
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

// ---- 5. the search ---------------------------------------------------------------------------------------------
template <typename T>
struct DtFace {
  T ax, ay, bx, by, cx, cy, az, bz, cz;
};

template <typename T>
__global__ __launch_bounds__(DT_THREADS) void dt_forward_kernel(int F, int P, int K, const T* __restrict__ fz,
                                                                const T* __restrict__ fimg, const T* __restrict__ fbb,
                                                                const T* __restrict__ pix, const T* __restrict__ range,
                                                                float eps, const void* ws, int sorted,
                                                                int64_t* __restrict__ face_idx, T* __restrict__ depth,
                                                                T* __restrict__ w0a, T* __restrict__ w1a,
                                                                int* __restrict__ hit_count) {
  __shared__ Box4<T> s_box[DT_CAP];
  __shared__ DtFace<T> s_face[DT_CAP];
  __shared__ int s_id[DT_CAP];
  __shared__ int s_wcnt[2][DT_WAVES];
  __shared__ T s_ext[4][DT_WAVES];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = blockIdx.x * DT_THREADS + tid;
  const bool live = slot < P;
  int p = 0;
  if (live) p = sorted ? dt_ws(const_cast<void*>(ws), P, b).order[slot] : slot;
  const size_t bp = (size_t)b * P + p;
  T x0 = 0, y0 = 0, dmin = 0, dmax = 0;
  if (live) {
    x0 = pix[bp * 2];
    y0 = pix[bp * 2 + 1];
    dmin = range[bp * 2];
    dmax = range[bp * 2 + 1];
  }
  // closed extent of this workgroup's finite pixel coordinates (a non-finite coordinate can never be inside a box)
  {
    T lx = INFINITY, ly = INFINITY, hx = -INFINITY, hy = -INFINITY;
    if (live && isfinite(x0) && isfinite(y0)) {
      lx = hx = x0;
      ly = hy = y0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lx = fmin(lx, __shfl_xor(lx, d, 64));
      ly = fmin(ly, __shfl_xor(ly, d, 64));
      hx = fmax(hx, __shfl_xor(hx, d, 64));
      hy = fmax(hy, __shfl_xor(hy, d, 64));
    }
    if (lane == 0) {
      s_ext[0][wave] = lx;
      s_ext[1][wave] = ly;
      s_ext[2][wave] = hx;
      s_ext[3][wave] = hy;
    }
  }
  __syncthreads();
  T ex0 = s_ext[0][0], ey0 = s_ext[1][0], ex1 = s_ext[2][0], ey1 = s_ext[3][0];
#pragma unroll
  for (int k = 1; k < DT_WAVES; ++k) {
    ex0 = fmin(ex0, s_ext[0][k]);
    ey0 = fmin(ey0, s_ext[1][k]);
    ex1 = fmax(ex1, s_ext[2][k]);
    ey1 = fmax(ey1, s_ext[3][k]);
  }
  const T norm_sign_eps = (T)(float)(double)eps;  // |copysignf((double)eps, .)|: eps as FLOAT (deftet_cuda.cu:135)
  const size_t row = bp * K;
  int n_hit = 0, n_list = 0;
  const Box4<T>* boxes = reinterpret_cast<const Box4<T>*>(fbb) + (size_t)b * F;
  // box loads run two chunks ahead of their use: the loop below has no other global access outside a flush
  Box4<T> fb = {0, 0, 0, 0}, fb_n1 = {0, 0, 0, 0}, fb_n2 = {0, 0, 0, 0};
  if (tid < F) fb_n1 = boxes[tid];
  if (tid + DT_THREADS < F) fb_n2 = boxes[tid + DT_THREADS];
  int it = 0;
  for (int base = 0; base < F; base += DT_THREADS, it ^= 1) {
    const int f = base + tid;
    fb = fb_n1;
    fb_n1 = fb_n2;
    if (f + 2 * DT_THREADS < F) fb_n2 = boxes[f + 2 * DT_THREADS];
    // a pixel is inside when min <= x < max: impossible when max <= (smallest x) or min > (largest x)
    const bool keep = (f < F) & !((fb.x1 <= ex0) | (fb.x0 > ex1) | (fb.y1 <= ey0) | (fb.y0 > ey1));
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_wcnt[it][wave] = __popcll(m);
    __syncthreads();  // the only barrier of a chunk that is not flushed (s_wcnt is double-buffered)
    int woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < DT_WAVES; ++k) {
      if (k < wave) woff += s_wcnt[it][k];
      total += s_wcnt[it][k];
    }
    if (keep) {
      const int pos = n_list + woff + __popcll(m & ((1ull << lane) - 1ull));
      s_box[pos] = fb;
      s_id[pos] = f;
    }
    n_list += total;
    if (n_list > DT_CAP - DT_THREADS || base + DT_THREADS >= F) {
      __syncthreads();  // list entries visible
      // vertices of the listed faces: one round of independent loads per flush, thread = entry
      for (int k = tid; k < n_list; k += DT_THREADS) {
        const size_t g = (size_t)b * F + s_id[k];
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
        s_face[k] = fc;
      }
      __syncthreads();
      for (int k = 0; k < n_list; ++k) {
        const Box4<T> bb = s_box[k];
        const bool inside = live & (n_hit < K) & (x0 >= bb.x0) & (x0 < bb.x1) & (y0 >= bb.y0) & (y0 < bb.y1);
        if (!__any(inside)) continue;
        if (inside) {
          const DtFace<T> fc = s_face[k];
          const T aex = fc.ax - x0, aey = fc.ay - y0, bex = fc.bx - x0, bey = fc.by - y0;
          const T cex = fc.cx - x0, cey = fc.cy - y0;
          const T u0 = bex * cey - bey * cex;
          const T u1 = cex * aey - cey * aex;
          const T u2 = aex * bey - aey * bex;
          const T norm = u0 + u1 + u2;
          const T norm_eps = (T)copysignf((float)norm_sign_eps, (float)norm);
          const T w0 = u0 / (norm + norm_eps), w1 = u1 / (norm + norm_eps), w2 = u2 / (norm + norm_eps);
          if (w0 >= 0. && w1 >= 0. && w2 >= 0.) {
            const T d = w0 * fc.az + w1 * fc.bz + w2 * fc.cz;
            if (d < dmax && d >= dmin) {
              face_idx[row + n_hit] = s_id[k];
              depth[row + n_hit] = d;
              w0a[row + n_hit] = w0;
              w1a[row + n_hit] = w1;
              ++n_hit;
            }
          }
        }
      }
      n_list = 0;
      __syncthreads();  // everyone is done reading the list before it is refilled
    }
  }
  if (live && hit_count) hit_count[bp] = n_hit;
}

// ---- 6. depth sort + weights + interpolation (deftet.py:297-311) ------------------------------------------------
// thread = (pixel, k): rank of entry k among the pixel's n hits by (depth descending, then position) = its output
// slot; slots >= n get the defaults.  Equal depths keep mesh order (torch.argsort in the reference leaves it open).
template <typename T>
__global__ __launch_bounds__(256) void dt_sort_interp_kernel(int B, int F, int P, int K, int D,
                                                             const int64_t* __restrict__ face_idx,
                                                             const T* __restrict__ depth, const T* __restrict__ w0a,
                                                             const T* __restrict__ w1a, const int* __restrict__ hit_count,
                                                             const T* __restrict__ feat, int64_t* __restrict__ out_idx,
                                                             T* __restrict__ out_w, T* __restrict__ out_feat) {
  const size_t total = (size_t)B * P * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t bp = i / K;
    const int k = (int)(i - bp * K);
    const int n = hit_count[bp];
    if (k >= n) {
      out_idx[i] = -1;
      out_w[i * 3 + 0] = 0;
      out_w[i * 3 + 1] = 0;
      out_w[i * 3 + 2] = 0;
      for (int d = 0; d < D; ++d) out_feat[i * D + d] = 0;
      continue;
    }
    const size_t row = bp * K;
    const T dk = depth[i];
    int r = 0;
    for (int j = 0; j < n; ++j) {
      const T dj = depth[row + j];
      r += (dj > dk) | ((dj == dk) & (j < k));
    }
    const int64_t f = face_idx[i];
    const T w0 = w0a[i], w1 = w1a[i];
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

// ---- 7. backward -----------------------------------------------------------------------------------------------
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
              const T* range, float eps, int64_t* face_idx, T* depth, T* w0, T* w1, int* hit_count, void* ws,
              size_t ws_bytes) {
  if (B <= 0 || P <= 0) return 0;
  // a single workgroup per batch item gains nothing from sorted pixels
  const bool sorted = P > DT_THREADS && ws != nullptr && ws_bytes >= (size_t)B * dt_ws_words(P) * 4;
  if (P > DT_THREADS && !sorted) return (int)hipErrorInvalidValue;
  kamd::ProfScope prof_(kamd::K_DEFTET_FORWARD, st);
  if (sorted) {
    // clear the cell counts (the partials / cursors / order are fully written)
    for (int b = 0; b < B; ++b) KAMD_CHECK(hipMemsetAsync(dt_ws(ws, P, b).start, 0, (DT_NC + 1) * sizeof(int), st));
    const int nblk = kamd_cdiv(P, DT_PPB);
    hipLaunchKernelGGL(dt_extent_kernel<T>, dim3(DT_NB, B), dim3(DT_THREADS), 0, st, P, pix, ws);
    hipLaunchKernelGGL(dt_count_kernel<T>, dim3(nblk, B), dim3(DT_THREADS), 0, st, P, pix, ws);
    hipLaunchKernelGGL(dt_scan_kernel, dim3(B), dim3(1024), 0, st, P, ws);
    hipLaunchKernelGGL(dt_scatter_kernel<T>, dim3(nblk, B), dim3(DT_THREADS), 0, st, P, pix, ws);
  }
  hipLaunchKernelGGL(dt_forward_kernel<T>, dim3(kamd_cdiv(P, DT_THREADS), B), dim3(DT_THREADS), 0, st, F, P, K, fz, fimg,
                     fbb, pix, range, eps, (const void*)ws, sorted ? 1 : 0, face_idx, depth, w0, w1, hit_count);
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
  return dt_search<T>(st, B, F, P, K, fz, fimg, fbb, pix, range, eps, face_idx, depth, w0, w1, nullptr, ws, ws_bytes);
}

template <typename T>
int dt_forward_fused(hipStream_t st, int B, int F, int P, int K, int D, const T* fz, const T* fimg, const T* fbb,
                     const T* pix, const T* range, const T* feat, float eps, int64_t* tmp_idx, T* tmp_depth, T* tmp_w0,
                     T* tmp_w1, int* hit_count, int64_t* out_idx, T* out_w, T* out_feat, void* ws, size_t ws_bytes) {
  const size_t n = (size_t)B * P * K;
  if (n == 0) return 0;
  KAMD_CHECK(dt_search<T>(st, B, F, P, K, fz, fimg, fbb, pix, range, eps, tmp_idx, tmp_depth, tmp_w0, tmp_w1, hit_count,
                          ws, ws_bytes));
  kamd::ProfScope prof_(kamd::K_DEFTET_SORT, st);
  hipLaunchKernelGGL(dt_sort_interp_kernel<T>, dim3(dt_grid_for(n)), dim3(256), 0, st, B, F, P, K, D, tmp_idx, tmp_depth,
                     tmp_w0, tmp_w1, hit_count, feat, out_idx, out_w, out_feat);
  KAMD_RETURN_LAST_ERROR();
}

template <typename T>
int dt_backward(hipStream_t st, int B, int F, int P, int K, int D, const T* grad, const int64_t* face_idx,
                const T* weights, const T* fimg, const T* feat, float eps, T* g_img, T* g_feat) {
  const size_t n = (size_t)B * P * K;
  if (n == 0 || F <= 0) return 0;
  kamd::ProfScope prof_(kamd::K_DEFTET_BACKWARD, st);
  hipLaunchKernelGGL(dt_backward_kernel<T>, dim3(dt_grid_for(n)), dim3(256), 0, st, B, F, P, K, D, grad, face_idx,
                     weights, fimg, feat, eps, g_img, g_feat);
  KAMD_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {
size_t kamd_deftet_forward_workspace(int B, int P) {
  if (B <= 0 || P <= DT_THREADS) return 0;
  return (size_t)B * dt_ws_words(P) * 4;
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
