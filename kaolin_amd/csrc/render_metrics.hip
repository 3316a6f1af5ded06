// The steps either side of DIB-R in the training loop, fused: mask_iou and texture_mapping.
//
// mask_iou (kaolin/metrics/render.py:18-40) is five full-size torch kernels forward (product, sum, difference, two
// reductions) and as many backward; here the forward is ONE pass over both masks (per-workgroup partial sums of
// I = sum(l*r) and U = sum(l + r - l*r) in double, then a one-workgroup finish that also writes the loss) and the backward
// one elementwise pass: d loss / d l = -(1/B) * (r * (U + eps) - I * (1 - r)) / (U + eps)^2 -- it does not even read l.
//
// texture_mapping (kaolin/render/mesh/utils.py:23-76) is clamp, scale, flip, grid_sample, permute; here one gather
// kernel each way with grid_sample's own coordinate arithmetic (align_corners = False, border padding, nearest = round half
// to even, bilinear), writing the (B, N, C) layout directly.
//
// weighted_sum2: the linear loss sum(x1 * w1) + sum(x2 * w2) over two G-buffers of one render (image features and soft
// mask against fixed weights -- what gradient checks and benchmarks of a renderer back-propagate).  In torch that is two
// rocBLAS dots (two kernels each), an add, and two full-size products backward; here ONE pass over the four arrays forward
// (16-byte loads, per-workgroup partial sums in double, a one-workgroup finish) and ONE elementwise pass backward that
// writes both gradients g * w1, g * w2.
#include "common.h"
#include "profile.h"
#include "../../include/kaolin_amd.h"

namespace {

constexpr int MI_THREADS = 256;
constexpr int MI_GROUPS = 64;  // partial sums per batch item

template <typename T>
__global__ __launch_bounds__(MI_THREADS) void mask_iou_partial_kernel(long long P, const T* __restrict__ lhs,
                                                                      const T* __restrict__ rhs, double* __restrict__ partial) {
  __shared__ double s_i[MI_THREADS / 64], s_u[MI_THREADS / 64];
  const int b = blockIdx.y, g = blockIdx.x;
  const T* L = lhs + (size_t)b * P;
  const T* R = rhs + (size_t)b * P;
  double inter = 0, uni = 0;
  for (long long i = (long long)g * MI_THREADS + threadIdx.x; i < P; i += (long long)MI_GROUPS * MI_THREADS) {
    const T l = L[i], r = R[i];
    const T m = l * r;            // the reference forms the product and the sum in the masks' dtype
    inter += (double)m;
    uni += (double)((l + r) - m);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    inter += __shfl_xor(inter, d, 64);
    uni += __shfl_xor(uni, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_i[threadIdx.x >> 6] = inter;
    s_u[threadIdx.x >> 6] = uni;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, c = 0;
    for (int w = 0; w < MI_THREADS / 64; ++w) {
      a += s_i[w];
      c += s_u[w];
    }
    partial[((size_t)b * MI_GROUPS + g) * 2 + 0] = a;
    partial[((size_t)b * MI_GROUPS + g) * 2 + 1] = c;
  }
}

// one workgroup: sums[b] = {I_b, U_b} and loss = 1 - mean_b(I_b / (U_b + 1e-10))
template <typename T>
__global__ __launch_bounds__(64) void mask_iou_finish_kernel(int B, const double* __restrict__ partial, double* __restrict__ sums,
                                                             T* __restrict__ loss) {
  double acc = 0;
  for (int b = 0; b < B; ++b) {
    double a = partial[((size_t)b * MI_GROUPS + threadIdx.x) * 2 + 0], c = partial[((size_t)b * MI_GROUPS + threadIdx.x) * 2 + 1];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      a += __shfl_xor(a, d, 64);
      c += __shfl_xor(c, d, 64);
    }
    if (threadIdx.x == 0) {
      sums[b * 2 + 0] = a;
      sums[b * 2 + 1] = c;
    }
    acc += (double)((T)a / ((T)c + (T)1e-10));   // per-item ratio in the masks' dtype, as the reference's torch ops
  }
  if (threadIdx.x == 0) *loss = (T)(1.0 - (double)(T)(acc / B));
}

template <typename T>
__global__ __launch_bounds__(256) void mask_iou_backward_kernel(int B, long long P, const T* __restrict__ grad_loss,
                                                                const T* __restrict__ other, const double* __restrict__ sums,
                                                                T* __restrict__ grad) {
  const int b = blockIdx.y;
  const double I = sums[b * 2 + 0], U = sums[b * 2 + 1] + 1e-10;
  const double scale = -(double)grad_loss[0] / ((double)B * U * U);
  const T* O = other + (size_t)b * P;
  T* G = grad + (size_t)b * P;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const double o = (double)O[i];
    G[i] = (T)(scale * (o * U - I * (1.0 - o)));
  }
}

// ---- texture mapping ---------------------------------------------------------------------------------------------------
// grid_sample's source index of a normalised coordinate c in [-1, 1], align_corners = False: ((c + 1) * size - 1) / 2,
// clipped to [0, size - 1] (padding_mode = 'border')
template <typename T>
__device__ __forceinline__ T tex_source_index(T c, int size) {
  T x = ((c + (T)1) * (T)size - (T)1) / (T)2;
  x = x < (T)0 ? (T)0 : x;
  x = x > (T)(size - 1) ? (T)(size - 1) : x;
  return x;
}
__device__ __forceinline__ float tex_rint(float x) { return nearbyintf(x); }
__device__ __forceinline__ double tex_rint(double x) { return nearbyint(x); }
__device__ __forceinline__ float tex_floor(float x) { return floorf(x); }
__device__ __forceinline__ double tex_floor(double x) { return floor(x); }

template <typename T, bool BILINEAR>
__global__ __launch_bounds__(256) void texture_mapping_forward_kernel(long long N, int C, int TH, int TW, const T* __restrict__ uv,
                                                                      const T* __restrict__ tex, T* __restrict__ out) {
  const int b = blockIdx.y;
  const T* texb = tex + (size_t)b * C * TH * TW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
    T u = uv[((size_t)b * N + i) * 2 + 0], v = uv[((size_t)b * N + i) * 2 + 1];
    u = u < (T)0 ? (T)0 : (u > (T)1 ? (T)1 : u);      // torch.clamp(uv, 0, 1) (a NaN stays NaN, as in torch)
    v = v < (T)0 ? (T)0 : (v > (T)1 ? (T)1 : v);
    const T gx = u * (T)2 - (T)1, gy = -(v * (T)2 - (T)1);
    const T sx = tex_source_index<T>(gx, TW), sy = tex_source_index<T>(gy, TH);
    T* o = out + ((size_t)b * N + i) * C;
    if (!BILINEAR) {
      const int ix = (int)tex_rint(sx), iy = (int)tex_rint(sy);
      const bool ok = ix >= 0 && ix < TW && iy >= 0 && iy < TH;  // (false only for NaN coordinates)
      for (int c = 0; c < C; ++c) o[c] = ok ? texb[((size_t)c * TH + iy) * TW + ix] : (T)0;
    } else {
      const T fx = tex_floor(sx), fy = tex_floor(sy);
      const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
      const T wx1 = sx - fx, wy1 = sy - fy, wx0 = (fx + (T)1) - sx, wy0 = (fy + (T)1) - sy;   // grid_sample's (x_se - x), (y_se - y)
      const T nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
      const bool in_x0 = x0 >= 0 && x0 < TW, in_x1 = x1 >= 0 && x1 < TW, in_y0 = y0 >= 0 && y0 < TH, in_y1 = y1 >= 0 && y1 < TH;
      for (int c = 0; c < C; ++c) {
        const T* pc = texb + (size_t)c * TH * TW;
        T acc = 0;
        if (in_x0 && in_y0) acc += pc[(size_t)y0 * TW + x0] * nw;
        if (in_x1 && in_y0) acc += pc[(size_t)y0 * TW + x1] * ne;
        if (in_x0 && in_y1) acc += pc[(size_t)y1 * TW + x0] * sw;
        if (in_x1 && in_y1) acc += pc[(size_t)y1 * TW + x1] * se;
        o[c] = acc;
      }
    }
  }
}

// gradient w.r.t. the texture (atomic scatter into a zeroed tensor) and, for bilinear sampling, w.r.t. the coordinates
template <typename T, bool BILINEAR>
__global__ __launch_bounds__(256) void texture_mapping_backward_kernel(long long N, int C, int TH, int TW, const T* __restrict__ uv,
                                                                       const T* __restrict__ tex, const T* __restrict__ grad_out,
                                                                       T* __restrict__ g_tex, T* __restrict__ g_uv) {
  const int b = blockIdx.y;
  const T* texb = tex + (size_t)b * C * TH * TW;
  T* gtb = g_tex ? g_tex + (size_t)b * C * TH * TW : nullptr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
    const T u_raw = uv[((size_t)b * N + i) * 2 + 0], v_raw = uv[((size_t)b * N + i) * 2 + 1];
    const T u = u_raw < (T)0 ? (T)0 : (u_raw > (T)1 ? (T)1 : u_raw), v = v_raw < (T)0 ? (T)0 : (v_raw > (T)1 ? (T)1 : v_raw);
    const T gx = u * (T)2 - (T)1, gy = -(v * (T)2 - (T)1);
    const T ux = ((gx + (T)1) * (T)TW - (T)1) / (T)2, uy = ((gy + (T)1) * (T)TH - (T)1) / (T)2;   // before the border clip
    const T sx = tex_source_index<T>(gx, TW), sy = tex_source_index<T>(gy, TH);
    const T* go = grad_out + ((size_t)b * N + i) * C;
    if (!BILINEAR) {
      const int ix = (int)tex_rint(sx), iy = (int)tex_rint(sy);
      if (gtb && ix >= 0 && ix < TW && iy >= 0 && iy < TH)
        for (int c = 0; c < C; ++c) kamd_atomic_add(gtb + ((size_t)c * TH + iy) * TW + ix, go[c]);
      if (g_uv) {
        g_uv[((size_t)b * N + i) * 2 + 0] = 0;
        g_uv[((size_t)b * N + i) * 2 + 1] = 0;
      }
    } else {
      const T fx = tex_floor(sx), fy = tex_floor(sy);
      const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
      const T wx1 = sx - fx, wy1 = sy - fy, wx0 = (fx + (T)1) - sx, wy0 = (fy + (T)1) - sy;
      const bool in_x0 = x0 >= 0 && x0 < TW, in_x1 = x1 >= 0 && x1 < TW, in_y0 = y0 >= 0 && y0 < TH, in_y1 = y1 >= 0 && y1 < TH;
      T gsx = 0, gsy = 0;
      for (int c = 0; c < C; ++c) {
        const T g = go[c];
        const T* pc = texb + (size_t)c * TH * TW;
        T* gc = gtb ? gtb + (size_t)c * TH * TW : nullptr;
        const T t00 = in_x0 && in_y0 ? pc[(size_t)y0 * TW + x0] : (T)0, t10 = in_x1 && in_y0 ? pc[(size_t)y0 * TW + x1] : (T)0;
        const T t01 = in_x0 && in_y1 ? pc[(size_t)y1 * TW + x0] : (T)0, t11 = in_x1 && in_y1 ? pc[(size_t)y1 * TW + x1] : (T)0;
        if (gc) {
          if (in_x0 && in_y0) kamd_atomic_add(gc + (size_t)y0 * TW + x0, g * wx0 * wy0);
          if (in_x1 && in_y0) kamd_atomic_add(gc + (size_t)y0 * TW + x1, g * wx1 * wy0);
          if (in_x0 && in_y1) kamd_atomic_add(gc + (size_t)y1 * TW + x0, g * wx0 * wy1);
          if (in_x1 && in_y1) kamd_atomic_add(gc + (size_t)y1 * TW + x1, g * wx1 * wy1);
        }
        gsx += g * ((t10 - t00) * wy0 + (t11 - t01) * wy1);
        gsy += g * ((t01 - t00) * wx0 + (t11 - t10) * wx1);
      }
      if (g_uv) {
        // chain: source index <- border clip <- * size / 2 <- (u * 2 - 1, flipped for v) <- clamp(uv, 0, 1) (gradient passes
        // inside the closed interval, as torch.clamp).  grid_sample's clip_coordinates_set_grad gives the borders THEMSELVES
        // gradient 0 (in <= 0 or in >= size - 1): the interval is open, e.g. TW = 2, u = 0.25 sits exactly on index 0
        const T cx = (ux > (T)0 && ux < (T)(TW - 1)) ? (T)1 : (T)0, cy = (uy > (T)0 && uy < (T)(TH - 1)) ? (T)1 : (T)0;
        const T ku = (u_raw >= (T)0 && u_raw <= (T)1) ? (T)1 : (T)0, kv = (v_raw >= (T)0 && v_raw <= (T)1) ? (T)1 : (T)0;
        g_uv[((size_t)b * N + i) * 2 + 0] = gsx * cx * ((T)TW / (T)2) * (T)2 * ku;
        g_uv[((size_t)b * N + i) * 2 + 1] = gsy * cy * ((T)TH / (T)2) * (T)(-2) * kv;
      }
    }
  }
}


// ---- weighted sum of two arrays -----------------------------------------------------------------------------------------
typedef unsigned int ws_u32x4 __attribute__((ext_vector_type(4)));
// streaming load: the arrays are far larger than the caches and read once (measured cold: 6.5 vs 6.0 TB/s, tools/bench_stream)
__device__ __forceinline__ uint4 ws_load_stream(const uint4* p) {
  const ws_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ws_u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
constexpr int WS_THREADS = 256;
constexpr int WS_GROUPS = 4096;  // partial sums (16 workgroups per CU)

// workgroup g of G takes the g-th contiguous share of the array's 16-byte chunks (DRAM pages and TLB entries are walked
// once, by one workgroup); the unaligned / tail scalars go to the last threads of the grid
template <typename T>
__device__ __forceinline__ double ws_dot_range(const T* __restrict__ x, const T* __restrict__ w, long long n) {
  constexpr int V = 16 / (int)sizeof(T);
  constexpr int U = 4;  // 16-byte load pairs in flight per thread
  double acc = 0;
  const bool wide = ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0;
  const long long nv = wide ? n / V : 0;
  const long long share = (nv + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * share, hi = lo + share < nv ? lo + share : nv;
  const uint4* xv = reinterpret_cast<const uint4*>(x);
  const uint4* wv = reinterpret_cast<const uint4*>(w);
  long long i = lo + threadIdx.x;
  for (; i + (U - 1) * WS_THREADS < hi; i += U * WS_THREADS) {
    uint4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = ws_load_stream(xv + i + u * WS_THREADS);
      b[u] = ws_load_stream(wv + i + u * WS_THREADS);
    }
    T part = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const T* pa = reinterpret_cast<const T*>(&a[u]);
      const T* pb = reinterpret_cast<const T*>(&b[u]);
#pragma unroll
      for (int k = 0; k < V; ++k) part += pa[k] * pb[k];
    }
    acc += (double)part;
  }
  for (; i < hi; i += WS_THREADS) {
    const uint4 a = xv[i], b = wv[i];
    const T* pa = reinterpret_cast<const T*>(&a);
    const T* pb = reinterpret_cast<const T*>(&b);
    T part = 0;
#pragma unroll
    for (int k = 0; k < V; ++k) part += pa[k] * pb[k];
    acc += (double)part;
  }
  const long long tid = (long long)blockIdx.x * WS_THREADS + threadIdx.x, nthreads = (long long)gridDim.x * WS_THREADS;
  for (long long j = nv * V + (nthreads - 1 - tid); j < n; j += nthreads) acc += (double)(x[j] * w[j]);
  return acc;
}

template <typename T>
__global__ __launch_bounds__(WS_THREADS) void weighted_sum2_partial_kernel(long long n1, const T* __restrict__ x1,
                                                                           const T* __restrict__ w1, long long n2,
                                                                           const T* __restrict__ x2, const T* __restrict__ w2,
                                                                           double* __restrict__ partial) {
  __shared__ double s_p[WS_THREADS / 64];
  double acc = ws_dot_range<T>(x1, w1, n1) + ws_dot_range<T>(x2, w2, n2);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0;
    for (int w = 0; w < WS_THREADS / 64; ++w) a += s_p[w];
    partial[blockIdx.x] = a;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void weighted_sum2_finish_kernel(int groups, const double* __restrict__ partial, T* __restrict__ out) {
  __shared__ double s_p[4];
  double acc = 0;
  for (int i = threadIdx.x; i < groups; i += 256) acc += partial[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = (T)(s_p[0] + s_p[1] + s_p[2] + s_p[3]);
}

template <typename T>
__device__ __forceinline__ void ws_scale_range(T g, const T* __restrict__ w, T* __restrict__ out, long long n) {
  constexpr int V = 16 / (int)sizeof(T);
  constexpr int U = 4;
  const bool wide = ((((uintptr_t)w) | ((uintptr_t)out)) & 15) == 0;
  const long long nv = wide ? n / V : 0;
  const long long share = (nv + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * share, hi = lo + share < nv ? lo + share : nv;
  const uint4* wv = reinterpret_cast<const uint4*>(w);
  uint4* ov = reinterpret_cast<uint4*>(out);
  long long i = lo + threadIdx.x;
  for (; i + (U - 1) * WS_THREADS < hi; i += U * WS_THREADS) {
    uint4 b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = ws_load_stream(wv + i + u * WS_THREADS);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      T* pb = reinterpret_cast<T*>(&b[u]);
#pragma unroll
      for (int k = 0; k < V; ++k) pb[k] = g * pb[k];
      ov[i + u * WS_THREADS] = b[u];
    }
  }
  for (; i < hi; i += WS_THREADS) {
    uint4 b = wv[i];
    T* pb = reinterpret_cast<T*>(&b);
#pragma unroll
    for (int k = 0; k < V; ++k) pb[k] = g * pb[k];
    ov[i] = b;
  }
  const long long tid = (long long)blockIdx.x * WS_THREADS + threadIdx.x, nthreads = (long long)gridDim.x * WS_THREADS;
  for (long long j = nv * V + (nthreads - 1 - tid); j < n; j += nthreads) out[j] = g * w[j];
}

template <typename T>
__global__ __launch_bounds__(WS_THREADS) void weighted_sum2_backward_kernel(const T* __restrict__ grad_out, long long n1,
                                                                            const T* __restrict__ w1, T* __restrict__ g1,
                                                                            long long n2, const T* __restrict__ w2,
                                                                            T* __restrict__ g2) {
  const T g = grad_out[0];
  if (g1 != nullptr) ws_scale_range<T>(g, w1, g1, n1);
  if (g2 != nullptr) ws_scale_range<T>(g, w2, g2, n2);
}

template <typename T>
int mask_iou_forward(hipStream_t st, int B, long long P, const T* lhs, const T* rhs, double* partial, double* sums, T* loss) {
  if (B <= 0) return 0;
  kamd::ProfScope prof_(kamd::K_MASK_IOU, st);
  hipLaunchKernelGGL(mask_iou_partial_kernel<T>, dim3(MI_GROUPS, B), dim3(MI_THREADS), 0, st, P, lhs, rhs, partial);
  hipLaunchKernelGGL(mask_iou_finish_kernel<T>, dim3(1), dim3(64), 0, st, B, (const double*)partial, sums, loss);
  return (int)hipGetLastError();
}
template <typename T>
int mask_iou_backward(hipStream_t st, int B, long long P, const T* grad_loss, const T* other, const double* sums, T* grad) {
  if (B <= 0 || P <= 0) return 0;
  long long blocks = (P + 255) / 256;
  if (blocks > (long long)KAMD_NUM_CU * 8) blocks = (long long)KAMD_NUM_CU * 8;
  kamd::ProfScope prof_(kamd::K_MASK_IOU, st);
  hipLaunchKernelGGL(mask_iou_backward_kernel<T>, dim3((unsigned)blocks, B), dim3(256), 0, st, B, P, grad_loss, other, sums, grad);
  return (int)hipGetLastError();
}
template <typename T>
int texture_forward(hipStream_t st, int B, long long N, int C, int TH, int TW, int bilinear, const T* uv, const T* tex, T* out) {
  if (B <= 0 || N <= 0 || C <= 0) return 0;
  long long blocks = (N + 255) / 256;
  if (blocks > (long long)KAMD_NUM_CU * 16) blocks = (long long)KAMD_NUM_CU * 16;
  const dim3 grid((unsigned)blocks, B);
  kamd::ProfScope prof_(kamd::K_TEXTURE_MAPPING, st);
  if (bilinear)
    hipLaunchKernelGGL((texture_mapping_forward_kernel<T, true>), grid, dim3(256), 0, st, N, C, TH, TW, uv, tex, out);
  else
    hipLaunchKernelGGL((texture_mapping_forward_kernel<T, false>), grid, dim3(256), 0, st, N, C, TH, TW, uv, tex, out);
  return (int)hipGetLastError();
}
template <typename T>
int texture_backward(hipStream_t st, int B, long long N, int C, int TH, int TW, int bilinear, const T* uv, const T* tex,
                     const T* grad_out, T* g_tex, T* g_uv) {
  if (B <= 0 || N <= 0 || C <= 0) return 0;
  long long blocks = (N + 255) / 256;
  if (blocks > (long long)KAMD_NUM_CU * 16) blocks = (long long)KAMD_NUM_CU * 16;
  const dim3 grid((unsigned)blocks, B);
  kamd::ProfScope prof_(kamd::K_TEXTURE_MAPPING, st);
  if (bilinear)
    hipLaunchKernelGGL((texture_mapping_backward_kernel<T, true>), grid, dim3(256), 0, st, N, C, TH, TW, uv, tex, grad_out, g_tex, g_uv);
  else
    hipLaunchKernelGGL((texture_mapping_backward_kernel<T, false>), grid, dim3(256), 0, st, N, C, TH, TW, uv, tex, grad_out, g_tex, g_uv);
  return (int)hipGetLastError();
}

template <typename T>
int weighted_sum2_forward(hipStream_t st, long long n1, const T* x1, const T* w1, long long n2, const T* x2, const T* w2,
                          double* partial, T* out) {
  kamd::ProfScope prof_(kamd::K_WEIGHTED_SUM, st);
  hipLaunchKernelGGL(weighted_sum2_partial_kernel<T>, dim3(WS_GROUPS), dim3(WS_THREADS), 0, st, n1 > 0 ? n1 : 0, x1, w1,
                     n2 > 0 ? n2 : 0, x2, w2, partial);
  hipLaunchKernelGGL(weighted_sum2_finish_kernel<T>, dim3(1), dim3(256), 0, st, WS_GROUPS, (const double*)partial, out);
  return (int)hipGetLastError();
}
template <typename T>
int weighted_sum2_backward(hipStream_t st, const T* grad_out, long long n1, const T* w1, T* g1, long long n2, const T* w2, T* g2) {
  if ((g1 == nullptr || n1 <= 0) && (g2 == nullptr || n2 <= 0)) return 0;
  kamd::ProfScope prof_(kamd::K_WEIGHTED_SUM, st);
  hipLaunchKernelGGL(weighted_sum2_backward_kernel<T>, dim3(WS_GROUPS), dim3(WS_THREADS), 0, st, grad_out, n1 > 0 ? n1 : 0, w1, g1,
                     n2 > 0 ? n2 : 0, w2, g2);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

size_t kamd_weighted_sum2_workspace(void) { return (size_t)WS_GROUPS * sizeof(double); }

size_t kamd_mask_iou_workspace(int B) { return B > 0 ? (size_t)B * MI_GROUPS * 2 * sizeof(double) : 0; }

#define KAMD_RM_ENTRY(SFX, T)                                                                                          \
  int kamd_mask_iou_forward_##SFX(void* stream, int B, int64_t P, const T* lhs, const T* rhs, void* workspace,        \
                                  double* sums, T* loss) {                                                            \
    return mask_iou_forward<T>((hipStream_t)stream, B, (long long)P, lhs, rhs, (double*)workspace, sums, loss);       \
  }                                                                                                                    \
  int kamd_mask_iou_backward_##SFX(void* stream, int B, int64_t P, const T* grad_loss, const T* other,                \
                                   const double* sums, T* grad) {                                                      \
    return mask_iou_backward<T>((hipStream_t)stream, B, (long long)P, grad_loss, other, sums, grad);                  \
  }                                                                                                                    \
  int kamd_weighted_sum2_forward_##SFX(void* stream, int64_t n1, const T* x1, const T* w1, int64_t n2, const T* x2,   \
                                       const T* w2, void* workspace, T* out) {                                        \
    return weighted_sum2_forward<T>((hipStream_t)stream, (long long)n1, x1, w1, (long long)n2, x2, w2,                \
                                    (double*)workspace, out);                                                         \
  }                                                                                                                    \
  int kamd_weighted_sum2_backward_##SFX(void* stream, const T* grad_out, int64_t n1, const T* w1, T* g1, int64_t n2,  \
                                        const T* w2, T* g2) {                                                         \
    return weighted_sum2_backward<T>((hipStream_t)stream, grad_out, (long long)n1, w1, g1, (long long)n2, w2, g2);    \
  }                                                                                                                    \
  int kamd_texture_mapping_forward_##SFX(void* stream, int B, int64_t N, int C, int TH, int TW, int bilinear,         \
                                         const T* uv, const T* tex, T* out) {                                          \
    return texture_forward<T>((hipStream_t)stream, B, (long long)N, C, TH, TW, bilinear, uv, tex, out);               \
  }                                                                                                                    \
  int kamd_texture_mapping_backward_##SFX(void* stream, int B, int64_t N, int C, int TH, int TW, int bilinear,        \
                                          const T* uv, const T* tex, const T* grad_out, T* g_tex, T* g_uv) {          \
    return texture_backward<T>((hipStream_t)stream, B, (long long)N, C, TH, TW, bilinear, uv, tex, grad_out, g_tex,   \
                               g_uv);                                                                                  \
  }
KAMD_RM_ENTRY(f32, float)
KAMD_RM_ENTRY(f64, double)
#undef KAMD_RM_ENTRY

}  // extern "C"
