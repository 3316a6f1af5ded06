// DIB-R soft mask forward / backward for MI355X (gfx950).
//
// Replaces kaolin/csrc/render/mesh/dibr_soft_mask_cuda.cu:27-228 (K3) and :230-402 (K4) behind the C ABI
// of include/kaolin_amd.h.  Semantics kept from the reference (restated in oracle/dibr_oracle.inc):
//   K3  covered pixel -> 1; otherwise the first `knum` faces, in ascending index, whose enlarged bbox
//       holds the pixel centre contribute prob = exp(-sigmainv * d2 / mult / mult), d2 = first minimum of
//       {3 squared edge distances (4*mult^2 when the foot point is off the segment), 3 squared vertex
//       distances}; mask = 1 - prod(1 - prob).  K-buffers record prob / face / which-of-6 per hit.
//   K4  per uncovered pixel and stored hit: d(mask)/d(vertices) through the stored type, divided by mult.
// EPS is the double literal 1e-7 exactly as in the reference, so `down + EPS` is a double add and the
// following divide is a double divide rounded back to T (C's usual arithmetic conversions; the same
// expressions are spelled in the oracle).  Built with -ffp-contract=off.
//
// MI355X design: the face search uses the same tile bitmasks as the rasterizer (tile_bins.h) on the
// enlarged boxes; only wavefronts that own an uncovered pixel do any work.  The K-buffers (13*knum bytes per
// pixel, 390 B at knum = 30) dominate HBM traffic: they are initialised by one streaming fill kernel
// (16-byte stores), after which the tile kernel only touches the entries of actual hits.
#include "common.h"
#include "profile.h"
#include "tile_bins.h"
#include "../../include/kaolin_amd.h"

#define DIBR_EPS 1e-7

namespace {
using namespace kamd;

template <typename T> __device__ __forceinline__ T dibr_exp(T x);
template <> __device__ __forceinline__ float dibr_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double dibr_exp<double>(double x) { return exp(x); }

// Per-face, per-edge quantities that do not depend on the pixel (dibr_soft_mask_cuda.cu:112-125): the line
// coefficients, the products the foot-point numerators are built from, and the divisor (down + EPS) -- a double
// -- with its correctly rounded reciprocal.  Computed once per face and tile while the records are staged into
// LDS, with the very expressions the per-pixel code of the reference evaluates, so nothing changes numerically.
template <typename T>
struct EdgeInv {
  T A, B, C, AA, BB, AB, AC, BC;
};
template <typename T>
__device__ __forceinline__ void edge_invariants(T x1, T y1, T x2, T y2, EdgeInv<T>* e, double* den, double* rcp) {
  const T A = y2 - y1, Bc = x1 - x2, C = x2 * y1 - x1 * y2;
  const T down = A * A + Bc * Bc;
  e->A = A;
  e->B = Bc;
  e->C = C;
  e->AA = A * A;
  e->BB = Bc * Bc;
  e->AB = A * Bc;
  e->AC = A * C;
  e->BC = Bc * C;
  *den = (double)down + DIBR_EPS;
  *rcp = 1.0 / *den;
}
// num / den for a divisor whose correctly rounded reciprocal r is known: q = num*r, one exact-residual
// correction (Markstein): the correctly rounded double quotient in 3 operations instead of a full IEEE divide
__device__ __forceinline__ double div_by_invariant(double num, double den, double r) {
  const double q = num * r;
  const double rem = __builtin_fma(-q, den, num);
  return __builtin_fma(rem, r, q);
}

// squared distance of pixel (x0,y0) to the triangle's 3 edges / 3 vertices; returns the first minimum and
// its slot 0..5 (dibr_soft_mask_cuda.cu:98-159)
template <typename T>
__device__ __forceinline__ T closest_of_six(const T* v, const EdgeInv<T>* e, const double* den, const double* rcp,
                                            T x0, T y0, float multiplier, int* which) {
  T pdis[6];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const T x1 = v[i * 2], y1 = v[i * 2 + 1];
    const T x2 = v[((i + 1) % 3) * 2], y2 = v[((i + 1) % 3) * 2 + 1];
    const T up = e[i].A * x0 + e[i].B * y0 + e[i].C;
    const T x3n = e[i].BB * x0 - e[i].AB * y0 - e[i].AC;
    const T y3n = e[i].AA * y0 - e[i].AB * x0 - e[i].BC;
    const T x3 = (T)div_by_invariant((double)x3n, den[i], rcp[i]);
    const T y3 = (T)div_by_invariant((double)y3n, den[i], rcp[i]);
    const T direct = (x3 - x1) * (x3 - x2) + (y3 - y1) * (y3 - y2);
    if (direct > 0)
      pdis[i] = 4 * multiplier * multiplier;
    else
      pdis[i] = (T)div_by_invariant((double)(T)(up * up), den[i], rcp[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const T x1 = v[i * 2], y1 = v[i * 2 + 1];
    pdis[i + 3] = (x0 - x1) * (x0 - x1) + (y0 - y1) * (y0 - y1);
  }
  int w = 0;
  T d2 = pdis[0];
#pragma unroll
  for (int i = 1; i < 6; ++i)
    if (d2 > pdis[i]) {
      d2 = pdis[i];
      w = i;
    }
  *which = w;
  return d2;
}

template <typename T> struct SoftCap;  // faces staged in LDS per round
template <> struct SoftCap<float> { static constexpr int value = 256; };   // 184 B per face
template <> struct SoftCap<double> { static constexpr int value = 128; };  // 320 B per face

// ---- K-buffer fill: prob = 0, idx = -1, type = 0 (dibr_soft_mask.cpp:86-96) ----------------------------
__global__ __launch_bounds__(256) void fill_regions_kernel(uint4* __restrict__ a, size_t na, unsigned int va,
                                                           uint4* __restrict__ b, size_t nb, unsigned int vb,
                                                           uint4* __restrict__ c, size_t nc, unsigned int vc) {
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint4 A = make_uint4(va, va, va, va), Bv = make_uint4(vb, vb, vb, vb), C = make_uint4(vc, vc, vc, vc);
  for (size_t i = i0; i < na; i += stride) a[i] = A;
  for (size_t i = i0; i < nb; i += stride) b[i] = Bv;
  for (size_t i = i0; i < nc; i += stride) c[i] = C;
}

// fills [p, p+bytes) with the byte `v`: 16-byte body through the kernel above, unaligned head/tail by memset
struct FillPlan {
  uint4* body;
  size_t n16;
};
inline int fill_edges(hipStream_t st, void* p, size_t bytes, int v, FillPlan* plan) {
  char* c = (char*)p;
  size_t head = ((uintptr_t)c & 15) ? 16 - ((uintptr_t)c & 15) : 0;
  if (head > bytes) head = bytes;
  if (head) KAMD_CHECK(hipMemsetAsync(c, v, head, st));
  const size_t n16 = (bytes - head) / 16;
  const size_t tail = bytes - head - n16 * 16;
  if (tail) KAMD_CHECK(hipMemsetAsync(c + head + n16 * 16, v, tail, st));
  plan->body = (uint4*)(c + head);
  plan->n16 = n16;
  return 0;
}

// ---- K3 tile kernel --------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(TILE_THREADS) void soft_mask_tile_kernel(
    int B, int F, TileGeom g, int K, float sigmainv, float multiplier, const T* __restrict__ rec,
    const unsigned int* __restrict__ masks, const int64_t* __restrict__ sel_idx, T* __restrict__ soft_mask,
    T* __restrict__ prob_out, int64_t* __restrict__ idx_out, uint8_t* __restrict__ type_out,
    const unsigned int* __restrict__ tile_flags, uint8_t* __restrict__ hit_count) {
  constexpr int CAP = SoftCap<T>::value;
  __shared__ __attribute__((aligned(16))) T s_bbox[CAP * 4];
  __shared__ __attribute__((aligned(16))) T s_vert[CAP * 6];
  __shared__ __attribute__((aligned(16))) EdgeInv<T> s_edge[CAP * 3];
  __shared__ __attribute__((aligned(16))) double s_den[CAP * 3];
  __shared__ __attribute__((aligned(16))) double s_rcp[CAP * 3];
  __shared__ int s_ids[CAP];
  __shared__ int s_scan[TILE_THREADS / 64 + 1];
  __shared__ int s_any_uncovered;

  const int b = blockIdx.x % B;
  const int tile = blockIdx.x / B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t first_b = (int64_t)b * F;
  const int stride_b = (F + 31) / 32;
  const unsigned int* tmask = masks + mask_base(g.ntiles, first_b, b, tile, stride_b);

  const int tile_x = (tile % g.tiles_x) * TILE_W, tile_y = (tile / g.tiles_x) * TILE_H;
  const int sub_x = tile_x + (wave & 1) * SUB_W, sub_y = tile_y + (wave >> 1) * SUB_H;
  const int col = sub_x + (lane & 15), row = sub_y + (lane >> 4);
  const bool in_image = col < g.W && row < g.H;
  const size_t p1 = ((size_t)b * g.H + row) * g.W + col;
  const size_t pk = p1 * K;
  const bool uncovered = in_image && (int)sel_idx[in_image ? p1 : 0] < 0;

  if (tid == 0) s_any_uncovered = 0;
  __syncthreads();
  const bool wave_has_work = __any(uncovered);
  if (wave_has_work && lane == 0) s_any_uncovered = 1;
  __syncthreads();
  if (in_image && !uncovered) {
    soft_mask[p1] = (T)1.0;
    if (hit_count) hit_count[p1] = 0;
  }
  if (!s_any_uncovered) return;  // fully covered tile: nothing to search (uniform for the workgroup)
  const int nwords = (tile_flags != nullptr && tile_flags[(size_t)b * g.ntiles + tile]) ? stride_b : 0;

  const T x0 = pixel_x(multiplier, g.W, col);
  const T y0 = pixel_y(multiplier, g.H, row);
  // extent of this wavefront's UNCOVERED pixel centres (a face is kept if its box can hold one of them)
  T ux_min = uncovered ? x0 : (T)INFINITY, ux_max = uncovered ? x0 : (T)-INFINITY;
  T uy_min = uncovered ? y0 : (T)INFINITY, uy_max = uncovered ? y0 : (T)-INFINITY;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ux_min = fmin(ux_min, __shfl_xor(ux_min, d, 64));
    ux_max = fmax(ux_max, __shfl_xor(ux_max, d, 64));
    uy_min = fmin(uy_min, __shfl_xor(uy_min, d, 64));
    uy_max = fmax(uy_max, __shfl_xor(uy_max, d, 64));
  }

  int kid = 0;
  T all = 1.0;
  bool active = uncovered && K > 0;

  for (int seg0 = 0; seg0 < nwords; seg0 += TILE_THREADS) {
    const int wi = seg0 + tid;
    unsigned int word = wi < nwords ? tmask[wi] : 0u;
    int total;
    const int excl = block_exclusive_scan(__popc(word), s_scan, &total);
    for (int c0 = 0; c0 < total; c0 += CAP) {
      __syncthreads();
      {
        unsigned int wv = word;
        int pos = excl;
        while (wv) {
          const int bit = __ffs(wv) - 1;
          wv &= wv - 1;
          if (pos >= c0 && pos < c0 + CAP) s_ids[pos - c0] = wi * 32 + bit;
          ++pos;
        }
      }
      __syncthreads();
      const int n = min(CAP, total - c0);
      for (int i = tid; i < n * 10; i += TILE_THREADS) {
        const int k = i / 10, e = i % 10;
        const T v = rec[((size_t)first_b + s_ids[k]) * REC_STRIDE + e];
        if (e < 4)
          s_bbox[k * 4 + e] = v;
        else
          s_vert[k * 6 + (e - 4)] = v;
      }
      __syncthreads();
      for (int i = tid; i < n * 3; i += TILE_THREADS) {  // one thread per (face, edge)
        const int k = i / 3, ed = i % 3;
        const T* v = s_vert + k * 6;
        edge_invariants<T>(v[ed * 2], v[ed * 2 + 1], v[((ed + 1) % 3) * 2], v[((ed + 1) % 3) * 2 + 1], &s_edge[i],
                           &s_den[i], &s_rcp[i]);
      }
      __syncthreads();
      if (wave_has_work) {
        for (int k0 = 0; k0 < n; k0 += 64) {
          if (!__any(active)) break;
          const int k = k0 + lane;
          bool keep = false;
          if (k < n) {
            const T xmin = s_bbox[k * 4 + 0], ymin = s_bbox[k * 4 + 1], xmax = s_bbox[k * 4 + 2], ymax = s_bbox[k * 4 + 3];
            keep = !(ux_max < xmin || ux_min >= xmax || uy_max < ymin || uy_min >= ymax);
          }
          unsigned long long m = __ballot(keep);
          while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int kk = k0 + j;
            if (!active) continue;
            const T xmin = s_bbox[kk * 4 + 0], ymin = s_bbox[kk * 4 + 1], xmax = s_bbox[kk * 4 + 2], ymax = s_bbox[kk * 4 + 3];
            if (x0 < xmin || x0 >= xmax || y0 < ymin || y0 >= ymax) continue;
            int which;
            const T d2 = closest_of_six<T>(s_vert + kk * 6, s_edge + kk * 3, s_den + kk * 3, s_rcp + kk * 3, x0, y0, multiplier, &which);
            const T zz = sigmainv * d2 / multiplier / multiplier;
            const T pr = dibr_exp<T>(-zz);
            prob_out[pk + kid] = pr;
            idx_out[pk + kid] = s_ids[kk];
            type_out[pk + kid] = (uint8_t)(which + 1);
            all = (T)((double)all * (1.0 - (double)pr));
            ++kid;
            if (kid >= K) active = false;
          }
        }
      }
    }
  }
  if (uncovered) {
    soft_mask[p1] = (T)(1.0 - (double)all);
    if (hit_count) hit_count[p1] = (uint8_t)(kid > 255 ? 255 : kid);
  }
}

// ---- K4 ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void soft_mask_backward_kernel(
    long long total_pixels, int H, int W, int F, int K, const T* __restrict__ grad, const T* __restrict__ soft_mask,
    const int64_t* __restrict__ sel_idx, const T* __restrict__ prob_in, const int64_t* __restrict__ idx_in,
    const uint8_t* __restrict__ type_in, const T* __restrict__ img, float sigmainv, float multiplier,
    T* __restrict__ g_img, const uint8_t* __restrict__ hit_count) {
  const long long p1 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p1 >= total_pixels) return;
  // hit_count (optional, produced by our own forward): pixels without hits never touch the K-buffers; a stored
  // 255 means "255 or more", then the -1 terminator decides as in the reference
  int limit = K;
  if (hit_count) {
    const int hc = hit_count[p1];
    if (hc == 0) return;
    if (hc < 255 && hc < K) limit = hc;
  }
  if ((int)sel_idx[p1] >= 0) return;
  const size_t pk = (size_t)p1 * K;
  const int col = (int)(p1 % W);
  const int row = (int)((p1 / W) % H);
  const int b = (int)(p1 / ((long long)W * H));
  const T x0 = pixel_x(multiplier, W, col);
  const T y0 = pixel_y(multiplier, H, row);
  const T dLdp = grad[p1];
  const T all = soft_mask[p1];
  for (int kid = 0; kid < limit; ++kid) {
    const int f = (int)idx_in[pk + kid];
    if (f < 0) break;
    const size_t s6 = ((size_t)b * F + f) * 6;
    const T pr = prob_in[pk + kid];
    const T dLdz = (T)(-1.0 * sigmainv * dLdp * (1.0 - all) / (1.0 - pr + DIBR_EPS) * pr);
    const int e = (int)type_in[pk + kid] - 1;
    if (e >= 3) {
      const size_t ps = s6 + (size_t)(e - 3) * 2;
      const T x1 = img[ps], y1 = img[ps + 1];
      const T dLdx1 = dLdz * 2 * (x1 - x0);
      const T dLdy1 = dLdz * 2 * (y1 - y0);
      kamd_atomic_add(g_img + ps, (T)(dLdx1 / multiplier));
      kamd_atomic_add(g_img + ps + 1, (T)(dLdy1 / multiplier));
    } else {
      const size_t ps = s6 + (size_t)e * 2, ps2 = s6 + (size_t)((e + 1) % 3) * 2;
      const T x1 = img[ps], y1 = img[ps + 1], x2 = img[ps2], y2 = img[ps2 + 1];
      const T A = y2 - y1, Bc = x1 - x2, C = x2 * y1 - x1 * y2;
      const T up = A * x0 + Bc * y0 + C;
      const T down = A * A + Bc * Bc;
      const T d2 = up * up / (down + DIBR_EPS);
      const T dzdA = 2 * (x0 * up - d2 * A) / (down + DIBR_EPS);
      const T dzdB = 2 * (y0 * up - d2 * Bc) / (down + DIBR_EPS);
      const T dzdC = 2 * up / (down + DIBR_EPS);
      const T dLdx1 = dLdz * (dzdB - y2 * dzdC);
      const T dLdy1 = dLdz * (x2 * dzdC - dzdA);
      const T dLdx2 = dLdz * (y1 * dzdC - dzdB);
      const T dLdy2 = dLdz * (dzdA - x1 * dzdC);
      kamd_atomic_add(g_img + ps, (T)(dLdx1 / multiplier));
      kamd_atomic_add(g_img + ps + 1, (T)(dLdy1 / multiplier));
      kamd_atomic_add(g_img + ps2, (T)(dLdx2 / multiplier));
      kamd_atomic_add(g_img + ps2 + 1, (T)(dLdy2 / multiplier));
    }
  }
}

template <typename T>
int soft_mask_forward_launch(hipStream_t st, int B, int H, int W, int F, int K, const T* img, const T* large_bbox,
                             const int64_t* sel_idx, float sigmainv, float multiplier, T* soft_mask, T* prob,
                             int64_t* idx, uint8_t* type, void* workspace, uint8_t* hit_count) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const TileGeom g = tile_geom(H, W);
  const long long total_faces = (long long)B * F;
  if (total_faces > 0 && workspace == nullptr) return (int)hipErrorInvalidValue;
  // 1. K-buffer initialisation (one streaming pass)
  const size_t nk = (size_t)B * H * W * K;
  if (nk > 0) {
    FillPlan pa, pb, pc;
    KAMD_CHECK(fill_edges(st, prob, nk * sizeof(T), 0x00, &pa));
    KAMD_CHECK(fill_edges(st, idx, nk * 8, 0xFF, &pb));
    KAMD_CHECK(fill_edges(st, type, nk, 0x00, &pc));
    const size_t most = pb.n16 > pa.n16 ? pb.n16 : pa.n16;
    int blocks = (int)((most + 255) / 256 < (size_t)KAMD_NUM_CU * 16 ? (most + 255) / 256 : (size_t)KAMD_NUM_CU * 16);
    if (blocks < 1) blocks = 1;
    {
      kamd::ProfScope prof_(kamd::K_SOFT_FILL, st);
      hipLaunchKernelGGL(fill_regions_kernel, dim3(blocks), dim3(256), 0, st, pa.body, pa.n16, 0u, pb.body, pb.n16,
                       0xFFFFFFFFu, pc.body, pc.n16, 0u);
    }
    KAMD_CHECK(hipGetLastError());
  }
  // 2. bin the enlarged boxes, 3. search
  T* rec = (T*)workspace;
  unsigned int* masks = (unsigned int*)((char*)workspace + align256((size_t)total_faces * REC_STRIDE * sizeof(T)));
  unsigned int* flags = total_faces > 0 ? masks + mask_words(g.ntiles, B, total_faces) : nullptr;
  if (total_faces > 0) {
    KAMD_CHECK(hipMemsetAsync(masks, 0, (mask_words(g.ntiles, B, total_faces) + flag_words(g.ntiles, B)) * 4, st));
    {
      kamd::ProfScope prof_(kamd::K_BIN_FACES, st);
      hipLaunchKernelGGL(bin_faces_kernel<T>, dim3(kamd_cdiv(total_faces, 256)), dim3(256), 0, st, B, F, total_faces,
                       (const int64_t*)nullptr, large_bbox, img, (const T*)nullptr, g, multiplier, rec, masks, flags);
    }
    KAMD_CHECK(hipGetLastError());
  }
  {
    kamd::ProfScope prof_(kamd::K_SOFT_TILE, st);
    hipLaunchKernelGGL(soft_mask_tile_kernel<T>, dim3(g.ntiles * B), dim3(TILE_THREADS), 0, st, B, F, g, K, sigmainv,
                     multiplier, rec, masks, sel_idx, soft_mask, prob, idx, type, flags, hit_count);
  }
  KAMD_RETURN_LAST_ERROR();
}

template <typename T>
int soft_mask_backward_launch(hipStream_t st, int B, int H, int W, int F, int K, const T* grad, const T* soft_mask,
                              const int64_t* sel_idx, const T* prob, const int64_t* idx, const uint8_t* type,
                              const T* img, float sigmainv, float multiplier, T* g_img, const uint8_t* hit_count) {
  const long long total = (long long)B * H * W;
  if (total <= 0 || F <= 0 || K <= 0) return 0;
  {
    kamd::ProfScope prof_(kamd::K_SOFT_BACKWARD, st);
    hipLaunchKernelGGL(soft_mask_backward_kernel<T>, dim3(kamd_cdiv(total, 256)), dim3(256), 0, st, total, H, W, F, K,
                     grad, soft_mask, sel_idx, prob, idx, type, img, sigmainv, multiplier, g_img, hit_count);
  }
  KAMD_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {

size_t kamd_dibr_soft_mask_forward_workspace(int B, int H, int W, int F, int elem_size) {
  if (B <= 0 || H <= 0 || W <= 0 || F <= 0) return 0;
  return kamd::bins_workspace_bytes(B, H, W, (long long)B * F, elem_size);
}

int kamd_dibr_soft_mask_forward_f32(void* stream, int B, int H, int W, int F, int K, const float* img,
                                    const float* large_bbox, const int64_t* sel_idx, float sigmainv, float multiplier,
                                    float* soft_mask, float* prob, int64_t* idx, uint8_t* type, void* workspace,
                                    uint8_t* hit_count) {
  return soft_mask_forward_launch<float>((hipStream_t)stream, B, H, W, F, K, img, large_bbox, sel_idx, sigmainv,
                                         multiplier, soft_mask, prob, idx, type, workspace, hit_count);
}
int kamd_dibr_soft_mask_forward_f64(void* stream, int B, int H, int W, int F, int K, const double* img,
                                    const double* large_bbox, const int64_t* sel_idx, float sigmainv, float multiplier,
                                    double* soft_mask, double* prob, int64_t* idx, uint8_t* type, void* workspace,
                                    uint8_t* hit_count) {
  return soft_mask_forward_launch<double>((hipStream_t)stream, B, H, W, F, K, img, large_bbox, sel_idx, sigmainv,
                                          multiplier, soft_mask, prob, idx, type, workspace, hit_count);
}
int kamd_dibr_soft_mask_backward_f32(void* stream, int B, int H, int W, int F, int K, const float* grad,
                                     const float* soft_mask, const int64_t* sel_idx, const float* prob,
                                     const int64_t* idx, const uint8_t* type, const float* img, float sigmainv,
                                     float multiplier, float* g_img, const uint8_t* hit_count) {
  return soft_mask_backward_launch<float>((hipStream_t)stream, B, H, W, F, K, grad, soft_mask, sel_idx, prob, idx, type,
                                          img, sigmainv, multiplier, g_img, hit_count);
}
int kamd_dibr_soft_mask_backward_f64(void* stream, int B, int H, int W, int F, int K, const double* grad,
                                     const double* soft_mask, const int64_t* sel_idx, const double* prob,
                                     const int64_t* idx, const uint8_t* type, const double* img, float sigmainv,
                                     float multiplier, double* g_img, const uint8_t* hit_count) {
  return soft_mask_backward_launch<double>((hipStream_t)stream, B, H, W, F, K, grad, soft_mask, sel_idx, prob, idx,
                                           type, img, sigmainv, multiplier, g_img, hit_count);
}

}  // extern "C"
