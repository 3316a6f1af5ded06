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
// MI355X design: tile face lists (tile_lists.h) built from the enlarged boxes; the search works on 16 x 4-pixel
// sub-tiles that hold an uncovered pixel within reach of a box (soft2.inc).  The K-buffers of the reference contract
// (13*knum bytes per pixel, 390 B at knum = 30) are initialised by one streaming fill kernel (16-byte stores), after which
// only the entries of actual hits are touched; the autograd path keeps a compact hit list instead.
#include "common.h"
#include <mutex>
#include <stdio.h>
#include "profile.h"
#include "tile_bins.h"
#include "tile_lists.h"
#include "dibr_internal.h"
#include "phase_prof.h"
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

// fills [p, p+bytes) with the byte `v`: 16-byte body through the kernel above, unaligned head/tail by the one-wave byte kernel
struct FillPlan {
  uint4* body;
  size_t n16;
};
inline int fill_edges(hipStream_t st, void* p, size_t bytes, int v, FillPlan* plan) {
  char* c = (char*)p;
  size_t head = ((uintptr_t)c & 15) ? 16 - ((uintptr_t)c & 15) : 0;
  if (head > bytes) head = bytes;
  const size_t n16 = (bytes - head) / 16;
  const size_t tail = bytes - head - n16 * 16;
  if (head || tail) {
    hipLaunchKernelGGL(kamd_fill_edges_kernel, dim3(1), dim3(64), 0, st, (unsigned char*)c, (int)head,
                       (unsigned char*)c + head + n16 * 16, (int)tail, (unsigned char)(v & 0xFF));
    KAMD_CHECK(hipGetLastError());
  }
  plan->body = (uint4*)(c + head);
  plan->n16 = n16;
  return 0;
}

#include "soft2.inc"

// ---- K4 ---------------------------------------------------------------------------------------------------
// One 64-lane workgroup per 16x4-pixel sub-tile.  The reference adds every (pixel, hit) contribution to the
// face's vertices with global atomics (dibr_soft_mask_cuda.cu:299-302,339-347); a silhouette face receives
// hundreds of them.  Here the wavefront first sums per face in an LDS hash table (ds_add_f32), then flushes one
// global atomic per touched (face, coordinate).
constexpr int SB_HT = 512;  // hash slots (a sub-tile rarely sees more than ~300 distinct faces; overflow -> global atomics)

// open addressing on the face id: the slot of face f (claimed on first use), or -1 when 16 probes find neither f nor a
// free slot (a crowded table degrades to global atomics, not to a scan).  One probe sequence per hit, not per value.
__device__ __forceinline__ int sb_find(int* s_key, int f) {
  int slot = (int)(((unsigned)f * 2654435761u) >> 22) & (SB_HT - 1);
  for (int probe = 0; probe < 16; ++probe) {
    const int k = atomicCAS(&s_key[slot], -1, f);
    if (k == -1 || k == f) return slot;
    slot = (slot + 1) & (SB_HT - 1);
  }
  return -1;
}
template <typename T>
__device__ __forceinline__ void sb_add(T* s_acc, T* __restrict__ g_face, int slot, int off, T v) {
  if (slot >= 0)
    atomicAdd(&s_acc[slot * 6 + off], v);
  else
    kamd_atomic_add(g_face + off, v);
}

template <typename T>
__global__ __launch_bounds__(64) void soft_mask_backward_kernel(
    int B, int H, int W, int F, int K, const T* __restrict__ grad, const T* __restrict__ soft_mask,
    const int64_t* __restrict__ sel_idx, const T* __restrict__ prob_in, const int64_t* __restrict__ idx_in,
    const uint8_t* __restrict__ type_in, const T* __restrict__ img, float sigmainv, float multiplier,
    T* __restrict__ g_img, const uint8_t* __restrict__ hit_count) {
  __shared__ int s_key[SB_HT];
  __shared__ T s_acc[SB_HT * 6];
  const int subs_x = (W + SUB_W - 1) / SUB_W, subs_y = (H + SUB_H - 1) / SUB_H;
  const int sub = blockIdx.x % (subs_x * subs_y), b = blockIdx.x / (subs_x * subs_y);
  const int lane = threadIdx.x;
  const int col = (sub % subs_x) * SUB_W + (lane & 15), row = (sub / subs_x) * SUB_H + (lane >> 4);
  const bool in_image = col < W && row < H;
  const size_t p1 = ((size_t)b * H + row) * W + col;
  const size_t pk = p1 * K;
  // hit_count (optional, produced by our own forward): pixels without hits never touch the K-buffers; a stored
  // 255 means "255 or more", then the -1 terminator decides as in the reference
  int limit = 0;
  if (in_image) {
    if (hit_count) {
      const int hc = hit_count[p1];
      limit = hc == 255 ? K : min(hc, K);
    } else if ((int)sel_idx[p1] < 0) {
      limit = K;
    }
  }
  if (limit > 0 && (int)sel_idx[p1] >= 0) limit = 0;
  if (!__any(limit > 0)) return;

  for (int i = lane; i < SB_HT; i += 64) s_key[i] = -1;
  for (int i = lane; i < SB_HT * 6; i += 64) s_acc[i] = 0;
  __syncthreads();

  const T x0 = pixel_x(multiplier, W, col);
  const T y0 = pixel_y(multiplier, H, row);
  const T dLdp = limit > 0 ? grad[p1] : (T)0;
  const T all = limit > 0 ? soft_mask[p1] : (T)0;
  int max_limit = limit;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) max_limit = max(max_limit, __shfl_xor(max_limit, d, 64));
  for (int kid = 0; kid < max_limit; ++kid) {
    if (kid >= limit) continue;
    const int f = (int)idx_in[pk + kid];
    if (f < 0) {
      limit = 0;
      continue;
    }
    const size_t s6 = ((size_t)b * F + f) * 6;
    const int key = f;  // the table is per image: b is fixed for the workgroup
    const int slot = sb_find(s_key, key);
    const T pr = prob_in[pk + kid];
    const T dLdz = (T)(-1.0 * sigmainv * dLdp * (1.0 - all) / (1.0 - pr + DIBR_EPS) * pr);
    const int e = (int)type_in[pk + kid] - 1;
    if (e >= 3) {
      const int o = (e - 3) * 2;
      const T x1 = img[s6 + o], y1 = img[s6 + o + 1];
      const T dLdx1 = dLdz * 2 * (x1 - x0);
      const T dLdy1 = dLdz * 2 * (y1 - y0);
      sb_add<T>(s_acc, g_img + s6, slot, o, (T)(dLdx1 / multiplier));
      sb_add<T>(s_acc, g_img + s6, slot, o + 1, (T)(dLdy1 / multiplier));
    } else {
      const int o = e * 2, o2 = ((e + 1) % 3) * 2;
      const T x1 = img[s6 + o], y1 = img[s6 + o + 1], x2 = img[s6 + o2], y2 = img[s6 + o2 + 1];
      const T A = y2 - y1, Bc = x1 - x2, C = x2 * y1 - x1 * y2;
      const T up = A * x0 + Bc * y0 + C;
      const T down = A * A + Bc * Bc;
      // the reference divides four times by the double (down + EPS) (dibr_soft_mask_cuda.cu:318-327); gradients are
      // compared at 1e-5, not bit for bit: one double reciprocal, four double products (each rounded to T as before)
      const double rden = 1.0 / ((double)down + DIBR_EPS);
      const T d2 = (T)((double)(T)(up * up) * rden);
      const T dzdA = (T)((double)(T)(2 * (x0 * up - d2 * A)) * rden);
      const T dzdB = (T)((double)(T)(2 * (y0 * up - d2 * Bc)) * rden);
      const T dzdC = (T)((double)(T)(2 * up) * rden);
      const T dLdx1 = dLdz * (dzdB - y2 * dzdC);
      const T dLdy1 = dLdz * (x2 * dzdC - dzdA);
      const T dLdx2 = dLdz * (y1 * dzdC - dzdB);
      const T dLdy2 = dLdz * (dzdA - x1 * dzdC);
      sb_add<T>(s_acc, g_img + s6, slot, o, (T)(dLdx1 / multiplier));
      sb_add<T>(s_acc, g_img + s6, slot, o + 1, (T)(dLdy1 / multiplier));
      sb_add<T>(s_acc, g_img + s6, slot, o2, (T)(dLdx2 / multiplier));
      sb_add<T>(s_acc, g_img + s6, slot, o2 + 1, (T)(dLdy2 / multiplier));
    }
  }
  __syncthreads();
  for (int i = lane; i < SB_HT * 6; i += 64) {
    const int k = s_key[i / 6];
    const T v = s_acc[i];
    if (k >= 0 && v != (T)0) kamd_atomic_add(g_img + ((size_t)b * F + k) * 6 + (i % 6), v);
  }
}

// ---- launches ---------------------------------------------------------------------------------------------------------
inline size_t flat_shard_cap_of(int B, int H, int W, int K) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return flat_shard_cap((size_t)B * tl::pass_geom(H, W, tl::S_TILE).ntiles * tl::S_SUBS, K);
}
// what a flat hit record can address: (b * F + face) in 29 bits, row and column in 16 bits each
inline bool flat_list_fits(int B, int H, int W, int F) {
  return (long long)B * (F > 0 ? F : 0) < (1ll << FLAT_KEY_BITS) && H <= 65535 && W <= 65535;
}
// The standalone operators: bin the enlarged boxes (count, scan, emit), classify the pixels from the given
// selected_face_idx, search.  `work` (kamd_dibr_soft_mask_work_words 32-bit words) receives the worklist; the autograd
// path keeps it for the backward pass.
template <typename T>
int soft2_search_launch(hipStream_t st, int B, int H, int W, int F, int K, float sigmainv, float multiplier, const T* rec,
                        const tl::Lists& LS, const unsigned int* work, T* soft_mask, T* prob, int64_t* idx, uint8_t* type,
                        uint8_t* hit_count, const HitList2<T>* lean, unsigned short* pixcnt, T* prob_pm,
                        void* zero_p = nullptr, size_t zero_bytes = 0,  // (a 16-byte aligned range the eval launch clears)
                        const unsigned int* span_src = nullptr, unsigned int* span_dst = nullptr, int span_n = 0,
                        unsigned int* magic_dst = nullptr, unsigned int* over_count = nullptr, unsigned int* over_list = nullptr,
                        unsigned int* ord_count = nullptr, uint4* ord_list = nullptr) {
  if (over_count == nullptr || over_list == nullptr || ord_count == nullptr || ord_list == nullptr)
    return (int)hipErrorInvalidValue;  // (the workspace's hand-over and dealing lists)
  const unsigned int shard_cap = tl::work_shard_cap(B, H, W);
  const long long n_sub = (long long)B * LS.ntiles * tl::S_SUBS;
  Select2Args<T> sa{};
  sa.B = B;
  sa.F = F;
  sa.H = H;
  sa.W = W;
  sa.K = K;
  sa.multiplier = multiplier;
  sa.rec = rec;
  sa.L = LS;
  sa.work = work;
  sa.shard_cap = shard_cap;
  sa.pixcnt = pixcnt;
  if (lean) sa.list = *lean;
  sa.idx_out = idx;
  sa.hit_count = hit_count;
  sa.over_count = over_count;
  sa.over_list = over_list;
  sa.ord_count = ord_count;
  sa.ord_list = ord_list;
  sa.ord_cap = (unsigned int)tl::WORK_SHARDS * shard_cap;
  sa.bitwords = s2_bitwords_of(F);
  const size_t sel_shmem = (size_t)2 * sa.bitwords * 4;
  Eval2Args<T> ea{};
  ea.B = B;
  ea.F = F;
  ea.H = H;
  ea.W = W;
  ea.K = K;
  ea.sigmainv = sigmainv;
  ea.multiplier = multiplier;
  ea.inv_mult2 = 1.0 / ((double)multiplier * (double)multiplier);
  ea.rec = rec;
  ea.tiles_x_s = LS.tiles_x;
  ea.work = work;
  ea.shard_cap = shard_cap;
  ea.pixcnt = pixcnt;
  ea.soft_mask = soft_mask;
  if (lean) ea.list = *lean;
  ea.prob_out = prob;
  ea.idx_out = idx;
  ea.type_out = type;
  ea.prob_pm = prob_pm;
  ea.zero_p = (uint4*)zero_p;
  ea.zero_n16 = zero_bytes / 16;
  ea.span_src = span_src;
  ea.span_dst = span_dst;
  ea.span_n = span_n;
  ea.magic_dst = magic_dst;
  ea.magic = tl::work_magic(B, H, W);
  ea.ord_count = sa.ord_count;
  ea.ord_list = sa.ord_list;
  ea.ord_cap = sa.ord_cap;
  // one wavefront (select) / workgroup (eval) per work item; the number of items is known on the device only, so the
  // grids cover the worklist round-robin
  // (select: 128 workgroups per CU -- with 32, a scene of many light items (the knot: 37 600) ran five places per workgroup one
  // after the other in a fixed deal: 237 us; 64: 213; 128: 203, most workgroups take one place and the dispatcher balances them;
  // C4's 5 300 items do not care: profiles/r04j_select_grid_ab.txt)
  static const int sel_per_cu = kamd_env_int("KAMD_SOFT_SELECT_PER_CU", 128);
  static const int eval_per_cu = kamd_env_int("KAMD_SOFT_EVAL_PER_CU", 32);
  const long long sel_want = (long long)KAMD_NUM_CU * sel_per_cu, eval_want = (long long)KAMD_NUM_CU * eval_per_cu;
  const dim3 sel_grid((unsigned)(n_sub < sel_want ? (n_sub > 0 ? n_sub : 1) : sel_want));
  const dim3 eval_grid((unsigned)(n_sub < eval_want ? (n_sub > 0 ? n_sub : 1) : eval_want));
  if (lean)
    KAMD_LAUNCH_TIMED(kamd::K_SOFT_SELECT, (soft_select_kernel<T, true>), sel_grid, dim3(64), sel_shmem, st, sa);
  else
    KAMD_LAUNCH_TIMED(kamd::K_SOFT_SELECT, (soft_select_kernel<T, false>), sel_grid, dim3(64), sel_shmem, st, sa);
  KAMD_CHECK(hipGetLastError());
  {  // the items of tiles with more entries than ordered slots (usually none), and the deal of all items for the eval launch
    const dim3 rounds_grid((unsigned)(n_sub < S2_ROUNDS_GRID ? (n_sub > 0 ? n_sub : 1) : S2_ROUNDS_GRID));
    if (lean)
      KAMD_LAUNCH_TIMED(kamd::K_SOFT_SELECT_ROUNDS, (soft_select_rounds_kernel<T, true>), rounds_grid, dim3(64), sel_shmem, st, sa);
    else
      KAMD_LAUNCH_TIMED(kamd::K_SOFT_SELECT_ROUNDS, (soft_select_rounds_kernel<T, false>), rounds_grid, dim3(64), sel_shmem, st, sa);
    KAMD_CHECK(hipGetLastError());
  }
  const bool lds_fold = K <= S2_LDS_KMAX;
  const size_t shmem = lds_fold ? (size_t)64 * (K > 0 ? K : 1) * sizeof(T) : 0;
  {
    if (lean) {
      if (lds_fold)
        KAMD_LAUNCH_TIMED(kamd::K_SOFT_TILE, (soft_eval_kernel<T, true, true>), eval_grid, dim3(S2_EVAL_THREADS), shmem, st, ea);
      else
        KAMD_LAUNCH_TIMED(kamd::K_SOFT_TILE, (soft_eval_kernel<T, true, false>), eval_grid, dim3(S2_EVAL_THREADS), 0, st, ea);
    } else {
      if (lds_fold)
        KAMD_LAUNCH_TIMED(kamd::K_SOFT_TILE, (soft_eval_kernel<T, false, true>), eval_grid, dim3(S2_EVAL_THREADS), shmem, st, ea);
      else
        KAMD_LAUNCH_TIMED(kamd::K_SOFT_TILE, (soft_eval_kernel<T, false, false>), eval_grid, dim3(S2_EVAL_THREADS), 0, st, ea);
    }
  }
  KAMD_CHECK(hipGetLastError());
  if (!lds_fold) {
    if (lean)
      hipLaunchKernelGGL((soft_fold_kernel<T, true>), sel_grid, dim3(64), 0, st, ea);
    else
      hipLaunchKernelGGL((soft_fold_kernel<T, false>), sel_grid, dim3(64), 0, st, ea);
  }
  return (int)hipGetLastError();
}

template <typename T>
int soft_mask_forward_launch(hipStream_t st, int B, int H, int W, int F, int K, const T* img, const T* large_bbox,
                             const int64_t* sel_idx, float sigmainv, float multiplier, T* soft_mask, T* prob,
                             int64_t* idx, uint8_t* type, void* workspace, uint8_t* hit_count, const HitList2<T>* lean,
                             unsigned int* work, bool raw = false, double raw_multiplier = 1.0, double raw_margin = 0.0) {
  // raw: `img` is the UNSCALED (B,F,3,2) operator input and `large_bbox` is unused; scaling by raw_multiplier and the
  // boxes enlarged by raw_margin (= boxlen * multiplier) are produced inside the bin kernel
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const long long total_faces = (long long)B * F;
  if (workspace == nullptr || work == nullptr) return (int)hipErrorInvalidValue;
  if (lean != nullptr && ((long long)B * H * W >= (1ll << 31) || !flat_list_fits(B, H, W, F))) return (int)hipErrorInvalidValue;
  const tl::Layout lay = tl::make_layout(B, H, W, total_faces, (int)sizeof(T), false, true, K);
  tl::Lists LS = tl::lists_of(workspace, lay.s, B, true);
  tl::Lists none{};
  T* rec = (T*)((char*)workspace + lay.s.rec);
  // K-buffer initialisation (one streaming pass) -- reference-contract outputs only
  const size_t nk = lean ? 0 : (size_t)B * H * W * K;
  if (nk > 0) {
    FillPlan pa, pb, pc;
    KAMD_CHECK(fill_edges(st, prob, nk * sizeof(T), 0x00, &pa));
    KAMD_CHECK(fill_edges(st, idx, nk * 8, 0xFF, &pb));
    KAMD_CHECK(fill_edges(st, type, nk, 0x00, &pc));
    const size_t most = pb.n16 > pa.n16 ? pb.n16 : pa.n16;
    int blocks = (int)((most + 255) / 256 < (size_t)KAMD_NUM_CU * 16 ? (most + 255) / 256 : (size_t)KAMD_NUM_CU * 16);
    if (blocks < 1) blocks = 1;
    kamd::ProfScope prof_(kamd::K_SOFT_FILL, st);
    hipLaunchKernelGGL(fill_regions_kernel, dim3(blocks), dim3(256), 0, st, pa.body, pa.n16, 0u, pb.body, pb.n16,
                       0xFFFFFFFFu, pc.body, pc.n16, 0u);
  }
  KAMD_CHECK(hipGetLastError());
  KAMD_CHECK(kamd_zero_async(workspace, lay.zero_bytes, st));
  KAMD_CHECK(kamd_zero_async(work, tl::WORK_HEADER * 4, st));
  if (total_faces > 0) {
    tl::BinIn<T> in{};
    in.B = B;
    in.F = F;
    in.total_faces = total_faces;
    in.img = img;
    in.lay = FaceLayout{3, 1, 1};
    in.bbox_s = raw ? nullptr : large_bbox;
    in.mult = raw ? (T)raw_multiplier : (T)1;
    in.margin = raw ? (T)raw_margin : (T)0;
    in.multiplier = multiplier;
    in.H = H;
    in.W = W;
    in.rec_s = rec;
    KAMD_LAUNCH_TIMED(kamd::K_BIN_FACES, (tl::bin_faces_kernel2<T, false, true>), dim3(kamd_cdiv(total_faces, 256)), dim3(256), 0, st, in, none, LS);
  }
  KAMD_CHECK(hipGetLastError());
  {
    const tl::PassGeom gr = tl::pass_geom(H, W, tl::R_TILE);
    kamd::ProfScope prof_(kamd::K_SOFT_CLASSIFY, st);
    hipLaunchKernelGGL(soft_items_kernel<T>, dim3(gr.ntiles * B), dim3(256), 0, st, B, H, W, total_faces > 0 ? 1 : 0, sel_idx,
                       soft_mask, lean ? (uint8_t*)nullptr : hit_count, LS, gr.tiles_x, work, tl::work_shard_cap(B, H, W));
  }
  KAMD_CHECK(hipGetLastError());
  if (total_faces > 0)
    KAMD_CHECK(soft2_search_launch<T>(st, B, H, W, F, K, sigmainv, multiplier, rec, LS, work, soft_mask, prob, idx, type,
                                      hit_count, lean, (unsigned short*)((char*)workspace + lay.s.pixcnt),
                                      (T*)((char*)workspace + lay.s.prob_pm), nullptr, 0, nullptr, nullptr, 0, nullptr,
                                      (unsigned int*)((char*)workspace + lay.s.over_count), (unsigned int*)((char*)workspace + lay.s.over_list),
                                      (unsigned int*)((char*)workspace + lay.s.ord_count), (uint4*)((char*)workspace + lay.s.ord_list)));
  KAMD_RETURN_LAST_ERROR();
}

template <typename T>
int soft_mask_backward_list_launch(hipStream_t st, int B, int H, int W, int F, int K, const T* grad, const T* soft_mask,
                                   const HitList2<T>& list, const unsigned int* work, const T* img, double img_scale,
                                   float sigmainv, float multiplier, T* g_img, unsigned int* bigwork = nullptr) {
  if ((long long)B * H * W <= 0 || F <= 0) return 0;
  {
    // persistent workgroups over the rounds of 256 hits (their number is known on the device only)
    static const int per_cu = kamd_env_int("KAMD_SOFT_BWD_PER_CU", 16);
    KAMD_LAUNCH_TIMED(kamd::K_SOFT_BACKWARD_LIST, soft_mask_backward_flat_kernel<T>, dim3(KAMD_NUM_CU * per_cu), dim3(256), 0, st, H, W, F,
                      flat_view_magic(F), grad, soft_mask, list, img, (T)img_scale, sigmainv, multiplier, 1.0 / (double)multiplier, g_img, bigwork);
  }
  KAMD_RETURN_LAST_ERROR();
}

template <typename T>
int soft_mask_backward_launch(hipStream_t st, int B, int H, int W, int F, int K, const T* grad, const T* soft_mask,
                              const int64_t* sel_idx, const T* prob, const int64_t* idx, const uint8_t* type,
                              const T* img, float sigmainv, float multiplier, T* g_img, const uint8_t* hit_count) {
  const long long total = (long long)B * H * W;
  if (total <= 0 || F <= 0 || K <= 0) return 0;
  const long long blocks = (long long)B * ((W + SUB_W - 1) / SUB_W) * ((H + SUB_H - 1) / SUB_H);
  {
    kamd::ProfScope prof_(kamd::K_SOFT_BACKWARD, st);
    hipLaunchKernelGGL(soft_mask_backward_kernel<T>, dim3((unsigned)blocks), dim3(64), 0, st, B, H, W, F, K, grad,
                       soft_mask, sel_idx, prob, idx, type, img, sigmainv, multiplier, g_img, hit_count);
  }
  KAMD_RETURN_LAST_ERROR();
}

// ---- fused DIB-R front door: rasterize (front faces) + soft mask (all faces) in one call ------------------------------------
// One binning launch per phase serves both passes (the vertices are read once); the rasterizer's tile kernel classifies
// the pixels for the soft mask and queues the search's work items, so the forward is: clear counters, count, scan, emit,
// raster tiles, search -- six launches on the caller's stream, no side stream.
template <typename T>
int dibr_forward_fused(hipStream_t st, int B, int H, int W, int F, int D, int K, const T* z, int64_t z_face_stride,
                       int64_t z_vertex_stride, const T* img, const T* feat, const uint8_t* valid, const T* front,
                       int64_t front_stride, double multiplier, float eps, float sigmainv, double margin, T* interp,
                       int64_t* face_idx, T* weights, T* soft_mask, const HitList2<T>& list, unsigned int* work,
                       void* workspace, T* g_img_zero) {
  if (B <= 0 || H <= 0 || W <= 0) {
    // an empty image: no kernel runs, but the caller's gradient buffer (allocated uninitialised) is returned by the backward
    if (g_img_zero != nullptr && B > 0 && F > 0) KAMD_CHECK(kamd_zero_async(g_img_zero, (size_t)B * F * 6 * sizeof(T), st));
    return 0;
  }
  const long long total_faces = (long long)B * F;
  if (workspace == nullptr || work == nullptr) return (int)hipErrorInvalidValue;
  if ((long long)B * H * W >= (1ll << 31) || !flat_list_fits(B, H, W, F)) return (int)hipErrorInvalidValue;
  const tl::Layout lay = tl::make_layout(B, H, W, total_faces, (int)sizeof(T), true, true, K);
  tl::Lists LR = tl::lists_of(workspace, lay.r, B, false);
  tl::Lists LS = tl::lists_of(workspace, lay.s, B, true);
  if (sizeof(T) == 4) LS.big_hash = work + tl::WORK_BIGHASH_WORD;  // (the soft pass' big faces are entered for the backward pass: tile_lists.h)
  // (KAMD_ROW_ORDER=2: the tile kernels start from the middle of the image instead of the middle of the covered rows: A/B runs)
  const bool row_span = kamd_env_int("KAMD_ROW_ORDER", 1) == 1;
  if (!row_span) LR.row_span = nullptr;
  T* rec_r = (T*)((char*)workspace + lay.r.rec);
  T* rec_s = (T*)((char*)workspace + lay.s.rec);
  // one launch clears the list heads, the work-list header and (when the caller will differentiate) the buffer the backward
  // kernels accumulate into
  // (the gradient buffer is not needed before the backward pass: when the eval kernel will run, IT clears the buffer --
  // 9.6 MB at C4 that would otherwise hold up the binning kernel by ~4 us)
  const size_t g_bytes = (size_t)total_faces * 6 * sizeof(T);
  const bool zero_in_eval = g_img_zero != nullptr && total_faces > 0 && ((uintptr_t)g_img_zero & 15) == 0 && g_bytes % 16 == 0 &&
                            kamd_env_int("KAMD_DIBR_ZERO_IN_EVAL", 1) == 1;
  KAMD_CHECK(kamd_zero3_async(workspace, lay.zero_bytes, work, tl::WORK_HEADER * 4, zero_in_eval ? nullptr : g_img_zero, g_bytes, st));
  if (total_faces > 0) {
    tl::BinIn<T> in{};
    in.B = B;
    in.F = F;
    in.total_faces = total_faces;
    in.img = img;
    in.z = z;
    in.lay = FaceLayout{z_face_stride, z_vertex_stride, front_stride};
    in.valid = valid;
    in.front = front;
    in.mult = (T)multiplier;
    in.margin = (T)margin;
    in.multiplier = (float)multiplier;
    in.H = H;
    in.W = W;
    in.rec_r = rec_r;
    in.rec_s = rec_s;
    KAMD_LAUNCH_TIMED(kamd::K_BIN_FACES, (tl::bin_faces_kernel2<T, true, true>), dim3(kamd_cdiv(total_faces, 256)), dim3(256), 0, st, in, LR, LS);
  }
  KAMD_CHECK(hipGetLastError());
  tl::ClassifyOut co{};
  co.soft_mask = soft_mask;
  co.sub_touched = LS.sub_touched;
  co.big_count_s = LS.big_count;
  co.tiles_x_s = LS.tiles_x;
  co.ntiles_s = LS.ntiles;
  co.work_items = reinterpret_cast<uint4*>(work + tl::WORK_HEADER);
  co.work_counts = work;
  co.shard_cap = tl::work_shard_cap(B, H, W);
  co.tile_cov = reinterpret_cast<unsigned char*>(work + tl::work_cov_offset_words(B, H, W));
  KAMD_CHECK(kamd::raster2_draw<T>(st, B, H, W, D, F, (float)multiplier, eps, rec_r, LR, feat, interp, face_idx, weights, co,
                                   kamd_env_int("KAMD_DIBR_BG_WEIGHTS", 2) != 1));
  if (total_faces > 0)
    KAMD_CHECK(soft2_search_launch<T>(st, B, H, W, F, K, sigmainv, (float)multiplier, rec_s, LS, work, soft_mask, (T*)nullptr,
                                      (int64_t*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr, &list,
                                      (unsigned short*)((char*)workspace + lay.s.pixcnt),
                                      (T*)((char*)workspace + lay.s.prob_pm), zero_in_eval ? (void*)g_img_zero : nullptr,
                                      zero_in_eval ? g_bytes : 0, LR.row_span, work + tl::work_span_offset_words(B, H, W), 2 * B,
                                      work + tl::WORK_MAGIC_WORD, (unsigned int*)((char*)workspace + lay.s.over_count),
                                      (unsigned int*)((char*)workspace + lay.s.over_list), (unsigned int*)((char*)workspace + lay.s.ord_count),
                                      (uint4*)((char*)workspace + lay.s.ord_list)));
  KAMD_RETURN_LAST_ERROR();
}

// the rasterizer's and the soft mask's backward kernels are independent and both accumulate atomically into the same
// zero-initialised g_img; with KAMD_BWD_SIDE_STREAM=1 they run concurrently, the soft mask's on a library-owned side stream
// (the default until the rasterizer's backward learnt to skip empty tiles; see dibr_backward_fused)
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
std::mutex g_side_mu;
SideStream g_side[16];
// the side stream / events of the current device (created on first use); the caller holds g_side_mu while enqueuing
int side_stream(SideStream** out) {
  int dev = 0;
  KAMD_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
  SideStream& ss = g_side[dev];
  if (ss.s == nullptr) {
    KAMD_CHECK(hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking));
    KAMD_CHECK(hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming));
    KAMD_CHECK(hipEventCreateWithFlags(&ss.join, hipEventDisableTiming));
  }
  *out = &ss;
  return 0;
}

template <typename T>
int dibr_backward_fused(hipStream_t st, int B, int H, int W, int F, int D, int K, const T* grad_feat, const T* grad_soft,
                        const int64_t* face_idx, const T* weights, const T* soft_mask, const HitList2<T>& list,
                        unsigned int* work, const T* img, const T* feat, double multiplier, float eps, float sigmainv,
                        T* g_img, T* g_feat) {
  // The two backward kernels are independent and used to overlap on a side stream (61 || 73 us: ~110 together).  Since the
  // rasterizer's backward leaves empty tiles at once (48 us) the fork / join events and the contention cost more than the
  // overlap saves: one stream, step -9 us (KAMD_BWD_SIDE_STREAM=1 restores the side stream, for A/B runs; while the
  // profiler times EVERY kernel everything stays on `st` anyway, so that each event pair times one kernel alone).
  static const bool use_side = kamd_env_int("KAMD_BWD_SIDE_STREAM", 0) == 1;
  const unsigned char* tile_cov = kamd_env_int("KAMD_BWD_TILE_COV", 1) == 1  // (2: off, for A/B runs)
                                      ? reinterpret_cast<const unsigned char*>(work + tl::work_cov_offset_words(B, H, W))
                                      : nullptr;
  const unsigned int* row_centre = F > 0 ? work + tl::work_span_offset_words(B, H, W) : nullptr;  // (copied there by the forward's eval launch)
  // the rasterizer's backward walks the forward's list of covered tiles (KAMD_BWD_COV_LIST=2: one workgroup per tile, for A/B runs)
  static const bool cov_list = kamd_env_int("KAMD_BWD_COV_LIST", 1) == 1;
  if (!use_side || kamd::prof_all()) {
    // (the hot faces' partial sums -- tile_lists.h, WORK_BIGHASH_WORD -- are scratch inside the forward's work buffer: written by the
    // first launch, folded into g_img and cleared by the second; only when the second one is the list form)
    const bool list_form = cov_list && tile_cov != nullptr;
    unsigned int* bigwork = list_form && kamd_env_int("KAMD_BWD_BIG_SIDE", 1) == 1 ? work + tl::WORK_BIGHASH_WORD : nullptr;
    KAMD_CHECK(soft_mask_backward_list_launch<T>(st, B, H, W, F, K, grad_soft, soft_mask, list, work, img, multiplier, sigmainv,
                                                 (float)multiplier, g_img, bigwork));
    if (list_form)
      return kamd::raster_backward_draw_list<T>(st, B, H, W, F, D, grad_feat, face_idx, weights, img, feat, eps, g_img, g_feat,
                                                work + tl::WORK_COV_WORD, work + tl::work_covlist_offset_words(B, H, W),
                                                tl::cov_shard_cap((size_t)B, (size_t)tl::pass_geom(H, W, tl::R_TILE).ntiles),
                                                work + tl::WORK_MAGIC_WORD, bigwork);
    return kamd::raster_backward_draw<T>(st, B, H, W, F, D, grad_feat, face_idx, weights, img, feat, eps, g_img, g_feat, tile_cov,
                                         row_centre);
  }
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream* ss;
  KAMD_CHECK(side_stream(&ss));
  const hipStream_t side = ss->s;
  KAMD_CHECK(hipEventRecord(ss->fork, st));
  KAMD_CHECK(hipStreamWaitEvent(side, ss->fork, 0));
  // after the fork every exit goes through the join: the caller may free the buffers as soon as this returns
  int rc = soft_mask_backward_list_launch<T>(side, B, H, W, F, K, grad_soft, soft_mask, list, work, img, multiplier, sigmainv,
                                             (float)multiplier, g_img);
  int rc2 = (int)hipEventRecord(ss->join, side);
  if (rc == 0) rc = rc2;
  if (rc == 0)
    rc = kamd::raster_backward_draw<T>(st, B, H, W, F, D, grad_feat, face_idx, weights, img, feat, eps, g_img, g_feat, tile_cov,
                                       row_centre);
  rc2 = (int)hipStreamWaitEvent(st, ss->join, 0);
  return rc != 0 ? rc : rc2;
}

}  // namespace

#ifdef KAMD_PHASE_PROF
extern "C" int kamd_debug_phase_cycles_bin(unsigned long long* out16, int reset) {
  // (rows summed, except [10]: the longest wavefront = the rows' maximum)
  static unsigned long long host[PHASE_ROWS * 16];
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tl::g_phase_bin), sizeof(host));
  for (int i = 0; i < 16; ++i) {
    out16[i] = 0;
    for (int r = 0; r < PHASE_ROWS; ++r) out16[i] = i == 10 ? (host[r * 16 + i] > out16[i] ? host[r * 16 + i] : out16[i]) : out16[i] + host[r * 16 + i];
  }
  if (reset) {
    for (size_t i = 0; i < sizeof(host) / 8; ++i) host[i] = 0;
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(tl::g_phase_bin), host, sizeof(host));
  }
  return rc;
}
extern "C" int kamd_debug_phase_cycles(unsigned long long* out16, int reset) {
  int rc = 0;
  PHASE_READ(g_phase_select, out16, reset, rc);
  return rc;
}
#endif

namespace {
__global__ __launch_bounds__(64) void debug_transpose64_kernel(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out, int reference) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  out[i] = reference ? kamd::wave_transpose64_reference(in[i]) : kamd::wave_transpose64(in[i]);
}
}  // namespace

extern "C" {

int kamd_debug_transpose64(void* stream, int n_matrices, const uint64_t* in, uint64_t* out, int reference) {
  if (n_matrices <= 0) return 0;
  hipLaunchKernelGGL(debug_transpose64_kernel, dim3((unsigned)n_matrices), dim3(64), 0, (hipStream_t)stream,
                     (const unsigned long long*)in, (unsigned long long*)out, reference);
  return (int)hipGetLastError();
}

size_t kamd_dibr_soft_mask_lean_capacity(int B, int H, int W, int K) {
  // records the segmented hit list must be able to hold: every 16x4-pixel sub-tile slot of every 32x32 tile owns 64*K
  if (B <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
  // (the flat list's FLAT_SHARDS equal shards together: at least 64*K per sub-tile slot, which is also what the segmented pairs need)
  return (size_t)FLAT_SHARDS * flat_shard_cap_of(B, H, W, K);
}
size_t kamd_dibr_soft_mask_work_words(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return kamd::tl::work_words(B, H, W);
}
size_t kamd_dibr_soft_mask_forward_workspace(int B, int H, int W, int F, int K, int elem_size) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return kamd::tl::make_layout(B, H, W, (long long)B * (F > 0 ? F : 0), elem_size, false, true, K).total;
}
size_t kamd_dibr_rasterization_workspace(int B, int H, int W, int F, int K, int elem_size) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return kamd::tl::make_layout(B, H, W, (long long)B * (F > 0 ? F : 0), elem_size, true, true, K).total;
}

#define KAMD_SOFT_ENTRY(SFX, T)                                                                                       \
  int kamd_dibr_soft_mask_forward_##SFX(void* stream, int B, int H, int W, int F, int K, const T* img,               \
                                        const T* large_bbox, const int64_t* sel_idx, float sigmainv,                 \
                                        float multiplier, T* soft_mask, T* prob, int64_t* idx, uint8_t* type,         \
                                        void* workspace, uint8_t* hit_count, uint32_t* work) {                        \
    return soft_mask_forward_launch<T>((hipStream_t)stream, B, H, W, F, K, img, large_bbox, sel_idx, sigmainv,       \
                                       multiplier, soft_mask, prob, idx, type, workspace, hit_count, nullptr, work);  \
  }                                                                                                                   \
  int kamd_dibr_soft_mask_backward_##SFX(void* stream, int B, int H, int W, int F, int K, const T* grad,              \
                                         const T* soft_mask, const int64_t* sel_idx, const T* prob,                  \
                                         const int64_t* idx, const uint8_t* type, const T* img, float sigmainv,      \
                                         float multiplier, T* g_img, const uint8_t* hit_count) {                      \
    return soft_mask_backward_launch<T>((hipStream_t)stream, B, H, W, F, K, grad, soft_mask, sel_idx, prob, idx,     \
                                        type, img, sigmainv, multiplier, g_img, hit_count);                           \
  }                                                                                                                   \
  int kamd_dibr_soft_mask_forward_lean_##SFX(void* stream, int B, int H, int W, int F, int K, const T* img,          \
                                             const T* large_bbox, const int64_t* sel_idx, float sigmainv,            \
                                             float multiplier, T* soft_mask, int32_t* hit_pair, T* hit_prob,         \
                                             int32_t* hit_rec, int32_t* item_count, uint32_t* work,                 \
                                             void* workspace) {                                                       \
    HitList2<T> l{(int2*)hit_pair, hit_prob, (uint2*)hit_rec, item_count, work + FLAT_COUNT_WORD, flat_shard_cap_of(B, H, W, K)};                                                  \
    return soft_mask_forward_launch<T>((hipStream_t)stream, B, H, W, F, K, img, large_bbox, sel_idx, sigmainv,       \
                                       multiplier, soft_mask, nullptr, nullptr, nullptr, workspace, nullptr, &l,     \
                                       work);                                                                         \
  }                                                                                                                   \
  int kamd_dibr_soft_mask_backward_lean_##SFX(void* stream, int B, int H, int W, int F, int K, const T* grad,        \
                                              const T* soft_mask, const int32_t* hit_pair, const T* hit_prob,        \
                                              const int32_t* hit_rec, const int32_t* item_count,                    \
                                              const uint32_t* work, const T* img, double img_scale, float sigmainv,   \
                                              float multiplier, T* g_img) {                                           \
    HitList2<T> l{(int2*)hit_pair, (T*)hit_prob, (uint2*)hit_rec, (int*)item_count, (unsigned int*)work + FLAT_COUNT_WORD, flat_shard_cap_of(B, H, W, K)};                              \
    return soft_mask_backward_list_launch<T>((hipStream_t)stream, B, H, W, F, K, grad, soft_mask, l, work, img,      \
                                             img_scale, sigmainv, multiplier, g_img);                                 \
  }                                                                                                                   \
  int kamd_dibr_soft_mask_forward_fused_##SFX(void* stream, int B, int H, int W, int F, int K, const T* img,         \
                                              double multiplier, double margin, const int64_t* sel_idx,              \
                                              float sigmainv, T* soft_mask, int32_t* hit_pair, T* hit_prob,          \
                                              int32_t* hit_rec, int32_t* item_count, uint32_t* work,                \
                                              void* workspace) {                                                      \
    HitList2<T> l{(int2*)hit_pair, hit_prob, (uint2*)hit_rec, item_count, work + FLAT_COUNT_WORD, flat_shard_cap_of(B, H, W, K)};                                                  \
    return soft_mask_forward_launch<T>((hipStream_t)stream, B, H, W, F, K, img, nullptr, sel_idx, sigmainv,          \
                                       (float)multiplier, soft_mask, nullptr, nullptr, nullptr, workspace, nullptr,  \
                                       &l, work, true, multiplier, margin);                                           \
  }                                                                                                                   \
  int kamd_dibr_rasterization_forward_##SFX(                                                                          \
      void* stream, int B, int H, int W, int F, int D, int K, const T* z, int64_t z_face_stride,                      \
      int64_t z_vertex_stride, const T* img, const T* feat, const uint8_t* valid, const T* front,                     \
      int64_t front_stride, double multiplier, float eps, float sigmainv, double margin, T* interp, int64_t* face_idx, \
      T* weights, T* soft_mask, int32_t* hit_pair, T* hit_prob, int32_t* hit_rec, int32_t* item_count,               \
      uint32_t* work, void* workspace, T* grad_img_to_zero) {                                                         \
    HitList2<T> l{(int2*)hit_pair, hit_prob, (uint2*)hit_rec, item_count, work + FLAT_COUNT_WORD, flat_shard_cap_of(B, H, W, K)};                                                   \
    return dibr_forward_fused<T>((hipStream_t)stream, B, H, W, F, D, K, z, z_face_stride, z_vertex_stride, img, feat,  \
                                 valid, front, front_stride, multiplier, eps, sigmainv, margin, interp, face_idx,     \
                                 weights, soft_mask, l, work, workspace, grad_img_to_zero);                           \
  }                                                                                                                   \
  int kamd_dibr_rasterization_backward_##SFX(                                                                         \
      void* stream, int B, int H, int W, int F, int D, int K, const T* grad_feat, const T* grad_soft,                 \
      const int64_t* face_idx, const T* weights, const T* soft_mask, const int32_t* hit_pair, const T* hit_prob,      \
      const int32_t* hit_rec, const int32_t* item_count, uint32_t* work, const T* img, const T* feat,                \
      double multiplier, float eps, float sigmainv, T* g_img, T* g_feat) {                                            \
    HitList2<T> l{(int2*)hit_pair, (T*)hit_prob, (uint2*)hit_rec, (int*)item_count, (unsigned int*)work + FLAT_COUNT_WORD, flat_shard_cap_of(B, H, W, K)};                               \
    return dibr_backward_fused<T>((hipStream_t)stream, B, H, W, F, D, K, grad_feat, grad_soft, face_idx, weights,     \
                                  soft_mask, l, work, img, feat, multiplier, eps, sigmainv, g_img, g_feat);           \
  }
KAMD_SOFT_ENTRY(f32, float)
KAMD_SOFT_ENTRY(f64, double)
#undef KAMD_SOFT_ENTRY

}  // extern "C"
