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
#include <mutex>
#include <stdio.h>
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

// ---- K3 search --------------------------------------------------------------------------------------------
// The reference's loop "for each pixel: for each face" spends its time on the few silhouette-band pixels
// (C4: 187k of 8.4M pixels carry all 5.1M hits).  Two kernels:
//   soft_classify_kernel : one wavefront per 16x4-pixel sub-tile reads sel_idx, settles covered pixels (mask = 1)
//       and uncovered pixels of tiles no face touches (mask = 0), and queues every other sub-tile in a worklist;
//   soft_search_kernel   : a persistent grid of single-wavefront workgroups pulls sub-tiles from the worklist
//       (the band work is spread over all SIMDs instead of sitting in the few workgroups that happen to own it):
//       1. the 32x32 tile's bitmask is expanded into an ascending id list (popcount + wave scan, 64 words a step);
//       2. 64 ids at a time, each lane culls one face against the extent of the wavefront's uncovered pixels; the
//          survivors are compacted, in order, into a candidate queue;
//       3. per 64 candidates: each lane parks ITS face in LDS; then, lane = pixel, every lane walks the 64 boxes
//          and keeps the faces holding its pixel centre as a 64-bit mask; the first (knum - hits so far) set bits
//          of every pixel, in face order, form a PAIR list (slices assigned by a wave scan);
//       4. the pair list is evaluated with one (pixel, face) pair per lane -- every evaluated pair is an accepted
//          hit, lanes are fully used -- and written to consecutive K-buffer slots (or the compact hit list);
//       5. each pixel's owner lane continues prod(1 - prob) over its pairs in order.
// Results are identical to the reference's pixel-major loop: same expressions per (pixel, face), same order of
// hits per pixel, same product order.
constexpr int SM_WORDS = 32;      // bitmask words expanded per step
constexpr int SM_IDCAP = SM_WORDS * 32;  // ids one step can produce
constexpr int SM_PAIRCAP = 512;   // (pixel, face) pairs per evaluation window
constexpr int SM_SUBS = (TILE_W / SUB_W) * (TILE_H / SUB_H);  // 16 sub-tiles per tile

template <typename T>
struct HitList {      // compact output (our own autograd path): one record per (pixel, hit), order irrelevant
  int* pix;           // b * H * W + row * W + col
  int* face;
  T* prob;
  uint8_t* type;
  // Segmented: worklist entry i (a 16x4-pixel sub-tile) owns records [i*64*K, i*64*K + item_count[i]).  A shared
  // append counter would serialise ~20k same-address atomics per step (measured: ~8 ns each = the whole kernel).
  int* item_count;          // one per worklist entry
  unsigned int* n_items;    // number of worklist entries (written by the search kernel)
};

constexpr int CL_IDCAP = 2048;   // tile ids the classify workgroup can expand in LDS
constexpr int SM_CANDCAP = 512;  // candidate faces per work item handed to the search kernel (more: it expands itself)

template <typename T>
__global__ __launch_bounds__(TILE_THREADS) void soft_classify_kernel(
    int B, int F, TileGeom g, float multiplier, const T* __restrict__ rec, const unsigned int* __restrict__ masks,
    const uint8_t* __restrict__ sub_flags, const int64_t* __restrict__ sel_idx, T* __restrict__ soft_mask,
    uint8_t* __restrict__ hit_count, int* __restrict__ worklist, unsigned int* __restrict__ work_count,
    int* __restrict__ cand, int* __restrict__ cand_count) {
  // workgroup = one (tile, mesh): 16 wavefronts = its 16 sub-tiles.  Settles the trivial pixels, queues the sub-tiles
  // that need a search (ONE worklist atomic per workgroup) and -- because the 16 sub-tiles share the tile's bitmask --
  // expands that bitmask once and lets every queued wavefront cull it against its own uncovered pixels: the search
  // kernel then starts from a short candidate list instead of scanning F/32 mask words per sub-tile.
  __shared__ int s_slot[SM_SUBS];
  __shared__ int s_nneed;
  constexpr int IDCAP = sizeof(T) == 4 ? CL_IDCAP : CL_IDCAP / 2;  // ids + boxes of a tile in <= 40 KiB of LDS
  __shared__ int s_ids[IDCAP];
  __shared__ Box4<T> s_box[IDCAP];
  __shared__ int s_scan[TILE_THREADS / 64 + 1];
  const int b = blockIdx.x % B, tile = blockIdx.x / B;
  const int tid = threadIdx.x, sub = tid >> 6, lane = tid & 63;
  const int sub_x = (tile % g.tiles_x) * TILE_W + (sub & 1) * SUB_W;
  const int sub_y = (tile / g.tiles_x) * TILE_H + (sub >> 1) * SUB_H;
  const int col = sub_x + (lane & 15), row = sub_y + (lane >> 4);
  const bool in_image = col < g.W && row < g.H;
  const size_t p1 = ((size_t)b * g.H + row) * g.W + col;
  const bool uncovered = in_image && (int)sel_idx[in_image ? p1 : 0] < 0;
  const int item = (tile * B + b) * SM_SUBS + sub;
  const bool touched = sub_flags != nullptr && sub_flags[item] != 0;  // some enlarged box reaches this sub-tile
  if (in_image && (!uncovered || !touched)) {
    soft_mask[p1] = uncovered ? (T)(1.0 - 1.0) : (T)1.0;
    if (hit_count) hit_count[p1] = 0;
  }
  const bool need = touched && __any(uncovered);
  if (lane == 0) s_slot[sub] = need ? 1 : 0;
  __syncthreads();
  if (tid == 0) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < SM_SUBS; ++i) n += s_slot[i];
    unsigned int base = n > 0 ? atomicAdd(work_count, (unsigned int)n) : 0u;
#pragma unroll
    for (int i = 0; i < SM_SUBS; ++i) {
      if (s_slot[i]) {
        worklist[base] = (tile * B + b) * SM_SUBS + i;
        s_slot[i] = (int)base++;
      } else {
        s_slot[i] = -1;
      }
    }
    s_nneed = n;
  }
  __syncthreads();
  if (s_nneed == 0) return;
  const int slot = s_slot[sub];

  // expand the tile's bitmask (ascending ids) -- two words per thread
  const int64_t first_b = (int64_t)b * F;
  const int nwords = (F + 31) / 32;
  const unsigned int* tmask = masks + mask_base(g.ntiles, first_b, b, tile, nwords);
  bool overflow = nwords > 2 * TILE_THREADS;
  int total = 0;
  if (!overflow) {
    const int wi = 2 * tid;
    const unsigned int w0 = wi < nwords ? tmask[wi] : 0u, w1 = wi + 1 < nwords ? tmask[wi + 1] : 0u;
    const int excl = block_exclusive_scan(__popc(w0) + __popc(w1), s_scan, &total);
    overflow = total > IDCAP;
    if (!overflow) {
      int pos = excl;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned int wv = h == 0 ? w0 : w1;
        while (wv) {
          const int bit = __ffs(wv) - 1;
          wv &= wv - 1;
          s_ids[pos++] = (wi + h) * 32 + bit;
        }
      }
    }
  }
  __syncthreads();
  // the 16 wavefronts cull the same faces: their boxes are fetched once (one 16-byte load per face) and shared
  if (!overflow)
    for (int k = tid; k < total; k += TILE_THREADS)
      s_box[k] = *reinterpret_cast<const Box4<T>*>(rec + ((size_t)first_b + s_ids[k]) * REC_STRIDE);
  __syncthreads();
  if (slot < 0) return;  // (whole wavefront)
  if (overflow) {
    if (lane == 0) cand_count[slot] = -1;  // the search kernel expands the bitmask itself
    return;
  }
  // cull against the extent of this wavefront's uncovered pixels
  const T x0 = pixel_x(multiplier, g.W, col);
  const T y0 = pixel_y(multiplier, g.H, row);
  T ux_min = uncovered ? x0 : (T)INFINITY, ux_max = uncovered ? x0 : (T)-INFINITY;
  T uy_min = uncovered ? y0 : (T)INFINITY, uy_max = uncovered ? y0 : (T)-INFINITY;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ux_min = fmin(ux_min, __shfl_xor(ux_min, d, 64));
    ux_max = fmax(ux_max, __shfl_xor(ux_max, d, 64));
    uy_min = fmin(uy_min, __shfl_xor(uy_min, d, 64));
    uy_max = fmax(uy_max, __shfl_xor(uy_max, d, 64));
  }
  int* mine = cand + (size_t)slot * SM_CANDCAP;
  int ncand = 0;
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int k = t0 + lane;
    bool keep = false;
    int id = 0;
    if (k < total) {
      id = s_ids[k];
      const Box4<T> bb = s_box[k];
      keep = !((ux_max < bb.x0) | (ux_min >= bb.x1) | (uy_max < bb.y0) | (uy_min >= bb.y1));
    }
    const unsigned long long m = __ballot(keep);
    const int pos = ncand + __popcll(m & ((1ull << lane) - 1ull));
    if (keep && pos < SM_CANDCAP) mine[pos] = id;
    ncand += __popcll(m);
  }
  if (lane == 0) cand_count[slot] = ncand <= SM_CANDCAP ? ncand : -1;
}

template <typename T, bool LEAN>
__global__ __launch_bounds__(64) void soft_search_kernel(
    int B, int F, TileGeom g, int K, float sigmainv, float multiplier, const T* __restrict__ rec,
    const unsigned int* __restrict__ masks, const int* __restrict__ worklist, const unsigned int* __restrict__ work_count,
    const int* __restrict__ cand, const int* __restrict__ cand_count,
    const int64_t* __restrict__ sel_idx, T* __restrict__ soft_mask,
    T* __restrict__ prob_out, int64_t* __restrict__ idx_out, uint8_t* __restrict__ type_out,
    uint8_t* __restrict__ hit_count, HitList<T> list) {
  __shared__ int s_tmp[SM_IDCAP];
  __shared__ int s_cand[128];
  __shared__ T s_fv[6][64];  // the chunk's face vertices (structure of arrays: conflict-free gathers)
  __shared__ int s_fid[64];
  __shared__ unsigned short s_pair[SM_PAIRCAP];
  __shared__ T s_pr[SM_PAIRCAP];
  __shared__ int s_off[64];

  const int lane = threadIdx.x;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const unsigned int n_items = *work_count;

  // static round-robin over the COMPACTED worklist (every entry is real work, so this balances well; a shared
  // "next item" counter costs more in same-address atomic latency than the imbalance it removes: measured 117 us)
  if (LEAN && blockIdx.x == 0 && lane == 0) *list.n_items = n_items;
  for (unsigned int wi_ = blockIdx.x; wi_ < n_items; wi_ += gridDim.x) {
    const int item = worklist[wi_];
    const size_t list_base = (size_t)wi_ * 64 * (size_t)K;
    int item_pairs = 0;
    const int sub = item % SM_SUBS;
    const int b = (item / SM_SUBS) % B;
    const int tile = item / (SM_SUBS * B);
    const int sub_x = (tile % g.tiles_x) * TILE_W + (sub & 1) * SUB_W;
    const int sub_y = (tile / g.tiles_x) * TILE_H + (sub >> 1) * SUB_H;
    const int col = sub_x + (lane & 15), row = sub_y + (lane >> 4);
    const bool in_image = col < g.W && row < g.H;
    const size_t p1 = ((size_t)b * g.H + row) * g.W + col;
    const bool uncovered = in_image && (int)sel_idx[in_image ? p1 : 0] < 0;

    const int64_t first_b = (int64_t)b * F;
    const int nwords = (F + 31) / 32;
    const unsigned int* tmask = masks + mask_base(g.ntiles, first_b, b, tile, nwords);

    const T x0 = pixel_x(multiplier, g.W, col);
    const T y0 = pixel_y(multiplier, g.H, row);
    T ux_min = uncovered ? x0 : (T)INFINITY, ux_max = uncovered ? x0 : (T)-INFINITY;
    T uy_min = uncovered ? y0 : (T)INFINITY, uy_max = uncovered ? y0 : (T)-INFINITY;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      ux_min = fmin(ux_min, __shfl_xor(ux_min, d, 64));
      ux_max = fmax(ux_max, __shfl_xor(ux_max, d, 64));
      uy_min = fmin(uy_min, __shfl_xor(uy_min, d, 64));
      uy_max = fmax(uy_max, __shfl_xor(uy_max, d, 64));
    }

    // per-PIXEL state lives in the lane that owns the pixel
    int kid = 0;
    T all = 1.0;
    bool active = uncovered && K > 0;

    // takes the first n (<= 64) candidates of s_cand.
    //   a. lane = FACE: the pixels of the sub-tile inside its box (64-bit mask); vertices and id parked in LDS;
    //   b. a 64 x 64 bit transpose across the wavefront turns those into, lane = PIXEL, the faces holding its centre as a
    //      64-bit mask -- ascending face order for free; the first (knum - kid) set bits are this pixel's new hits;
    //   c. a wave scan of the hit counts gives every pixel a slice of the PAIR list; pairs are evaluated one per
    //      lane (every evaluated pair is an accepted hit) in windows of SM_PAIRCAP;
    //   d. each pixel folds the probabilities of its slice, in order, into prod(1 - prob).
    auto process_chunk = [&](int n) {
      unsigned long long inside = 0ull;  // lane = face: the pixels of this sub-tile whose centre its box holds
      if (lane < n) {
        const int id = s_cand[lane];
        const T* r = rec + ((size_t)first_b + id) * REC_STRIDE;
        const Box4<T> bb = *reinterpret_cast<const Box4<T>*>(r);
        inside = sub_tile_pixels_in_box<T>(bb, multiplier, g, sub_x, sub_y);
#pragma unroll
        for (int i = 0; i < 6; ++i) s_fv[i][lane] = r[4 + i];
        s_fid[lane] = id;
      }
      // lane = pixel: the faces (ascending) whose box holds its centre
      const unsigned long long hm = wave_transpose64(inside);
      __syncthreads();
      const int cnt = active ? min(__popcll(hm), K - kid) : 0;
      const int incl = wave_inclusive_scan(cnt);
      const int start = incl - cnt;
      const int total = __shfl(incl, 63, 64);
      if (total == 0) {
        __syncthreads();
        return;
      }
      s_off[lane] = kid - start;
      for (int lo = 0; lo < total; lo += SM_PAIRCAP) {
        const int np = min(SM_PAIRCAP, total - lo);
        {  // this pixel's pairs that fall into the window [lo, lo + np)
          unsigned long long m = hm;
          for (int i = 0; i < cnt; ++i) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int pos = start + i - lo;
            if (pos >= 0 && pos < np) s_pair[pos] = (unsigned short)((lane << 6) | k);
          }
        }
        __syncthreads();
        const size_t base = list_base + (size_t)item_pairs;
        item_pairs += np;
        for (int t0 = 0; t0 < np; t0 += 64) {
          const int t = t0 + lane;
          if (t < np) {
            const int pair = s_pair[t];
            const int u = pair >> 6, fs = pair & 63;
            const int ucol = sub_x + (u & 15), urow = sub_y + (u >> 4);
            const T xu = pixel_x(multiplier, g.W, ucol), yu = pixel_y(multiplier, g.H, urow);
            T v[6];
            EdgeInv<T> e[3];
            double den[3], rcp[3];
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = s_fv[i][fs];
            // the per-edge invariants are re-derived per pair (3 reciprocals) rather than parked in LDS: the kernel is
            // latency-bound and 12 KiB less LDS per wavefront lets a SIMD interleave more wavefronts
#pragma unroll
            for (int k = 0; k < 3; ++k)
              edge_invariants<T>(v[k * 2], v[k * 2 + 1], v[((k + 1) % 3) * 2], v[((k + 1) % 3) * 2 + 1], &e[k], &den[k], &rcp[k]);
            int which;
            const T d2 = closest_of_six<T>(v, e, den, rcp, xu, yu, multiplier, &which);
            const T zz = sigmainv * d2 / multiplier / multiplier;
            const T pr = dibr_exp<T>(-zz);
            s_pr[t] = pr;
            const size_t p1u = ((size_t)b * g.H + urow) * g.W + ucol;
            if (LEAN) {
              list.pix[base + t] = (int)p1u;
              list.face[base + t] = s_fid[fs];
              list.prob[base + t] = pr;
              list.type[base + t] = (uint8_t)(which + 1);
            } else {
              const size_t o = p1u * K + (size_t)(lo + t + s_off[u]);
              prob_out[o] = pr;
              idx_out[o] = s_fid[fs];
              type_out[o] = (uint8_t)(which + 1);
            }
          }
        }
        __syncthreads();
        // continue prod(1 - prob) in hit order (dibr_soft_mask_cuda.cu:174-179)
        for (int i = 0; i < cnt; ++i) {
          const int pos = start + i - lo;
          if (pos >= 0 && pos < np) all = (T)((double)all * (1.0 - (double)s_pr[pos]));
        }
        __syncthreads();
      }
      kid += cnt;
      if (kid >= K) active = false;
    };

    const int given = cand_count[wi_];
    if (given >= 0) {
      // the classify kernel already expanded and culled the tile's bitmask for this sub-tile
      const int* mine = cand + (size_t)wi_ * SM_CANDCAP;
      for (int c0 = 0; c0 < given; c0 += 64) {
        const int n = min(64, given - c0);
        __syncthreads();
        if (lane < n) s_cand[lane] = mine[c0 + lane];
        __syncthreads();
        process_chunk(n);
        if (!__any(active)) break;
      }
    } else {
      int ncand = 0;
      bool done = false;
      unsigned int next_word = (lane < SM_WORDS && lane < nwords) ? tmask[lane] : 0u;
      for (int w0 = 0; w0 < nwords && !done; w0 += SM_WORDS) {
        const int wi = w0 + lane;
        unsigned int word = next_word;
        next_word = (lane < SM_WORDS && wi + SM_WORDS < nwords) ? tmask[wi + SM_WORDS] : 0u;  // in flight while this step is processed
        const int c = __popc(word);
        const int incl = wave_inclusive_scan(c);
        const int total = __shfl(incl, 63, 64);
        if (total == 0) continue;
        __syncthreads();
        {
          int pos = incl - c;
          while (word) {
            const int bit = __ffs(word) - 1;
            word &= word - 1;
            s_tmp[pos++] = wi * 32 + bit;
          }
        }
        __syncthreads();
        for (int t0 = 0; t0 < total; t0 += 64) {
          const int k = t0 + lane;
          bool keep = false;
          int id = 0;
          if (k < total) {
            id = s_tmp[k];
            const T* r = rec + ((size_t)first_b + id) * REC_STRIDE;
            const T b0 = r[0], b1 = r[1], b2 = r[2], b3 = r[3];
            keep = !(ux_max < b0 || ux_min >= b2 || uy_max < b1 || uy_min >= b3);
          }
          const unsigned long long m = __ballot(keep);
          if (m == 0) continue;
          if (keep) s_cand[ncand + __popcll(m & lt_mask)] = id;
          ncand += __popcll(m);
          __syncthreads();
          if (ncand >= 64) {
            process_chunk(64);
            __syncthreads();
            const int rest = ncand - 64;
            const int moved = lane < rest ? s_cand[64 + lane] : 0;
            __syncthreads();
            if (lane < rest) s_cand[lane] = moved;
            ncand = rest;
            __syncthreads();
            if (!__any(active)) {
              done = true;
              break;
            }
          }
        }
      }
      if (!done && ncand > 0) {
        __syncthreads();
        process_chunk(ncand);
      }
    }
    if (uncovered) {
      soft_mask[p1] = (T)(1.0 - (double)all);
      if (!LEAN && hit_count) hit_count[p1] = (uint8_t)(kid > 255 ? 255 : kid);
    }
    if (LEAN && lane == 0) list.item_count[wi_] = item_pairs;
    __syncthreads();
  }
}

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
      const T d2 = up * up / (down + DIBR_EPS);
      const T dzdA = 2 * (x0 * up - d2 * A) / (down + DIBR_EPS);
      const T dzdB = 2 * (y0 * up - d2 * Bc) / (down + DIBR_EPS);
      const T dzdC = 2 * up / (down + DIBR_EPS);
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

// ---- K4, compact-list form (our own autograd path) ---------------------------------------------------------
// One workgroup per worklist entry (round-robin), one lane per recorded (pixel, hit).  An entry is one 16x4-pixel
// sub-tile, so its records touch few faces: contributions are summed per face in an LDS hash table and flushed
// with one global atomic per touched (face, coordinate).
constexpr int SL_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(SL_THREADS) void soft_mask_backward_list_kernel(
    int H, int W, int F, int K, const T* __restrict__ grad, const T* __restrict__ soft_mask, HitList<T> list,
    const T* __restrict__ img, T img_scale, float sigmainv, float multiplier, T* __restrict__ g_img) {
  __shared__ int s_key[SB_HT];
  __shared__ T s_acc[SB_HT * 6];
  const unsigned int n_items = *list.n_items;
  const long long P = (long long)H * W;
  for (unsigned int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int n = list.item_count[it];
    if (n <= 0) continue;
    const size_t base = (size_t)it * 64 * (size_t)K;
    __syncthreads();
    for (int i = threadIdx.x; i < SB_HT; i += SL_THREADS) s_key[i] = -1;
    for (int i = threadIdx.x; i < SB_HT * 6; i += SL_THREADS) s_acc[i] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += SL_THREADS) {
      const int pix = list.pix[base + t];
      const int f = list.face[base + t];
      const T pr = list.prob[base + t];
      const int e = (int)list.type[base + t] - 1;
      const int b = (int)(pix / P);
      const int rem = (int)(pix - (long long)b * P);
      const int col = rem % W, row = rem / W;
      const T x0 = pixel_x(multiplier, W, col);
      const T y0 = pixel_y(multiplier, H, row);
      const T dLdp = grad[pix];
      const T all = soft_mask[pix];
      const size_t s6 = ((size_t)b * F + f) * 6;
      const int key = (int)(((long long)b * F + f) & 0x7fffffff);
      const int slot = sb_find(s_key, key);
      const T dLdz = (T)(-1.0 * sigmainv * dLdp * (1.0 - all) / (1.0 - pr + DIBR_EPS) * pr);
      if (e >= 3) {
        const int o = (e - 3) * 2;
        const T x1 = img[s6 + o] * img_scale, y1 = img[s6 + o + 1] * img_scale;
        const T dLdx1 = dLdz * 2 * (x1 - x0);
        const T dLdy1 = dLdz * 2 * (y1 - y0);
        sb_add<T>(s_acc, g_img + s6, slot, o, (T)(dLdx1 / multiplier));
        sb_add<T>(s_acc, g_img + s6, slot, o + 1, (T)(dLdy1 / multiplier));
      } else {
        const int o = e * 2, o2 = ((e + 1) % 3) * 2;
        const T x1 = img[s6 + o] * img_scale, y1 = img[s6 + o + 1] * img_scale;
        const T x2 = img[s6 + o2] * img_scale, y2 = img[s6 + o2 + 1] * img_scale;
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
        sb_add<T>(s_acc, g_img + s6, slot, o, (T)(dLdx1 / multiplier));
        sb_add<T>(s_acc, g_img + s6, slot, o + 1, (T)(dLdy1 / multiplier));
        sb_add<T>(s_acc, g_img + s6, slot, o2, (T)(dLdx2 / multiplier));
        sb_add<T>(s_acc, g_img + s6, slot, o2 + 1, (T)(dLdy2 / multiplier));
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SB_HT * 6; i += SL_THREADS) {
      const int k = s_key[i / 6];
      const T v = s_acc[i];
      if (k >= 0 && v != (T)0) kamd_atomic_add(g_img + (size_t)k * 6 + (i % 6), v);
    }
  }
}

// The forward is two phases: BIN (memset + bin kernel: needs only the vertices) and SEARCH (classify + search: needs
// the rasterizer's face_idx).  `bin_st` lets the fused DIB-R entry point run the BIN phase on a side stream, concurrently
// with the rasterizer; with bin_st == st everything is stream-ordered as usual.
template <typename T>
int soft_mask_forward_launch(hipStream_t st, int B, int H, int W, int F, int K, const T* img, const T* large_bbox,
                             const int64_t* sel_idx, float sigmainv, float multiplier, T* soft_mask, T* prob,
                             int64_t* idx, uint8_t* type, void* workspace, uint8_t* hit_count, const HitList<T>* lean,
                             bool raw = false, double raw_multiplier = 1.0, double raw_margin = 0.0,
                             int phases = 3 /* bit 0: BIN, bit 1: SEARCH */) {
  // raw: `img` is the UNSCALED (B,F,3,2) operator input and `large_bbox` is unused; scaling by raw_multiplier and the
  // boxes enlarged by raw_margin (= boxlen * multiplier) are produced inside the bin kernel
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const TileGeom g = tile_geom(H, W);
  const long long total_faces = (long long)B * F;
  if (total_faces > 0 && workspace == nullptr) return (int)hipErrorInvalidValue;
  if (lean != nullptr && (long long)B * H * W >= (1ll << 31)) return (int)hipErrorInvalidValue;
  T* rec = (T*)workspace;
  unsigned int* masks = (unsigned int*)((char*)workspace + align256((size_t)total_faces * REC_STRIDE * sizeof(T)));
  unsigned int* flags = total_faces > 0 ? masks + mask_words(g.ntiles, B, total_faces) : nullptr;
  // after the tile flags: sub-tile flags (1 byte each), the worklist header {count, unused}, the worklist items
  const int n_sub = g.ntiles * B * SM_SUBS;
  uint8_t* sub_flags = total_faces > 0 ? (uint8_t*)(flags + flag_words(g.ntiles, B)) : nullptr;
  unsigned int* work = total_faces > 0 ? (unsigned int*)(sub_flags + align256((size_t)n_sub)) : nullptr;
  if (phases & 1) {
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
    if (total_faces > 0) {
      // rounded up to 16 bytes (one fill kernel, no tail memset): the few extra bytes are worklist items, written before read
      KAMD_CHECK(kamd_zero_async(masks, ((mask_words(g.ntiles, B, total_faces) + flag_words(g.ntiles, B) + 2) * 4 +
                                         align256((size_t)n_sub) + 15) & ~(size_t)15, st));
      kamd::ProfScope prof_(kamd::K_BIN_FACES, st);
      if (raw)
        hipLaunchKernelGGL(bin_faces_raw_kernel<T>, dim3(kamd_cdiv(total_faces, 256)), dim3(256), 0, st, B, F, img,
                           (const T*)nullptr, FaceLayout{3, 1, 1}, (const uint8_t*)nullptr, (const T*)nullptr,
                           (T)raw_multiplier, (T)raw_margin, g, multiplier, rec, masks, flags, sub_flags);
      else
        hipLaunchKernelGGL(bin_faces_kernel<T>, dim3(kamd_cdiv(total_faces, 256)), dim3(256), 0, st, B, F, total_faces,
                           (const int64_t*)nullptr, large_bbox, img, (const T*)nullptr, g, multiplier, rec, masks, flags,
                           sub_flags);
    }
    KAMD_CHECK(hipGetLastError());
  }
  if (phases & 2) {
    int* worklist = (int*)(work + 2);
    // after the worklist: per-item candidate counts and candidate lists (filled by the classify kernel)
    int* cand_count = worklist + n_sub;
    int* cand = (int*)((char*)cand_count + align256((size_t)n_sub * 4));
    {
      kamd::ProfScope prof_(kamd::K_SOFT_CLASSIFY, st);
      hipLaunchKernelGGL(soft_classify_kernel<T>, dim3(g.ntiles * B), dim3(TILE_THREADS), 0, st, B, F, g, multiplier, rec, masks,
                         sub_flags, sel_idx, soft_mask, lean ? (uint8_t*)nullptr : hit_count, worklist, work, cand,
                         cand_count);
    }
    KAMD_CHECK(hipGetLastError());
    if (total_faces > 0) {
      kamd::ProfScope prof_(kamd::K_SOFT_TILE, st);
      static const int per_cu_lean = kamd_env_int("KAMD_SOFT_SEARCH_PER_CU", 24);
      static const int per_cu_full = kamd_env_int("KAMD_SOFT_SEARCH_PER_CU", 24);
      if (getenv("KAMD_VERBOSE")) fprintf(stderr, "[kamd] soft_search per CU: lean %d full %d\n", per_cu_lean, per_cu_full);
      const int resident = KAMD_NUM_CU * (lean ? per_cu_lean : per_cu_full);
      const dim3 grid((unsigned)(n_sub < resident ? n_sub : resident));
      if (lean)
        hipLaunchKernelGGL((soft_search_kernel<T, true>), grid, dim3(64), 0, st, B, F, g, K, sigmainv, multiplier, rec,
                           masks, worklist, work, cand, cand_count, sel_idx, soft_mask, (T*)nullptr, (int64_t*)nullptr,
                           (uint8_t*)nullptr, (uint8_t*)nullptr, *lean);
      else
        hipLaunchKernelGGL((soft_search_kernel<T, false>), grid, dim3(64), 0, st, B, F, g, K, sigmainv, multiplier,
                           rec, masks, worklist, work, cand, cand_count, sel_idx, soft_mask, prob, idx, type, hit_count,
                           HitList<T>{});
    }
  }
  KAMD_RETURN_LAST_ERROR();
}

template <typename T>
int soft_mask_backward_list_launch(hipStream_t st, int B, int H, int W, int F, int K, const T* grad, const T* soft_mask,
                                   const HitList<T>& list, const T* img, double img_scale, float sigmainv,
                                   float multiplier, T* g_img) {
  if ((long long)B * H * W <= 0 || F <= 0) return 0;
  {
    kamd::ProfScope prof_(kamd::K_SOFT_BACKWARD_LIST, st);
    // (10 per CU although 6 fit: in the fused backward this kernel shares the GPU with raster_backward, whose one-shot
    // workgroups need the slots the surplus leaves free at the start; a one-resident-set grid is unmeasured there)
    hipLaunchKernelGGL(soft_mask_backward_list_kernel<T>, dim3(KAMD_NUM_CU * kamd_env_int("KAMD_SOFT_BWD_PER_CU", 16)), dim3(SL_THREADS), 0, st, H, W, F, K, grad,
                       soft_mask, list, img, (T)img_scale, sigmainv, multiplier, g_img);
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

}  // namespace

extern "C" {

size_t kamd_dibr_soft_mask_lean_capacity(int B, int H, int W, int K) {
  // records the segmented hit list must be able to hold: every 16x4-pixel sub-tile of every image owns 64*K slots
  if (B <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
  return (size_t)B * ((W + kamd::SUB_W - 1) / kamd::SUB_W) * ((H + kamd::SUB_H - 1) / kamd::SUB_H) * 64 * (size_t)K;
}

size_t kamd_dibr_soft_mask_forward_workspace(int B, int H, int W, int F, int elem_size) {
  if (B <= 0 || H <= 0 || W <= 0 || F <= 0) return 0;
  // bins + the sub-tile worklist {count, next, items[B * ntiles * 16]}
  const kamd::TileGeom g = kamd::tile_geom(H, W);
  const size_t n_sub = (size_t)g.ntiles * B * 16;
  return kamd::bins_workspace_bytes(B, H, W, (long long)B * F, elem_size) + kamd::align256(n_sub) +
         kamd::align256((n_sub + 2) * 4) + kamd::align256(n_sub * 4) + n_sub * 512 * 4 + 256;
}

int kamd_dibr_soft_mask_forward_f32(void* stream, int B, int H, int W, int F, int K, const float* img,
                                    const float* large_bbox, const int64_t* sel_idx, float sigmainv, float multiplier,
                                    float* soft_mask, float* prob, int64_t* idx, uint8_t* type, void* workspace,
                                    uint8_t* hit_count) {
  return soft_mask_forward_launch<float>((hipStream_t)stream, B, H, W, F, K, img, large_bbox, sel_idx, sigmainv,
                                         multiplier, soft_mask, prob, idx, type, workspace, hit_count, nullptr);
}
int kamd_dibr_soft_mask_forward_f64(void* stream, int B, int H, int W, int F, int K, const double* img,
                                    const double* large_bbox, const int64_t* sel_idx, float sigmainv, float multiplier,
                                    double* soft_mask, double* prob, int64_t* idx, uint8_t* type, void* workspace,
                                    uint8_t* hit_count) {
  return soft_mask_forward_launch<double>((hipStream_t)stream, B, H, W, F, K, img, large_bbox, sel_idx, sigmainv,
                                          multiplier, soft_mask, prob, idx, type, workspace, hit_count, nullptr);
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


#define KAMD_LEAN_ENTRY(SFX, T)                                                                                       \
  int kamd_dibr_soft_mask_forward_lean_##SFX(void* stream, int B, int H, int W, int F, int K, const T* img,          \
                                             const T* large_bbox, const int64_t* sel_idx, float sigmainv,            \
                                             float multiplier, T* soft_mask, int32_t* hit_pix, int32_t* hit_face,    \
                                             T* hit_prob, uint8_t* hit_type, int32_t* item_count,                    \
                                             uint32_t* n_items, void* workspace) {                                   \
    HitList<T> l{hit_pix, hit_face, hit_prob, hit_type, item_count, n_items};                                        \
    return soft_mask_forward_launch<T>((hipStream_t)stream, B, H, W, F, K, img, large_bbox, sel_idx, sigmainv,       \
                                       multiplier, soft_mask, nullptr, nullptr, nullptr, workspace, nullptr, &l);    \
  }                                                                                                                   \
  int kamd_dibr_soft_mask_backward_lean_##SFX(void* stream, int B, int H, int W, int F, int K, const T* grad,        \
                                              const T* soft_mask, const int32_t* hit_pix, const int32_t* hit_face,   \
                                              const T* hit_prob, const uint8_t* hit_type, const int32_t* item_count,  \
                                              const uint32_t* n_items, const T* img, double img_scale,               \
                                              float sigmainv, float multiplier, T* g_img) {                           \
    HitList<T> l{(int*)hit_pix, (int*)hit_face, (T*)hit_prob, (uint8_t*)hit_type, (int*)item_count,                  \
                 (unsigned int*)n_items};                                                                             \
    return soft_mask_backward_list_launch<T>((hipStream_t)stream, B, H, W, F, K, grad, soft_mask, l, img,            \
                                             img_scale, sigmainv, multiplier, g_img);                                 \
  }                                                                                                                   \
  int kamd_dibr_soft_mask_forward_fused_##SFX(void* stream, int B, int H, int W, int F, int K, const T* img,         \
                                              double multiplier, double margin, const int64_t* sel_idx,              \
                                              float sigmainv, T* soft_mask, int32_t* hit_pix, int32_t* hit_face,     \
                                              T* hit_prob, uint8_t* hit_type, int32_t* item_count,                   \
                                              uint32_t* n_items, void* workspace) {                                  \
    HitList<T> l{hit_pix, hit_face, hit_prob, hit_type, item_count, n_items};                                        \
    return soft_mask_forward_launch<T>((hipStream_t)stream, B, H, W, F, K, img, nullptr, sel_idx, sigmainv,          \
                                       (float)multiplier, soft_mask, nullptr, nullptr, nullptr, workspace, nullptr,  \
                                       &l, true, multiplier, margin);                                                 \
  }
KAMD_LEAN_ENTRY(f32, float)
KAMD_LEAN_ENTRY(f64, double)
#undef KAMD_LEAN_ENTRY

}  // extern "C"

// ---- fused DIB-R front door: rasterize + soft mask in one call, independent kernels on two streams ---------------
namespace {
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
int dibr_forward_fused(hipStream_t st, int B, int H, int W, int F, int D, int K, const T* z, int64_t z_face_stride,
                       int64_t z_vertex_stride, const T* img, const T* feat, const uint8_t* valid, const T* front,
                       int64_t front_stride, double multiplier, float eps, float sigmainv, double margin, T* interp,
                       int64_t* face_idx, T* weights, T* soft_mask, const HitList<T>& list, void* ws_raster, void* ws_soft) {
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream* ss;
  KAMD_CHECK(side_stream(&ss));
  // the soft mask's BIN phase needs only the vertices: side stream, concurrently with the rasterizer.  (While the
  // profiler times EVERY kernel everything stays on `st`, so that each kernel's event pair times that kernel alone.)
  const hipStream_t side = kamd::prof_all() ? st : ss->s;
  KAMD_CHECK(hipEventRecord(ss->fork, st));
  KAMD_CHECK(hipStreamWaitEvent(side, ss->fork, 0));
  KAMD_CHECK(soft_mask_forward_launch<T>(side, B, H, W, F, K, img, nullptr, nullptr, sigmainv, (float)multiplier, soft_mask,
                                         nullptr, nullptr, nullptr, ws_soft, nullptr, &list, true, multiplier, margin, 1));
  KAMD_CHECK(hipEventRecord(ss->join, side));
  int rc;
  if (sizeof(T) == 4)
    rc = kamd_rasterize_forward_fused_strided_f32(st, B, H, W, F, D, (const float*)z, z_face_stride, z_vertex_stride,
                                                  (const float*)img, (const float*)feat, valid, (const float*)front,
                                                  front_stride, multiplier, eps, (float*)interp, face_idx, (float*)weights,
                                                  ws_raster);
  else
    rc = kamd_rasterize_forward_fused_strided_f64(st, B, H, W, F, D, (const double*)z, z_face_stride, z_vertex_stride,
                                                  (const double*)img, (const double*)feat, valid, (const double*)front,
                                                  front_stride, multiplier, eps, (double*)interp, face_idx,
                                                  (double*)weights, ws_raster);
  KAMD_CHECK(rc);
  KAMD_CHECK(hipStreamWaitEvent(st, ss->join, 0));
  return soft_mask_forward_launch<T>(st, B, H, W, F, K, img, nullptr, face_idx, sigmainv, (float)multiplier, soft_mask,
                                     nullptr, nullptr, nullptr, ws_soft, nullptr, &list, true, multiplier, margin, 2);
}

template <typename T>
int dibr_backward_fused(hipStream_t st, int B, int H, int W, int F, int D, int K, const T* grad_feat, const T* grad_soft,
                        const int64_t* face_idx, const T* weights, const T* soft_mask, const HitList<T>& list, const T* img,
                        const T* feat, double multiplier, float eps, float sigmainv, T* g_img, T* g_feat) {
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream* ss;
  KAMD_CHECK(side_stream(&ss));
  // the two backward kernels are independent and both accumulate atomically into the same zero-initialised g_img
  const hipStream_t side = kamd::prof_all() ? st : ss->s;
  KAMD_CHECK(hipEventRecord(ss->fork, st));
  KAMD_CHECK(hipStreamWaitEvent(side, ss->fork, 0));
  KAMD_CHECK(soft_mask_backward_list_launch<T>(side, B, H, W, F, K, grad_soft, soft_mask, list, img, multiplier, sigmainv,
                                               (float)multiplier, g_img));
  KAMD_CHECK(hipEventRecord(ss->join, side));
  int rc;
  if (sizeof(T) == 4)
    rc = kamd_rasterize_backward_f32(st, B, H, W, F, D, (const float*)grad_feat, face_idx, (const float*)weights,
                                     (const float*)img, (const float*)feat, eps, (float*)g_img, (float*)g_feat);
  else
    rc = kamd_rasterize_backward_f64(st, B, H, W, F, D, (const double*)grad_feat, face_idx, (const double*)weights,
                                     (const double*)img, (const double*)feat, eps, (double*)g_img, (double*)g_feat);
  KAMD_CHECK(rc);
  return (int)hipStreamWaitEvent(st, ss->join, 0);
}
}  // namespace

extern "C" {
#define KAMD_DIBR_ENTRY(SFX, T)                                                                                       \
  int kamd_dibr_rasterization_forward_##SFX(                                                                          \
      void* stream, int B, int H, int W, int F, int D, int K, const T* z, int64_t z_face_stride,                      \
      int64_t z_vertex_stride, const T* img, const T* feat, const uint8_t* valid, const T* front,                     \
      int64_t front_stride, double multiplier, float eps, float sigmainv, double margin, T* interp, int64_t* face_idx, \
      T* weights, T* soft_mask, int32_t* hit_pix, int32_t* hit_face, T* hit_prob, uint8_t* hit_type,                  \
      int32_t* item_count, uint32_t* n_items, void* ws_raster, void* ws_soft) {                                       \
    HitList<T> l{hit_pix, hit_face, hit_prob, hit_type, item_count, n_items};                                         \
    return dibr_forward_fused<T>((hipStream_t)stream, B, H, W, F, D, K, z, z_face_stride, z_vertex_stride, img, feat,  \
                                 valid, front, front_stride, multiplier, eps, sigmainv, margin, interp, face_idx,     \
                                 weights, soft_mask, l, ws_raster, ws_soft);                                          \
  }                                                                                                                   \
  int kamd_dibr_rasterization_backward_##SFX(                                                                         \
      void* stream, int B, int H, int W, int F, int D, int K, const T* grad_feat, const T* grad_soft,                 \
      const int64_t* face_idx, const T* weights, const T* soft_mask, const int32_t* hit_pix, const int32_t* hit_face,  \
      const T* hit_prob, const uint8_t* hit_type, const int32_t* item_count, const uint32_t* n_items, const T* img,   \
      const T* feat, double multiplier, float eps, float sigmainv, T* g_img, T* g_feat) {                             \
    HitList<T> l{(int*)hit_pix, (int*)hit_face, (T*)hit_prob, (uint8_t*)hit_type, (int*)item_count,                   \
                 (unsigned int*)n_items};                                                                             \
    return dibr_backward_fused<T>((hipStream_t)stream, B, H, W, F, D, K, grad_feat, grad_soft, face_idx, weights,     \
                                  soft_mask, l, img, feat, multiplier, eps, sigmainv, g_img, g_feat);                 \
  }
KAMD_DIBR_ENTRY(f32, float)
KAMD_DIBR_ENTRY(f64, double)
#undef KAMD_DIBR_ENTRY
}  // extern "C"

