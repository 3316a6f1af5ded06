// DIB-R rasterizer forward / backward for MI355X (gfx950).
//
// Replaces kaolin/csrc/render/mesh/rasterization_cuda.cu:43-236 (K1) and :238-442 (K2) behind the C ABI
// of include/kaolin_amd.h.  Semantics kept from the reference (restated in oracle/dibr_oracle.inc):
//   K1  per pixel, faces in ascending packed index: half-open bbox reject, edge functions on
//       (vertex - pixel), norm += copysign((double)eps, norm), inside iff all w >= 0,
//       z0 = w0*az + w1*bz + w2*cz, strictly larger z0 wins (ties keep the lowest index).
//   K2  per covered pixel: grad*w into the face's 3xD feature slots, barycentric Jacobian into its
//       3x2 vertex slots (atomic accumulation into caller-zeroed outputs).
// Arithmetic: compiled with -ffp-contract=off, every expression in the reference's operand order and
// types, so face_idx is bit-exact against the oracle.
//
// MI355X design: see tile_lists.h / raster2.inc.  K1 = face binning into per-tile lists (count, scan, emit) + raster_tile_kernel2;
// every output element is written by the tile kernel (uncovered pixels get -1 / 0), so no pre-fill pass over the G-buffer
// is needed.  A 16x4-pixel sub-tile per wavefront makes each row of the G-buffer a 128-B (idx), 192-B (weights) or
// 64*D/4-B (features) contiguous store per wavefront.
#include "common.h"
#include "profile.h"
#include "tile_bins.h"
#include "tile_lists.h"
#include "dibr_internal.h"
#include "phase_prof.h"
#include "../../include/kaolin_amd.h"

namespace {
using namespace kamd;

#include "raster2.inc"

// ---- K2 -------------------------------------------------------------------------------------------------
// Per covered pixel the reference issues 3*D + 6*D float atomics on addresses shared by every pixel of the same
// face (rasterization_cuda.cu:283,391-398): the run time is atomic contention (measured 2.4 ms for C4).  Here a
// workgroup owns a 16x16-pixel block: every lane computes its pixel's 6 + 3*D contributions, runs of one face are merged
// in registers (down the columns, then along the rows: DPP), and every run leaves as ONE atomic request.

// numerators of d(w1)/d(.) and d(w2)/d(.) for the six vertex coordinates (ax, ay, bx, by, cx, cy), and k3
// (rasterization_cuda.cu:287-371); the common 1/k3^2 is applied by the caller
template <typename T>
__device__ __forceinline__ T barycentric_jacobian(const T* v, T aw, T bw, T cw, float eps, T* dw1, T* dw2) {
  const T ax = v[0], ay = v[1], bx = v[2], by = v[3], cx = v[4], cy = v[5];
  const T x0 = aw * ax + bw * bx + cw * cx;
  const T y0 = aw * ay + bw * by + cw * cy;
  const T m = bx - ax, p = by - ay, n = cx - ax, q = cy - ay, s = x0 - ax, t = y0 - ay;
  const T k1 = s * q - n * t;
  const T k2 = m * t - s * p;
  T k3 = m * q - n * p;
  k3 = (T)((double)k3 + copysign((double)eps, (double)k3));
  // dk_i * k3 - dk3 * k_i with the reference's explicit zero terms kept (0 * k3 - q * k1 ...)
  const T zero = 0;
  const T dw1dm = zero * k3 - q * k1, dw1dn = (-t) * k3 - (-p) * k1;
  const T dw1dp = zero * k3 - (-n) * k1, dw1dq = s * k3 - m * k1;
  const T dw1ds = q * k3 - zero * k1, dw1dt = (-n) * k3 - zero * k1;
  const T dw2dm = t * k3 - q * k2, dw2dn = zero * k3 - (-p) * k2;
  const T dw2dp = (-s) * k3 - (-n) * k2, dw2dq = zero * k3 - m * k2;
  const T dw2ds = (-p) * k3 - zero * k2, dw2dt = m * k3 - zero * k2;
  dw1[0] = -(dw1dm + dw1dn + dw1ds);
  dw1[1] = -(dw1dp + dw1dq + dw1dt);
  dw1[2] = dw1dm;
  dw1[3] = dw1dp;
  dw1[4] = dw1dn;
  dw1[5] = dw1dq;
  dw2[0] = -(dw2dm + dw2dn + dw2ds);
  dw2[1] = -(dw2dp + dw2dq + dw2dt);
  dw2[2] = dw2dm;
  dw2[3] = dw2dp;
  dw2[4] = dw2dn;
  dw2[5] = dw2dq;
  return k3;
}

// How a tile's per-face sums reach memory (measured at C4, profiles/r03n / r03o): global float atomics cost ~60 ps per REQUEST (a line
// touched by an instruction) chip-wide, whatever the lanes in it -- run-end lanes adding value by value: 219 us; runs merged per face
// in a per-tile LDS hash table first (a CAS probe + six LDS float atomics per run: 20 us of LDS pipe): 50, 46 with the columns merged
// down the rows first; every wavefront staging its runs' totals in LDS rows of its own and sending ONE atomic request per run (a face's
// values in consecutive lanes), columns merged first: 41-43 -- the form below.
#ifndef KAMD_RBWD_ORDER
#define KAMD_RBWD_ORDER 1  // workgroup order of the backward: 1 = views interleaved, tile rows from the middle of the image outwards
#endif                     // (as the forward's tile kernel; 0 = view-major, row-major: 49.4 vs 45.4 us at C4, 9 us of the step with feature gradients)
// DT > 0: feature count known at compile time (block-merged through LDS); DT == 0: any D, per-lane global atomics.
// GF = false: the caller does not need d/d(face_features) (static texture coordinates, the usual DIB-R set-up): only the
// 6 image-coordinate values per face are merged instead of 6 + 3*D -- 2.5x fewer DPP merges and LDS atomics at D = 3.
PHASE_TABLE(g_phase_rbwd)
#ifdef KAMD_PHASE_PROF
#define RBWD_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")  // (profiling builds: a phase ends when its memory operations have)
#else
#define RBWD_DRAIN()
#endif

// one 16 x 16-pixel tile of image b: workgroup = the tile, wavefront = 16 x 4 pixels.  Called by every thread of the workgroup
// (barriers inside); may be called for one tile after another (its LDS tables are re-initialised behind a barrier).
template <typename T, int DT, bool GF>
__device__ __forceinline__ void raster_backward_tile(
    int b, int tile, int tiles_x, int H, int W, int F, int D, const T* __restrict__ grad, const int64_t* __restrict__ face_idx,
    const T* __restrict__ weights, const T* __restrict__ img, const T* __restrict__ feat, float eps,
    T* __restrict__ g_img, T* __restrict__ g_feat) {
  constexpr int NV = (DT > 0 && GF) ? 6 + 3 * DT : 6;
  constexpr bool STAGED = DT > 0;                              // run totals go to global memory through per-wavefront staging rows
  __shared__ int s_runf[STAGED ? 4 : 1][STAGED ? 64 : 1];      // the runs' faces ...
  __shared__ T s_runv[STAGED ? 4 : 1][STAGED ? 64 * NV : 1];   // ... and their NV sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = (tile % tiles_x) * 16 + (lane & 15), row = (tile / tiles_x) * 16 + wave * 4 + (lane >> 4);
  const bool in_image = col < W && row < H;
  PHASE_DECL;
  const size_t tp = ((size_t)b * H + row) * W + col;
  const int f = in_image ? (int)face_idx[tp] : -1;
  if (DT > 0) {
    if (__ballot(f >= 0) == 0ull) return;  // nothing covered in this wavefront's 16 x 4 pixels (no workgroup-wide state: wavefronts are on their own)
  }
  T vals[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) vals[i] = 0;
  if (f >= 0) {
    const size_t tf = (size_t)b * F + (size_t)f;
    const T aw = weights[tp * 3 + 0], bw = weights[tp * 3 + 1], cw = weights[tp * 3 + 2];
    const T* g = grad + tp * D;
    T dw1[6], dw2[6];
    const T k3 = barycentric_jacobian<T>(img + tf * 6, aw, bw, cw, eps, dw1, dw2);
    const T* ff = feat + tf * 3 * D;
    const int nd = DT > 0 ? DT : D;
    for (int d = 0; d < nd; ++d) {
      const T gd = g[d];
      const T c0 = ff[d], c1 = ff[D + d], c2 = ff[2 * D + d];
      const T dldI = gd / (k3 * k3);
#pragma unroll
      for (int j = 0; j < 6; ++j) vals[j] += (T)(dldI * ((c1 - c0) * dw1[j] + (c2 - c0) * dw2[j]));
      if constexpr (!GF) {
      } else if constexpr (DT > 0) {
        vals[6 + d] = (T)(gd * aw);
        vals[6 + DT + d] = (T)(gd * bw);
        vals[6 + 2 * DT + d] = (T)(gd * cw);
      } else {
        kamd_atomic_add(g_feat + (tf * 3 + 0) * D + d, (T)(gd * aw));
        kamd_atomic_add(g_feat + (tf * 3 + 1) * D + d, (T)(gd * bw));
        kamd_atomic_add(g_feat + (tf * 3 + 2) * D + d, (T)(gd * cw));
      }
    }
    if constexpr (DT == 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j) kamd_atomic_add(g_img + tf * 6 + j, vals[j]);
    }
  }
  RBWD_DRAIN();
  PHASE_MARK(2);
  if constexpr (DT > 0) {
    // Pixels of one face are neighbours: in lane order (16 per row) they form runs.  A segmented inclusive scan sums
    // every run (6 shuffle steps per value), and only the LAST lane of a run touches the LDS table: same-address LDS
    // atomics were 56 % of this kernel's wave time (SQ_WAIT_INST_LDS) when every lane added on its own.
    if (__ballot(f >= 0) != 0ull) {
    // rows of 16 lanes = 16 horizontally adjacent pixels: runs are merged inside a row with DPP row shifts (register
    // moves).  The 64-lane version went through ds_bpermute: 96 LDS-crossbar operations per wavefront, and this kernel
    // spent 42 % of its wave cycles waiting on LDS (SQ_WAIT_INST_LDS).
    const int rl = lane & 15;
    int fm = f;  // the face this lane's sums still belong to (-1 once they have been handed to the lane below)
    {
      // A face covers a few pixels in each of two or three rows: before the rows' runs are merged, every COLUMN of the
      // wavefront's 4 rows is merged downwards -- row r takes over the sums of row r - 1 where both hold the same face
      // (three steps of one ds_bpermute per value), and a lane whose sums moved down drops out.  A compact blob then ends
      // as ONE run in its lowest row instead of one per row: fewer table inserts / atomic requests per wavefront.
      const int wr = lane >> 4;
      const int up_f = __shfl_up(f, 16, 64);
      const bool same_up = wr >= 1 && f >= 0 && up_f == f;
#pragma unroll
      for (int step = 1; step <= 3; ++step) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const T o = __shfl_up(vals[i], 16, 64);
          if (wr == step && same_up) vals[i] += o;
        }
      }
      const int down_f = __shfl_down(f, 16, 64);
      if (wr < 3 && f >= 0 && down_f == f) fm = -1;
    }
    const int prev_f = row_shr<1>(fm);
    const int next_f = row_shl<1>(fm);
    const bool run_start = rl == 0 || prev_f != fm;
    int start_lane = run_start ? rl : 0;
    {
      int o;
      o = row_shr<1>(start_lane); if (rl >= 1) start_lane = max(start_lane, o);
      o = row_shr<2>(start_lane); if (rl >= 2) start_lane = max(start_lane, o);
      o = row_shr<4>(start_lane); if (rl >= 4) start_lane = max(start_lane, o);
      o = row_shr<8>(start_lane); if (rl >= 8) start_lane = max(start_lane, o);
    }
#define KAMD_RB_STAGE(DD)                                        \
    {                                                            \
      const bool take = rl >= DD && start_lane <= rl - DD;       \
      _Pragma("unroll") for (int i = 0; i < NV; ++i) {           \
        const T o = row_shr<DD>(vals[i]);                        \
        if (take) vals[i] += o;                                  \
      }                                                          \
    }
    KAMD_RB_STAGE(1)
    KAMD_RB_STAGE(2)
    KAMD_RB_STAGE(4)
    KAMD_RB_STAGE(8)
#undef KAMD_RB_STAGE
    PHASE_MARK(3);
    const bool run_end = rl == 15 || next_f != fm;
    // The run totals leave through a per-wavefront LDS staging row so that ONE atomic instruction carries all the values of a
    // face in consecutive lanes (24 contiguous bytes = one L2 atomic request per face; issued value by value from the
    // run-end lanes, every instruction touches a different line per lane: 219 us, r03n).  No workgroup-wide state: no
    // table to clear, no barrier, no LDS atomics (the round-2 table's cost 20 of the kernel's 49 us).
    const unsigned long long ends = __ballot(fm >= 0 && run_end);
    const int n_ends = __popcll(ends);
    wave_lds_fence();  // (the previous tile's readers of this wavefront's rows are done)
    if (fm >= 0 && run_end) {
      const int q = __popcll(ends & ((1ull << lane) - 1ull));
      s_runf[wave][q] = fm;
#pragma unroll
      for (int i = 0; i < NV; ++i) s_runv[wave][q * NV + i] = vals[i];
    }
    wave_lds_fence();
    for (int j = lane; j < n_ends * NV; j += 64) {
      const int q = j / NV, c = j - q * NV;
      const size_t tf = (size_t)b * F + (size_t)s_runf[wave][q];
      const T v = s_runv[wave][j];
      if (c < 6)
        kamd_atomic_add(g_img + tf * 6 + c, v);
      else if constexpr (GF)
        kamd_atomic_add(g_feat + tf * 3 * D + (c - 6), v);
    }
    }
  }
  PHASE_FLUSH(g_phase_rbwd);
}

template <typename T, int DT, bool GF>
__global__ __launch_bounds__(256) void raster_backward_kernel(
    int B, int H, int W, int F, int D, const T* __restrict__ grad, const int64_t* __restrict__ face_idx,
    const T* __restrict__ weights, const T* __restrict__ img, const T* __restrict__ feat, float eps,
    T* __restrict__ g_img, T* __restrict__ g_feat, const unsigned char* __restrict__ tile_cov,
    const unsigned int* __restrict__ row_span) {
  // (fused dibr_rasterization: the forward pass noted which tiles hold a covered pixel -- 85 % of C4's do not, and
  // finding that out from face_idx costs a 2-KB read and a barrier per workgroup: 24 of this kernel's 60 us)
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
#if KAMD_RBWD_ORDER
  // views interleaved, a view's tile rows in the forward pass' order (from the middle outwards, shifted to start in the middle
  // of the covered rows): the workgroups that find covered pixels start first
  const int b = blockIdx.x % B, k_ = blockIdx.x / B, kr_ = k_ / tiles_x;
  const int tile = tl::row_of_order(kr_, tl::row_centre(row_span, b, tiles_y), tiles_y) * tiles_x + (k_ - kr_ * tiles_x);
#else
  const int tile = blockIdx.x % (tiles_x * tiles_y), b = blockIdx.x / (tiles_x * tiles_y);
#endif
  if (tile_cov != nullptr && tile_cov[(size_t)b * (tiles_x * tiles_y) + tile] == 0) return;
  raster_backward_tile<T, DT, GF>(b, tile, tiles_x, H, W, F, D, grad, face_idx, weights, img, feat, eps, g_img, g_feat);
}

// The fused operator's backward: the forward's tile kernel left the list of the tiles that hold a covered pixel (15 % of C4's
// tiles; tl::queue_items_reached, COV_SHARDS shards) -- a persistent grid walks it, workgroup w taking entries w, w + grid, ...
// The one-workgroup-per-tile launch above spent ~45 % of its wavefront time on the 26k workgroups whose only act is to
// find their tile's coverage byte clear (index arithmetic, two dependent loads, exit): 49 -> xx us at C4.
template <typename T, int DT, bool GF>
__global__ __launch_bounds__(256) void raster_backward_list_kernel(
    int B, int H, int W, int F, int D, const T* __restrict__ grad, const int64_t* __restrict__ face_idx,
    const T* __restrict__ weights, const T* __restrict__ img, const T* __restrict__ feat, float eps,
    T* __restrict__ g_img, T* __restrict__ g_feat, const unsigned int* __restrict__ cov_counts,
    const unsigned int* __restrict__ cov_list, unsigned int cov_cap, int grouped, const unsigned int* __restrict__ magic_word,
    unsigned int magic, unsigned int* __restrict__ bigwork) {
  __shared__ unsigned int s_end[tl::COV_SHARDS];  // inclusive prefix of the shards' entry counts
  const int tiles_x = (W + 15) / 16, ntiles = tiles_x * ((H + 15) / 16);
  const int lane = threadIdx.x & 63;
  // The soft mask's backward launch, which ran before this one, left the hot faces' terms in per-XCD copies of their records
  // (tile_lists.h, WORK_BIGHASH_WORD): fold them into the gradient -- six consecutive lanes = one face's record = one request --
  // and clear them (a second backward through a retained graph starts from zero again).  fp32 only; nothing to do without big faces.
  if constexpr (sizeof(T) == 4) {
    if (bigwork != nullptr && tl::big_hash_usable(bigwork[0])) {
      T* const side = reinterpret_cast<T*>(bigwork + (tl::WORK_BIGSIDE_WORD - tl::WORK_BIGHASH_WORD));
      for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < (unsigned int)tl::BIG_HASH_SLOTS * 6u; i += gridDim.x * 256u) {
        const unsigned int slot = i / 6u, c = i - slot * 6u;
        const unsigned int tag = bigwork[tl::BIG_HASH_TAGS_OFF + slot];
        if (tag == 0u) continue;
        T v = 0;
#pragma unroll
        for (int x = 0; x < tl::BIG_SIDE_COPIES; ++x) {
          T* p = side + ((size_t)x * tl::BIG_HASH_SLOTS + slot) * 8 + c;
          v += *p;
          *p = 0;
        }
        if (v != (T)0) kamd_atomic_add(g_img + (size_t)(tag - 1u) * 6 + c, v);
      }
    }
  }
  // The list is trusted only with the forward's signature in the header (tl::WORK_MAGIC_WORD: a work buffer of another
  // operator / build / shape would otherwise be read as tile indices); without it every tile is visited -- each wavefront
  // finds out from face_idx whether it has anything to do, as the one-workgroup-per-tile launch does.  (Uniform: a scalar load.)
  if (magic_word == nullptr || *magic_word != magic) {
    const unsigned int all = (unsigned int)B * (unsigned int)ntiles;
    for (unsigned int id = blockIdx.x; id < all; id += gridDim.x) {
      const int b = (int)(id / (unsigned int)ntiles), tile = (int)(id - (unsigned int)b * (unsigned int)ntiles);
      raster_backward_tile<T, DT, GF>(b, tile, tiles_x, H, W, F, D, grad, face_idx, weights, img, feat, eps, g_img, g_feat);
    }
    return;
  }
  if (threadIdx.x < 64) {
    static_assert(tl::COV_SHARDS <= 64, "one wavefront scans the shard counts");
    unsigned int c = lane < tl::COV_SHARDS ? min(cov_counts[lane * tl::COUNTER_STRIDE], cov_cap) : 0u;
    c = (unsigned int)wave_inclusive_scan((int)c);
    if (lane < tl::COV_SHARDS) s_end[lane] = c;
  }
  __syncthreads();
  // workgroup w runs on XCD w % 8 (round-robin dispatch) and takes the entries of group w % 8 (shards 4 (w % 8) .. + 3: with 8 or
  // more views, the views b % 8 == w % 8 -- a view's gradient lines stay in one XCD's L2), every (grid / 8)-th of them
  // (grouped == 0, an A/B knob: every workgroup strides over the whole list -- a view's lines then bounce between the XCDs' L2s)
  const unsigned int grp = blockIdx.x & 7u, first = !grouped ? 0u : (grp ? s_end[4 * grp - 1] : 0u);
  const unsigned int last = !grouped ? s_end[tl::COV_SHARDS - 1] : s_end[4 * grp + 3];
  const unsigned int step = grouped ? gridDim.x >> 3 : gridDim.x;
  for (unsigned int i = first + (grouped ? blockIdx.x >> 3 : blockIdx.x); i < last; i += step) {
    // the shard that holds entry i: the first whose inclusive prefix exceeds i (one ballot)
    const unsigned int e = s_end[lane & (tl::COV_SHARDS - 1)];
    const unsigned long long m = __ballot(lane < tl::COV_SHARDS && e > i);
    const int sh = __ffsll((long long)m) - 1;
    const unsigned int start = sh > 0 ? s_end[sh - 1] : 0u;
    const unsigned int id = cov_list[(size_t)sh * cov_cap + (i - start)];
    const int b = (int)(id / (unsigned int)ntiles), tile = (int)(id - (unsigned int)b * (unsigned int)ntiles);
    if (b >= B) continue;  // (never with a list the forward wrote: belt and braces behind the signature check)
    raster_backward_tile<T, DT, GF>(b, tile, tiles_x, H, W, F, D, grad, face_idx, weights, img, feat, eps, g_img, g_feat);
  }
}

template <typename T>
int rasterize_forward_launch(hipStream_t st, int B, int H, int W, int D, int64_t total_faces, const T* z, const T* img,
                             const T* bbox, const T* feat, const int64_t* first_idx, float multiplier, float eps,
                             T* interp, int64_t* sel_idx, T* weights, void* workspace) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  if (total_faces > 0 && workspace == nullptr) return (int)hipErrorInvalidValue;
  tl::BinIn<T> in{};
  in.B = B;
  in.F = 0;
  in.total_faces = total_faces;
  in.first = first_idx;
  in.img = img;      // already scaled by the caller (rasterization.py:320)
  in.z = z;
  in.lay = FaceLayout{3, 1, 1};
  in.bbox_r = bbox;  // given (rasterization.py:325-327)
  in.mult = (T)1;
  in.margin = (T)0;
  in.multiplier = multiplier;
  in.H = H;
  in.W = W;
  return raster2_bin_and_draw<T>(st, B, H, W, D, 0, (long long)total_faces, first_idx, in, feat, multiplier, eps, interp,
                                 sel_idx, weights, workspace);
}

// fused front door: raw (B,F,...) inputs + optional valid mask; scaling, bounding boxes and packing happen in the bin
// kernel; sel_idx comes out as the mesh-relative face index (what the Python layer returns)
template <typename T>
int rasterize_forward_fused_launch(hipStream_t st, int B, int H, int W, int F, int D, const T* z, FaceLayout lay, const T* img,
                                   const T* feat, const uint8_t* valid, const T* front, double multiplier, float eps,
                                   T* interp, int64_t* face_idx, T* weights, void* workspace) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const long long total_faces = (long long)B * F;
  if (total_faces > 0 && workspace == nullptr) return (int)hipErrorInvalidValue;
  tl::BinIn<T> in{};
  in.B = B;
  in.F = F;
  in.total_faces = total_faces;
  in.img = img;
  in.z = z;
  in.lay = lay;
  in.valid = valid;
  in.front = front;
  in.mult = (T)multiplier;
  in.margin = (T)0;
  in.multiplier = (float)multiplier;
  in.H = H;
  in.W = W;
  return raster2_bin_and_draw<T>(st, B, H, W, D, F, total_faces, (const int64_t*)nullptr, in, feat, (float)multiplier, eps,
                                 interp, face_idx, weights, workspace);
}

template <typename T>
int rasterize_backward_launch(hipStream_t st, int B, int H, int W, int F, int D, const T* grad, const int64_t* face_idx,
                              const T* weights, const T* img, const T* feat, float eps, T* g_img, T* g_feat,
                              const unsigned char* tile_cov = nullptr, const unsigned int* row_span = nullptr) {
  const long long total = (long long)B * H * W;
  if (total <= 0 || F <= 0) return 0;
  const dim3 grid((unsigned)(B * ((W + 15) / 16) * ((H + 15) / 16)));
#define KAMD_RB(DT)                                                                                                   \
  if (g_feat != nullptr)                                                                                              \
    KAMD_LAUNCH_TIMED(kamd::K_RASTER_BACKWARD, (raster_backward_kernel<T, DT, true>), grid, dim3(256), 0, st, B, H, W, F, D, grad, face_idx,  \
                      weights, img, feat, eps, g_img, g_feat, tile_cov, row_span);                                    \
  else                                                                                                                \
    KAMD_LAUNCH_TIMED(kamd::K_RASTER_BACKWARD, (raster_backward_kernel<T, DT, false>), grid, dim3(256), 0, st, B, H, W, F, D, grad, face_idx, \
                      weights, img, feat, eps, g_img, g_feat, tile_cov, row_span)
  switch (D) {
    case 1: KAMD_RB(1); break;
    case 2: KAMD_RB(2); break;
    case 3: KAMD_RB(3); break;
    case 4: KAMD_RB(4); break;
    default: KAMD_RB(0); break;
  }
#undef KAMD_RB
  return (int)hipGetLastError();
}

// the fused operator's backward over the forward's covered-tile list (persistent grid)
template <typename T>
int rasterize_backward_list_launch(hipStream_t st, int B, int H, int W, int F, int D, const T* grad, const int64_t* face_idx,
                                   const T* weights, const T* img, const T* feat, float eps, T* g_img, T* g_feat,
                                   const unsigned int* cov_counts, const unsigned int* cov_list, unsigned int cov_cap,
                                   const unsigned int* magic_word, unsigned int* bigwork) {
  const long long n_groups = (long long)B * ((W + 15) / 16) * ((H + 15) / 16);
  const unsigned int magic = tl::work_magic(B, H, W);
  if (n_groups <= 0 || F <= 0) return 0;
  static const int per_cu = kamd_env_int("KAMD_RBWD_PER_CU", 16);
  static const int grouped = kamd_env_int("KAMD_RBWD_GROUPED", 1) == 1 ? 1 : 0;  // (2: off, for A/B runs)
  const dim3 grid((unsigned)(((std::min<long long>(n_groups, (long long)KAMD_NUM_CU * per_cu) + 7) / 8) * 8));  // (a multiple of 8: every group served)
#define KAMD_RBL(DT)                                                                                                       \
  if (g_feat != nullptr)                                                                                                   \
    KAMD_LAUNCH_TIMED(kamd::K_RASTER_BACKWARD, (raster_backward_list_kernel<T, DT, true>), grid, dim3(256), 0, st, B, H, W, F, D, grad, face_idx,  \
                      weights, img, feat, eps, g_img, g_feat, cov_counts, cov_list, cov_cap, grouped, magic_word, magic, bigwork);  \
  else                                                                                                                     \
    KAMD_LAUNCH_TIMED(kamd::K_RASTER_BACKWARD, (raster_backward_list_kernel<T, DT, false>), grid, dim3(256), 0, st, B, H, W, F, D, grad, face_idx, \
                      weights, img, feat, eps, g_img, g_feat, cov_counts, cov_list, cov_cap, grouped, magic_word, magic, bigwork)
  switch (D) {
    case 1: KAMD_RBL(1); break;
    case 2: KAMD_RBL(2); break;
    case 3: KAMD_RBL(3); break;
    case 4: KAMD_RBL(4); break;
    default: KAMD_RBL(0); break;
  }
#undef KAMD_RBL
  return (int)hipGetLastError();
}

}  // namespace

namespace kamd {
template <typename T>
int raster_backward_draw_list(hipStream_t st, int B, int H, int W, int F, int D, const T* grad, const int64_t* face_idx, const T* weights,
                              const T* img, const T* feat, float eps, T* g_img, T* g_feat, const unsigned int* cov_counts,
                              const unsigned int* cov_list, unsigned int cov_cap, const unsigned int* magic_word, unsigned int* bigwork) {
  return rasterize_backward_list_launch<T>(st, B, H, W, F, D, grad, face_idx, weights, img, feat, eps, g_img, g_feat, cov_counts, cov_list, cov_cap,
                                           magic_word, bigwork);
}
template int raster_backward_draw_list<float>(hipStream_t, int, int, int, int, int, const float*, const int64_t*, const float*, const float*,
                                              const float*, float, float*, float*, const unsigned int*, const unsigned int*, unsigned int,
                                              const unsigned int*, unsigned int*);
template int raster_backward_draw_list<double>(hipStream_t, int, int, int, int, int, const double*, const int64_t*, const double*, const double*,
                                               const double*, float, double*, double*, const unsigned int*, const unsigned int*, unsigned int,
                                               const unsigned int*, unsigned int*);
template <typename T>
int raster2_draw(hipStream_t st, int B, int H, int W, int D, int F_dense, float multiplier, float eps, const T* rec,
                 const tl::Lists& LR, const T* feat, T* interp, int64_t* sel_idx, T* weights, const tl::ClassifyOut& co,
                 bool weights_internal) {
  if (!raster2_grid_fits(H, W)) return (int)hipErrorInvalidValue;
  const int wide_ok = raster2_wide_ok(W, interp, sel_idx, weights, co.soft_mask) | (weights_internal ? 2 : 0);
  KAMD_LAUNCH_TIMED(kamd::K_RASTER_TILE, (raster_tile_kernel2<T, true, false>), raster2_grid(LR, B), dim3(256), 0, st, KAMD_TILE_HEAD_ARGS(LR, co, F_dense), B,
                    (const int64_t*)nullptr, H, W, D, pixel_scale(multiplier, H, W), eps, wide_ok,
                    rec, LR, feat, interp, sel_idx, weights, co);
  return (int)hipGetLastError();
}
template <typename T>
int raster_backward_draw(hipStream_t st, int B, int H, int W, int F, int D, const T* grad, const int64_t* face_idx, const T* weights,
                         const T* img, const T* feat, float eps, T* g_img, T* g_feat, const unsigned char* tile_cov,
                         const unsigned int* row_span) {
  return rasterize_backward_launch<T>(st, B, H, W, F, D, grad, face_idx, weights, img, feat, eps, g_img, g_feat, tile_cov, row_span);
}
template int raster_backward_draw<float>(hipStream_t, int, int, int, int, int, const float*, const int64_t*, const float*,
                                         const float*, const float*, float, float*, float*, const unsigned char*, const unsigned int*);
template int raster_backward_draw<double>(hipStream_t, int, int, int, int, int, const double*, const int64_t*, const double*,
                                          const double*, const double*, float, double*, double*, const unsigned char*, const unsigned int*);
template int raster2_draw<float>(hipStream_t, int, int, int, int, int, float, float, const float*, const tl::Lists&, const float*,
                                 float*, int64_t*, float*, const tl::ClassifyOut&, bool);
template int raster2_draw<double>(hipStream_t, int, int, int, int, int, float, float, const double*, const tl::Lists&,
                                  const double*, double*, int64_t*, double*, const tl::ClassifyOut&, bool);
}  // namespace kamd

#ifdef KAMD_PHASE_PROF
extern "C" int kamd_debug_phase_cycles_rbwd(unsigned long long* out16, int reset) {
  int rc = 0;
  PHASE_READ(g_phase_rbwd, out16, reset, rc);
  return rc;
}
extern "C" int kamd_debug_tile_times(unsigned long long* out, int n_tiles) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_times), (size_t)(n_tiles < 65536 ? n_tiles : 65536) * 32);
}
extern "C" int kamd_debug_phase_cycles_raster(unsigned long long* out16, int reset) {
  int rc = 0;
  PHASE_READ(g_phase_raster, out16, reset, rc);
  return rc;
}
#endif

extern "C" {

size_t kamd_rasterize_forward_workspace(int B, int H, int W, int64_t total_faces, int elem_size) {
  if (B <= 0 || H <= 0 || W <= 0 || total_faces <= 0) return 0;
  return kamd::tl::make_layout(B, H, W, total_faces, elem_size, true, false).total;
}

int kamd_packed_rasterize_forward_f32(void* stream, int B, int H, int W, int D, int64_t total_faces, const float* z,
                                      const float* img, const float* bbox, const float* feat, const int64_t* first_idx,
                                      float multiplier, float eps, float* interp, int64_t* sel_idx, float* weights,
                                      void* workspace) {
  return rasterize_forward_launch<float>((hipStream_t)stream, B, H, W, D, total_faces, z, img, bbox, feat, first_idx,
                                         multiplier, eps, interp, sel_idx, weights, workspace);
}
int kamd_packed_rasterize_forward_f64(void* stream, int B, int H, int W, int D, int64_t total_faces, const double* z,
                                      const double* img, const double* bbox, const double* feat,
                                      const int64_t* first_idx, float multiplier, float eps, double* interp,
                                      int64_t* sel_idx, double* weights, void* workspace) {
  return rasterize_forward_launch<double>((hipStream_t)stream, B, H, W, D, total_faces, z, img, bbox, feat, first_idx,
                                          multiplier, eps, interp, sel_idx, weights, workspace);
}
int kamd_rasterize_forward_fused_f32(void* stream, int B, int H, int W, int F, int D, const float* z, const float* img,
                                     const float* feat, const uint8_t* valid, double multiplier, float eps,
                                     float* interp, int64_t* face_idx, float* weights, void* workspace) {
  return rasterize_forward_fused_launch<float>((hipStream_t)stream, B, H, W, F, D, z, FaceLayout{3, 1, 1}, img, feat, valid,
                                               (const float*)nullptr, multiplier, eps, interp, face_idx, weights, workspace);
}
int kamd_rasterize_forward_fused_f64(void* stream, int B, int H, int W, int F, int D, const double* z, const double* img,
                                     const double* feat, const uint8_t* valid, double multiplier, float eps,
                                     double* interp, int64_t* face_idx, double* weights, void* workspace) {
  return rasterize_forward_fused_launch<double>((hipStream_t)stream, B, H, W, F, D, z, FaceLayout{3, 1, 1}, img, feat, valid,
                                                (const double*)nullptr, multiplier, eps, interp, face_idx, weights, workspace);
}
// same, with z and the front-facing scalar read in place through element strides (used by kamd_dibr_rasterization_forward_*)
int kamd_rasterize_forward_fused_strided_f32(void* stream, int B, int H, int W, int F, int D, const float* z,
                                             int64_t z_face_stride, int64_t z_vertex_stride, const float* img,
                                             const float* feat, const uint8_t* valid, const float* front,
                                             int64_t front_stride, double multiplier, float eps, float* interp,
                                             int64_t* face_idx, float* weights, void* workspace) {
  return rasterize_forward_fused_launch<float>((hipStream_t)stream, B, H, W, F, D, z,
                                               FaceLayout{z_face_stride, z_vertex_stride, front_stride}, img, feat, valid, front,
                                               multiplier, eps, interp, face_idx, weights, workspace);
}
int kamd_rasterize_forward_fused_strided_f64(void* stream, int B, int H, int W, int F, int D, const double* z,
                                             int64_t z_face_stride, int64_t z_vertex_stride, const double* img,
                                             const double* feat, const uint8_t* valid, const double* front,
                                             int64_t front_stride, double multiplier, float eps, double* interp,
                                             int64_t* face_idx, double* weights, void* workspace) {
  return rasterize_forward_fused_launch<double>((hipStream_t)stream, B, H, W, F, D, z,
                                                FaceLayout{z_face_stride, z_vertex_stride, front_stride}, img, feat, valid,
                                                front, multiplier, eps, interp, face_idx, weights, workspace);
}
int kamd_rasterize_backward_f32(void* stream, int B, int H, int W, int F, int D, const float* grad,
                                const int64_t* face_idx, const float* weights, const float* img, const float* feat,
                                float eps, float* g_img, float* g_feat) {
  return rasterize_backward_launch<float>((hipStream_t)stream, B, H, W, F, D, grad, face_idx, weights, img, feat, eps,
                                          g_img, g_feat);
}
int kamd_rasterize_backward_f64(void* stream, int B, int H, int W, int F, int D, const double* grad,
                                const int64_t* face_idx, const double* weights, const double* img, const double* feat,
                                float eps, double* g_img, double* g_feat) {
  return rasterize_backward_launch<double>((hipStream_t)stream, B, H, W, F, D, grad, face_idx, weights, img, feat, eps,
                                           g_img, g_feat);
}

}  // extern "C"
