// Point -> triangle-soup distance forward / backward for MI355X (gfx950).
//
// Replaces kaolin/csrc/metrics/unbatched_triangle_distance_cuda.cu:237-317 (K7) and :319-416 (K8) behind the C
// ABI of include/kaolin_amd.h.  Semantics kept from the reference (restated in oracle/tridist_oracle.inc):
//   K7  per point, faces in ascending index: edge parameters uab/ubc/uca = dot(p - v, e) / dot(e, e); vertex
//       regions (type 1-3), edge regions (4-6, closed [0,1] parameter and "not above" the edge), otherwise the
//       plane projection (0); dist = |p - closest|^2 ROUNDED TO float (even for double inputs, :302); a face
//       wins only if strictly closer => lowest index on ties; face 0 seeds unconditionally (:303,:309).
//   K8  analytic gradient of the stored region: plane (:342-373), vertex (:374-397), edge (:176-233).
// Arithmetic: -ffp-contract=off and the contraction pattern PINNED with explicit fma() -- dot = fma(z,z', fma(y,y', x*x')),
// cross.x = fma(a.y, b.z, -(a.z*b.y)), point_at = fma(edge, t, vertex), plane: fma(-un, dist, p) -- what nvcc's default
// -fmad=true makes of the reference's expressions (same pin in oracle/tridist_oracle.inc; the unfused evaluation of rounds
// 1-2 failed the reference's own tolerance test six times as often: profiles/r03a_k7_contraction_ab.txt); point_at's
// parameter is a float (:172).  The reference's rsqrt() (:144) is evaluated as 1 / sqrt() (IEEE), the same pin as the oracle.
// The reference re-seeds its running best at every block of 512 faces (the forward launcher's BLOCK_SIZE for float and double,
// :424-433; :303,:310): a block whose FIRST face yields a NaN distance for a query is ignored as a whole for it.  The searches
// here take the minimum over all faces (NaN never wins, face 0 seeds); td_reseed_check_kernel / td_reseed_slow_kernel then redo
// the queries whose winner sits in such a block over the live blocks only -- the reference's answer on every input.
//
// MI355X design.  This is an all-pairs search (N x F closest-point evaluations, ~90 VALU each with three IEEE
// divides): VALU-bound by orders of magnitude, so the work is cut rather than the bytes:
//   * per-face invariants (edges, normal, edge normals, unit normal, edge lengths) are computed ONCE per face by
//     td_prep_kernel with the exact expressions the per-pair code would use (bit-identical results), leaving
//     ~45 VALU per evaluated pair;
//   * every face record carries a bounding sphere (centre, radius + rounding margin); a lane skips a face when
//     |p - c| - r exceeds sqrt(best) by more than a conservative margin (such a face cannot win, not even a
//     tie), and a wavefront skips it when all 64 lanes do: 11 VALU instead of the full evaluation;
//   * face records are staged through LDS in tiles and read with broadcast ds_read_b128;
//   * when N alone cannot fill 256 CUs the face range is split over blockIdx.y and merged in index order;
//   * from 65536 queries on, triangle_sweep.inc: faces and queries sorted along a Hilbert curve, a bounding sphere per tile of 64 faces,
//     and only the tiles that come closer than a query's best bound are staged and walked (identical results).
#include <string.h>
#include <hip/hip_runtime.h>
#include "common.h"
#include <stdlib.h>
#include "profile.h"
#include "grid_common.h"
#include "phase_prof.h"
#include "../../include/kaolin_amd.h"

namespace {

constexpr int TD_REC = 40;       // scalars per face record
constexpr int TD_TILE = 64;      // faces per LDS tile
constexpr int TD_THREADS = 256;

template <typename T> struct V3 { T x, y, z; };
template <typename T> __device__ __forceinline__ V3<T> mk(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> __device__ __forceinline__ V3<T> operator-(V3<T> a, V3<T> b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> __device__ __forceinline__ V3<T> operator+(V3<T> a, V3<T> b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> __device__ __forceinline__ V3<T> operator*(V3<T> a, T s) { return mk<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> __device__ __forceinline__ V3<T> operator/(V3<T> a, T s) { return mk<T>(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float td_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double td_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return td_fma(a.z, b.z, td_fma(a.y, b.y, a.x * b.x)); }
template <typename T> __device__ __forceinline__ V3<T> cross(V3<T> a, V3<T> b) {
  return mk<T>(td_fma(a.y, b.z, -(a.z * b.y)), td_fma(a.z, b.x, -(a.x * b.z)), td_fma(a.x, b.y, -(a.y * b.x)));
}
template <typename T> __device__ __forceinline__ V3<T> ld3(const T* p) { return mk<T>(p[0], p[1], p[2]); }
template <typename T> __device__ __forceinline__ void st3(T* p, V3<T> v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ float td_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double td_sqrt(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T td_abs(T x) { return x < 0 ? -x : x; }
template <typename T> __device__ __forceinline__ T max3abs(V3<T> v) { return fmax(td_abs(v.x), fmax(td_abs(v.y), td_abs(v.z))); }

// ---- per-face invariants ----------------------------------------------------------------------------
// blk_flag[b] != 0: face 512 b (the first of the reference's block b) may yield a NaN distance for an ordinary query;
// queue_count: the re-seed pass' append counter, zeroed here.
template <typename T>
__global__ __launch_bounds__(256) void td_prep_kernel(int F, const T* __restrict__ faces, T* __restrict__ rec,
                                                      float* __restrict__ centres, float* __restrict__ radius,
                                                      int* __restrict__ blk_flag, int* __restrict__ queue_count) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f == 0) *queue_count = 0;
  if (f >= F) return;
  const T* fv = faces + (size_t)f * 9;
  const V3<T> v1 = ld3(fv), v2 = ld3(fv + 3), v3 = ld3(fv + 6);
  const V3<T> e12 = v2 - v1, e23 = v3 - v2, e31 = v1 - v3;
  const V3<T> normal = cross(v1 - v2, e31);
  const T inv_len = (T)1 / td_sqrt(dot(normal, normal));
  T* r = rec + (size_t)f * TD_REC;
  st3(r + 0, v1);
  st3(r + 3, v2);
  st3(r + 6, v3);
  st3(r + 9, e12);
  st3(r + 12, e23);
  st3(r + 15, e31);
  st3(r + 18, normal);
  r[21] = dot(e12, e12);
  r[22] = dot(e23, e23);
  r[23] = dot(e31, e31);
  st3(r + 24, cross(normal, e12));
  st3(r + 27, cross(normal, e23));
  st3(r + 30, cross(normal, e31));
  st3(r + 33, normal * inv_len);
  // bounding sphere with rounding head-room: radius inflated, plus 1e-5 of the coordinate magnitude
  const V3<T> c = mk<T>((v1.x + v2.x + v3.x) / 3, (v1.y + v2.y + v3.y) / 3, (v1.z + v2.z + v3.z) / 3);
  const T r2 = fmax(dot(v1 - c, v1 - c), fmax(dot(v2 - c, v2 - c), dot(v3 - c, v3 - c)));
  const T rad = td_sqrt(r2) * (T)1.0001;
  st3(r + 36, c);
  r[39] = rad + (T)1e-5 * (max3abs(c) + rad);
  // Every bound the searches cull with (face sphere, face plane, tile sphere, tile slab) is a bound on the TRUE distance to the
  // triangle.  The reference's value is the computed one, and for a face without area it can be far below the true distance: the
  // cross product of two (nearly) parallel edges is the rounding error of its products, project_plane() then measures the distance
  // to a plane of arbitrary orientation through v1 (dist_type 0) -- a face with two equal vertices seen from afar may "win" with a
  // distance of 1e-4.  The reference's sequential loop takes such a face; so must we.  Three kinds of face whose normal is not
  // resolved (|n| <= 4e4 eps |a| |b|: relative direction error above ~1e-4, the head-room of the bounds):
  //  * the normal is EXACTLY zero (v1 == v2, v1 == v3, exactly cancelling products -- most duplicate-vertex faces of real
  //    meshes): the unit normal is 0 * inf = NaN, the plane case yields a NaN distance, which never wins; every other case
  //    measures the distance to a point ON the face, so the sphere bound stands: the face keeps its finite radius;
  //  * the normal is rounding noise with a finite unit vector g: whatever case is taken, the computed distance is at least
  //    |g . (p - c)| - radius (the plane case measures the distance to the plane through v1 with normal g, the others to points
  //    of the face): the record carries -radius, td_face_far() tests that slab instead of the sphere, and the sweep moves such
  //    faces behind the Hilbert order so that they do not blow up the tiles of their neighbours;
  //  * anything else (non-finite vertices, an underflowed normal): an infinite radius -- no test culls the face.
  const T nn = dot(normal, normal);
  bool finite_v = true;
#pragma unroll
  for (int k = 0; k < 9; ++k) finite_v = finite_v && td_abs(fv[k]) < (T)INFINITY;
  {
    const V3<T> a = v1 - v2;
    const T k = (T)4e4 * (sizeof(T) == 4 ? (T)1.1920929e-7 : (T)2.220446049250313e-16);
    const bool resolved = nn > k * k * dot(a, a) * dot(e31, e31);  // false for NaN
    if (!finite_v || !(r[39] < (T)INFINITY)) {
      r[39] = (T)INFINITY;
    } else if (!resolved) {
      const bool zero_normal = normal.x == 0 && normal.y == 0 && normal.z == 0;
      const T uu = dot(ld3(r + 33), ld3(r + 33));
      if (zero_normal) {
        // (finite radius kept)
      } else if (uu > (T)0.9999 && uu < (T)1.0001) {  // (|g| = 1 within the slab test's 0.1 % head-room -- ADVICE r05: 0.98..1.02 left a 1 % error in |g|)
        r[39] = -r[39];
      } else {
        r[39] = (T)INFINITY;
      }
    }
  }
  // The reference's block-first faces (512 b): can this one yield a NaN distance for a query of ordinary size?  Not if its
  // vertices are finite and of moderate size, its three edges have a non-zero squared length and its unit normal is finite (then
  // every quotient is finite or +-inf, never 0/0 or inf/inf, and no sum mixes infinities) -- td_reseed_check_kernel evaluates
  // the flagged ones, and every block-first face for a query that is itself huge or not finite.
  if ((f & 511) == 0) {
    const T big = sizeof(T) == 4 ? (T)1e9 : (T)1e75;
    bool clean = finite_v && r[21] > 0 && r[22] > 0 && r[23] > 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) clean = clean && td_abs(fv[k]) <= big;
#pragma unroll
    for (int k = 33; k < 36; ++k) clean = clean && td_abs(r[k]) < (T)INFINITY;
    blk_flag[f >> 9] = clean ? 0 : 1;
  }
  if (centres != nullptr) {  // float copies for the Morton sort / tile spheres (radius rounded up)
    centres[(size_t)f * 3 + 0] = (float)c.x;
    centres[(size_t)f * 3 + 1] = (float)c.y;
    centres[(size_t)f * 3 + 2] = (float)c.z;
    // (a face tested by its own slab says nothing about its tile: infinite, like the faces nothing culls)
    radius[f] = r[39] < 0 ? INFINITY : (float)r[39] * 1.000001f + 1e-30f;
  }
}

// can face record r be skipped for a query with pc = p - centre, d2c = |pc|^2 and reach = sqrt(best) with head-room?  (false for
// NaN / inf: then the face is evaluated).  PLANE: also test the distance to the face's plane (the sweep; the all-pairs kernel keeps
// its 11-instruction sphere test).  A negative radius marks a face whose normal is rounding noise (td_prep_kernel).
template <typename T, bool PLANE>
__device__ __forceinline__ bool td_face_far(const T* __restrict__ r, V3<T> pc, T d2c, T reach, int mode = 0) {
  const T rad = r[39];
#ifdef KAMD_TD_FAR_BRANCHY
  if (rad < 0) return td_abs(pc.x * r[33] + pc.y * r[34] + pc.z * r[35]) + rad > reach;
  const T rr = reach + rad;
  if (d2c > rr * rr) return true;
  return PLANE && !(mode & 2) && rad < (T)INFINITY && td_abs(pc.x * r[33] + pc.y * r[34] + pc.z * r[35]) > reach;
#else
  // (straight-line: this test runs ~1e8 times per call at C5, a branch on the face's kind costs more than the four selects)
  const bool slab_only = rad < 0;
  const T rr = reach + rad;
  const bool sphere_far = d2c > rr * rr && !slab_only;
  if (!PLANE) return sphere_far || (slab_only && td_abs(pc.x * r[33] + pc.y * r[34] + pc.z * r[35]) + rad > reach);
  const T h = td_abs(pc.x * r[33] + pc.y * r[34] + pc.z * r[35]);
  const bool plane_far = (slab_only || !(mode & 2)) && rad < (T)INFINITY && h > (slab_only ? reach - rad : reach);
  return sphere_far || plane_far;
#endif
}

// closest-point evaluation of one (point, face record) pair; returns the float-rounded squared distance
template <typename T>
__device__ __forceinline__ float td_eval(const T* __restrict__ r, V3<T> p, int* type_out) {
  const V3<T> v1 = ld3(r), v2 = ld3(r + 3), v3 = ld3(r + 6);
  const V3<T> e12 = ld3(r + 9), e31 = ld3(r + 15);
  const V3<T> pv1 = p - v1, pv3 = p - v3;
  const T uab = dot(pv1, e12) / r[21];
  const T uca = dot(pv3, e31) / r[23];
  V3<T> closest;
  int type;
  if (uca > 1 && uab < 0) {
    closest = v1;
    type = 1;
  } else {
    const V3<T> e23 = ld3(r + 12);
    const V3<T> pv2 = p - v2;
    const T ubc = dot(pv2, e23) / r[22];
    if (uab > 1 && ubc < 0) {
      closest = v2;
      type = 2;
    } else if (ubc > 1 && uca < 0) {
      closest = v3;
      type = 3;
    } else if ((uab <= 1 && uab >= 0) && dot(ld3(r + 24), pv1) <= 0) {
      const float t = (float)uab;
      closest = mk<T>(td_fma(e12.x, (T)t, v1.x), td_fma(e12.y, (T)t, v1.y), td_fma(e12.z, (T)t, v1.z));
      type = 4;
    } else if ((ubc <= 1 && ubc >= 0) && dot(ld3(r + 27), pv2) <= 0) {
      const float t = (float)ubc;
      closest = mk<T>(td_fma(e23.x, (T)t, v2.x), td_fma(e23.y, (T)t, v2.y), td_fma(e23.z, (T)t, v2.z));
      type = 5;
    } else if ((uca <= 1 && uca >= 0) && dot(ld3(r + 30), pv3) <= 0) {
      const float t = (float)uca;
      closest = mk<T>(td_fma(e31.x, (T)t, v3.x), td_fma(e31.y, (T)t, v3.y), td_fma(e31.z, (T)t, v3.z));
      type = 6;
    } else {
      const V3<T> un = ld3(r + 33);
      const T dist = dot(pv1, un);
      closest = mk<T>(td_fma(-un.x, dist, p.x), td_fma(-un.y, dist, p.y), td_fma(-un.z, dist, p.z));
      type = 0;
    }
  }
  const V3<T> dv = p - closest;
  *type_out = type;
  return (float)dot(dv, dv);
}

template <typename T>
__global__ __launch_bounds__(TD_THREADS) void td_main_kernel(
    int N, int F, int Fs, const T* __restrict__ points, const T* __restrict__ rec,
    T* __restrict__ out_dist, int* __restrict__ out_idx, int* __restrict__ out_type) {
  __shared__ __attribute__((aligned(16))) T tile[TD_TILE * TD_REC];
  const int s = blockIdx.y;
  const int i = blockIdx.x * TD_THREADS + threadIdx.x;
  const bool active = i < N;
  const V3<T> p = active ? ld3(points + (size_t)i * 3) : mk<T>(0, 0, 0);
  const T pmag = (T)1e-5 * max3abs(p);
  T best = INFINITY;
  T bound = INFINITY;  // sqrt(best) with head-room; a face whose sphere is farther than this cannot win
  int best_face = s * Fs, best_type = 0;
  const int f0 = s * Fs, f1 = min(F, f0 + Fs);
  for (int t0 = f0; t0 < f1; t0 += TD_TILE) {
    const int cnt = min(TD_TILE, f1 - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt * TD_REC; k += TD_THREADS) tile[k] = rec[(size_t)t0 * TD_REC + k];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const T* r = tile + k * TD_REC;
      const V3<T> pc = p - ld3(r + 36);
      const T d2c = dot(pc, pc);
      const bool skip = td_face_far<T, false>(r, pc, d2c, bound);  // false for NaN / inf: then the face is evaluated
      if (!__any(active && !skip)) continue;
      if (!active || skip) continue;
      int type;
      const float dist = td_eval<T>(r, p, &type);
      if ((t0 + k) == 0 || best > dist) {
        best = dist;
        best_type = type;
        best_face = t0 + k;
        bound = td_sqrt(best) * (T)1.001 + pmag;
      }
    }
  }
  if (active) {
    const size_t o = (size_t)s * N + i;
    out_dist[o] = best;
    out_idx[o] = best_face;
    out_type[o] = best_type;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void td_final_kernel(int N, int S, const T* __restrict__ part_d,
                                                       const int* __restrict__ part_i, const int* __restrict__ part_t,
                                                       T* __restrict__ dist, int64_t* __restrict__ idx,
                                                       int32_t* __restrict__ type) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  T best = part_d[i];
  int bi = part_i[i], bt = part_t[i];
  for (int s = 1; s < S; ++s) {
    const T d = part_d[(size_t)s * N + i];
    if (best > d) {  // strict: the lower split (lower face indices) keeps ties
      best = d;
      bi = part_i[(size_t)s * N + i];
      bt = part_t[(size_t)s * N + i];
    }
  }
  dist[i] = best;
  idx[i] = bi;
  type[i] = bt;
}

// ---- K8 ---------------------------------------------------------------------------------------------
template <typename T>
// (fa[9]: the lane's contribution to its face's nine gradient values; ia / ib: where vertex a / b sits in it)
__device__ __forceinline__ void td_edge_backward(V3<T> vab, V3<T> pb, T* fa, int ia, int ib, T* g_p, T grad) {
  const T l = dot(vab, pb);
  const T m = dot(vab, vab);
  const T k = l / m;
  const T j = fmax((T)0.0, fmin((T)1.0, k));
  const V3<T> i = (vab * j) - pb;
  const V3<T> i_bar = i * grad;
  const T j_bar = dot(i_bar, vab);
  const T dj_dk = (k > 0 && k < 1) ? 1 : 0;
  const T k_bar = j_bar * dj_dk;
  const T m_bar = k_bar * (-l / (m * m));
  const T l_bar = k_bar * (1 / m);
  const V3<T> di_dpb = mk<T>(-i_bar.x, -i_bar.y, -i_bar.z);
  const V3<T> pb_bar = vab * l_bar + di_dpb;
  const V3<T> dm_dvab = vab * (T)2.;
  const V3<T> vab_bar = ((dm_dvab * m_bar) + (pb * l_bar)) + (i_bar * j);
  const V3<T> vb_bar = mk<T>(-vab_bar.x - pb_bar.x, -vab_bar.y - pb_bar.y, -vab_bar.z - pb_bar.z);
  st3(g_p, pb_bar);
  fa[ia + 0] = vab_bar.x;
  fa[ia + 1] = vab_bar.y;
  fa[ia + 2] = vab_bar.z;
  fa[ib + 0] = vb_bar.x;
  fa[ib + 1] = vb_bar.y;
  fa[ib + 2] = vb_bar.z;
}

template <typename T>
__global__ __launch_bounds__(256) void td_backward_kernel(
    int N, const T* __restrict__ grad_dist, const T* __restrict__ points, const T* __restrict__ faces,
    const int64_t* __restrict__ face_idx, const int32_t* __restrict__ dist_type, T* __restrict__ g_points,
    T* __restrict__ g_faces) {
  // The gradient of a point's nearest face touches 3, 6 or all 9 of the face's values.  Added by the point's own lane value by
  // value, every atomic instruction touches a different line per lane -- nine REQUESTS per point, and global float atomics
  // cost per request (~60 ps chip-wide on MI355X, whatever the lanes in it): 0.42 ms at 1M points.  Here a wavefront's 64
  // contributions are staged in LDS rows of its own and leave as consecutive lanes: one request per point.
  __shared__ T s_val[4][64 * 9];
  __shared__ long long s_face[4][64];
  __shared__ unsigned int s_mask[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pi = blockIdx.x * 256 + threadIdx.x;
  T fa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned int touched = 0u;  // bit c: fa[c] is a contribution (a vertex the region's formula does not involve is not touched)
  long long f = -1;
  if (pi < N) {
    const int type = dist_type[pi];
    f = (long long)face_idx[pi];
    const V3<T> p = ld3(points + (size_t)pi * 3);
    const T* fv = faces + (size_t)f * 9;
    const V3<T> v1 = ld3(fv), v2 = ld3(fv + 3), v3 = ld3(fv + 6);
    const V3<T> e12 = v2 - v1, e23 = v3 - v2, e31 = v1 - v3;
    const T grad_out = (T)(2. * grad_dist[pi]);
    T* gp = g_points + (size_t)pi * 3;
    if (type == 0) {
      const V3<T> pv = p - v1;
      const V3<T> e21 = v1 - v2;
      const V3<T> normal = cross(e21, e31);
      const T len = td_sqrt(dot(normal, normal));
      const V3<T> un = normal / len;
      const T dist = dot(pv, un);
      const V3<T> gdv = un * (dist * grad_out);
      const T gd = dot(un, gdv);
      const V3<T> gpv = un * gd;
      const V3<T> gun = gdv * dist + pv * gd;
      const T glen = -dot(normal, gun) / (len * len);
      const T gdot2 = glen / (2 * td_sqrt(dot(normal, normal)));
      const V3<T> gn = (gun / len) + normal * (gdot2 * (T)2.);
      const V3<T> ge31 = cross(gn, e21);
      const V3<T> ge21 = cross(e31, gn);
      st3(gp, gpv);
      const V3<T> tmp = ge31 + ge21 - gpv;
      fa[0] = tmp.x;
      fa[1] = tmp.y;
      fa[2] = tmp.z;
      fa[3] = -ge21.x;
      fa[4] = -ge21.y;
      fa[5] = -ge21.z;
      fa[6] = -ge31.x;
      fa[7] = -ge31.y;
      fa[8] = -ge31.z;
      touched = 0x1FFu;
    } else if (type >= 1 && type <= 3) {
      const V3<T> v = type == 1 ? v1 : (type == 2 ? v2 : v3);
      const V3<T> g = (p - v) * grad_out;
      const int o = (type - 1) * 3;
#pragma unroll
      for (int c = 0; c < 9; ++c) fa[c] = c == o ? -g.x : (c == o + 1 ? -g.y : (c == o + 2 ? -g.z : fa[c]));
      touched = 7u << o;
      st3(gp, g);
    } else if (type == 4) {
      td_edge_backward<T>(e12, p - v1, fa, 3, 0, gp, grad_out);
      touched = 0x03Fu;
    } else if (type == 5) {
      td_edge_backward<T>(e23, p - v2, fa, 6, 3, gp, grad_out);
      touched = 0x1F8u;
    } else {
      td_edge_backward<T>(e31, p - v3, fa, 0, 6, gp, grad_out);
      touched = 0x1C7u;
    }
  }
  s_face[wave][lane] = f;
  s_mask[wave][lane] = touched;
#pragma unroll
  for (int c = 0; c < 9; ++c) s_val[wave][lane * 9 + c] = fa[c];
  // (every wavefront owns its rows: wavefront-level ordering is all that is needed)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int j = lane; j < 64 * 9; j += 64) {
    const int q = j / 9, c = j - q * 9;
    if (((s_mask[wave][q] >> c) & 1u) == 0u) continue;
    kamd_atomic_add(g_faces + (size_t)s_face[wave][q] * 9 + c, s_val[wave][j]);
  }
}


// ---- the reference's block re-seed ---------------------------------------------------------------------------------------
// unbatched_triangle_distance_cuda.cu:247-313: faces are walked in blocks of 512; inside a block the running best is seeded
// UNCONDITIONALLY by the block's first face (`sub_face_idx == 0 ||`, :303) and improved by `best_dist > dist`; blocks are merged
// with `start_face_idx == 0 || out_dist > best_dist` (:310).  A block whose first face yields NaN for a query is therefore
// ignored as a whole for it (block 0: the NaN sticks).  The searches above return the minimum over ALL faces with face 0's seed
// rule, which is the reference's answer unless the winner sits in such a dead block: td_reseed_check_kernel evaluates the first
// face of the winner's block (only where td_prep_kernel flagged it, or for a query that is itself huge / not finite: everything
// else cannot yield NaN) and queues the query; td_reseed_slow_kernel redoes a queued query over the live blocks, a workgroup each.
constexpr int TD_REF_BLOCK = 512;

struct TdReseed {
  int* blk_flag;     // ceil(F / 512)
  int* queue_count;  // 1 (zeroed by td_prep_kernel)
  int* queue;        // N
};
inline size_t td_reseed_bytes(int N, int F) {
  return (((size_t)kamd_cdiv(F, TD_REF_BLOCK) * 4 + 255) & ~(size_t)255) + 256 + (((size_t)N * 4 + 255) & ~(size_t)255);
}
inline TdReseed td_reseed_layout(char* base, int N, int F) {
  TdReseed r;
  r.blk_flag = (int*)base;
  base += ((size_t)kamd_cdiv(F, TD_REF_BLOCK) * 4 + 255) & ~(size_t)255;
  r.queue_count = (int*)base;
  r.queue = (int*)(base + 256);
  (void)N;
  return r;
}

template <typename T>
__global__ __launch_bounds__(256) void td_reseed_check_kernel(int N, const T* __restrict__ points, const T* __restrict__ rec,
                                                              const int* __restrict__ blk_flag, const int64_t* __restrict__ face_idx,
                                                              int* __restrict__ queue_count, int* __restrict__ queue) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool dead = false;
  if (i < N) {
    const int bf = (int)face_idx[i];
    if (bf >= TD_REF_BLOCK) {
      const int blk = bf / TD_REF_BLOCK;
      const V3<T> p = ld3(points + (size_t)i * 3);
      const T big = sizeof(T) == 4 ? (T)1e9 : (T)1e75;
      if (blk_flag[blk] != 0 || !(max3abs(p) <= big)) {
        int type;
        const float d = td_eval<T>(rec + (size_t)blk * TD_REF_BLOCK * TD_REC, p, &type);
        dead = d != d;
      }
    }
  }
  const unsigned long long m = __ballot(dead);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(queue_count, __popcll(m));
  base = __shfl(base, leader, 64);
  if (dead) queue[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
}

template <typename T>
__global__ __launch_bounds__(256) void td_reseed_slow_kernel(int F, const T* __restrict__ points, const T* __restrict__ rec,
                                                             const int* __restrict__ queue_count, const int* __restrict__ queue,
                                                             T* __restrict__ out_dist, int64_t* __restrict__ out_idx,
                                                             int32_t* __restrict__ out_type) {
  __shared__ float s_d[4];
  __shared__ int s_f[4], s_t[4];
  const int n = *queue_count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = blockIdx.x; q < n; q += gridDim.x) {
    const int i = queue[q];
    const V3<T> p = ld3(points + (size_t)i * 3);
    // block 0 is live (a NaN at face 0 sticks and such a query is never queued): every thread starts from face 0
    int type;
    float best = td_eval<T>(rec, p, &type);
    int best_f = 0;
    for (int b0 = 0; b0 < F; b0 += TD_REF_BLOCK) {
      if (b0 != 0) {
        int t0;
        const float d0 = td_eval<T>(rec + (size_t)b0 * TD_REC, p, &t0);  // (uniform over the workgroup)
        if (d0 != d0) continue;
      }
      const int b1 = min(F, b0 + TD_REF_BLOCK);
      for (int f = b0 + (int)threadIdx.x; f < b1; f += 256) {
        int t;
        const float d = td_eval<T>(rec + (size_t)f * TD_REC, p, &t);
        if (d < best) {  // ascending f per thread: the lowest index of a thread's ties stays; NaN never wins
          best = d;
          best_f = f;
          type = t;
        }
      }
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
      const float od = __shfl_xor(best, sh, 64);
      const int of = __shfl_xor(best_f, sh, 64), ot = __shfl_xor(type, sh, 64);
      if (od < best || (od == best && of < best_f)) {
        best = od;
        best_f = of;
        type = ot;
      }
    }
    __syncthreads();  // the previous query's reader is done
    if (lane == 0) {
      s_d[wave] = best;
      s_f[wave] = best_f;
      s_t[wave] = type;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w)
        if (s_d[w] < best || (s_d[w] == best && s_f[w] < best_f)) {
          best = s_d[w];
          best_f = s_f[w];
          type = s_t[w];
        }
      out_dist[i] = (T)best;
      out_idx[i] = best_f;
      out_type[i] = type;
    }
  }
}

template <typename T>
int td_reseed_launch(hipStream_t st, int N, int F, const T* points, const T* rec, const TdReseed& rs, T* dist, int64_t* face_idx,
                     int32_t* dist_type) {
  if (F <= TD_REF_BLOCK) return 0;  // a single block: nothing to re-seed
  kamd::ProfScope prof_(kamd::K_TD_FINAL, st);
  hipLaunchKernelGGL(td_reseed_check_kernel<T>, dim3(kamd_cdiv(N, 256)), dim3(256), 0, st, N, points, rec, (const int*)rs.blk_flag,
                     (const int64_t*)face_idx, rs.queue_count, rs.queue);
  hipLaunchKernelGGL(td_reseed_slow_kernel<T>, dim3(KAMD_NUM_CU * 4), dim3(256), 0, st, F, points, rec, (const int*)rs.queue_count,
                     (const int*)rs.queue, dist, face_idx, dist_type);
  return (int)hipGetLastError();
}

inline size_t td_align(size_t x) { return (x + 255) & ~(size_t)255; }

#include "triangle_sweep.inc"

inline bool td_sweep_applicable(int N, int F) {
  const char* e = getenv("KAMD_TRIANGLE_DISTANCE");
  if (e != nullptr) return e[0] == 's' && F >= 1;  // =sweep forces it, =brute keeps the all-pairs kernels (A/B timing, tests)
  // measured at F = 50k: 100k queries 4.3 ms vs 9.5 ms all-pairs, 1M queries 10.5 ms vs 56 ms; below ~64k queries the
  // all-pairs kernel with its face-range split fills the machine better than <= 256 sweep workgroups
  return F >= 2048 && N >= 65536;
}

struct TdPlan {
  int nx, S, Fs;
};
inline TdPlan td_plan(int N, int F) {
  TdPlan p;
  p.nx = kamd_cdiv(N, TD_THREADS);
  const int ntiles = kamd_cdiv(F, TD_TILE);
  int S = (KAMD_NUM_CU * 8 + p.nx - 1) / p.nx;  // aim at >= 8 workgroups per CU
  if (S > ntiles / 4) S = ntiles / 4;           // but keep >= 4 LDS tiles per split
  if (S < 1) S = 1;
  p.Fs = kamd_cdiv(ntiles, S) * TD_TILE;
  p.S = kamd_cdiv(F, p.Fs);
  return p;
}

template <typename T>
int td_forward_launch(hipStream_t st, int N, int F, const T* points, const T* faces, T* dist, int64_t* face_idx,
                      int32_t* dist_type, void* workspace) {
  if (N <= 0 || F <= 0) return 0;  // the caller's zero-initialised outputs stay (reference: the loops never run)
  if (workspace == nullptr) return (int)hipErrorInvalidValue;
  if (td_sweep_applicable(N, F)) return ts_forward_launch<T>(st, N, F, points, faces, dist, face_idx, dist_type, workspace);
  const TdPlan p = td_plan(N, F);
  T* rec = (T*)workspace;
  char* w = (char*)workspace + td_align((size_t)F * TD_REC * sizeof(T));
  const TdReseed rs = td_reseed_layout(w, N, F);
  w += td_reseed_bytes(N, F);
  T* part_d = (T*)w;
  w += td_align((size_t)p.S * N * sizeof(T));
  int* part_i = (int*)w;
  w += td_align((size_t)p.S * N * sizeof(int));
  int* part_t = (int*)w;
  {
    kamd::ProfScope prof_(kamd::K_TD_PREP, st);
    hipLaunchKernelGGL(td_prep_kernel<T>, dim3(kamd_cdiv(F, 256)), dim3(256), 0, st, F, faces, rec, (float*)nullptr,
                       (float*)nullptr, rs.blk_flag, rs.queue_count);
  }
  KAMD_CHECK(hipGetLastError());
  {
    kamd::ProfScope prof_(kamd::K_TD_MAIN, st);
    hipLaunchKernelGGL(td_main_kernel<T>, dim3(p.nx, p.S), dim3(TD_THREADS), 0, st, N, F, p.Fs, points, rec, part_d,
                     part_i, part_t);
  }
  KAMD_CHECK(hipGetLastError());
  {
    kamd::ProfScope prof_(kamd::K_TD_FINAL, st);
    hipLaunchKernelGGL(td_final_kernel<T>, dim3(kamd_cdiv(N, 256)), dim3(256), 0, st, N, p.S, part_d, part_i, part_t,
                     dist, face_idx, dist_type);
  }
  KAMD_CHECK(hipGetLastError());
  return td_reseed_launch<T>(st, N, F, points, (const T*)rec, rs, dist, face_idx, dist_type);
}

template <typename T>
int td_backward_launch(hipStream_t st, int N, int F, const T* grad, const T* points, const T* faces,
                       const int64_t* face_idx, const int32_t* dist_type, T* g_points, T* g_faces) {
  if (N <= 0 || F <= 0) return 0;
  {
    kamd::ProfScope prof_(kamd::K_TD_BACKWARD, st);
    hipLaunchKernelGGL(td_backward_kernel<T>, dim3(kamd_cdiv(N, 256)), dim3(256), 0, st, N, grad, points, faces,
                     face_idx, dist_type, g_points, g_faces);
  }
  KAMD_RETURN_LAST_ERROR();
}

}  // namespace

#ifdef KAMD_PHASE_PROF
extern "C" int kamd_debug_phase_cycles_ts(unsigned long long* out16, int reset) {
  int rc = 0;
  PHASE_READ(g_phase_ts, out16, reset, rc);
  return rc;
}
#endif

extern "C" {

size_t kamd_triangle_distance_forward_workspace(int N, int F, int elem_size) {
  if (N <= 0 || F <= 0) return 0;
  const TdPlan p = td_plan(N, F);
  const size_t brute = td_align((size_t)F * TD_REC * elem_size) + td_reseed_bytes(N, F) + td_align((size_t)p.S * N * elem_size) +
                       2 * td_align((size_t)p.S * N * sizeof(int));
  const size_t sweep = td_align((size_t)F * TD_REC * elem_size) + td_reseed_bytes(N, F) + ts_layout(nullptr, N, F, elem_size).total;
  return brute > sweep ? brute : sweep;  // either path may be taken (KAMD_TRIANGLE_DISTANCE)
}
int kamd_triangle_distance_work_counters(int on, unsigned long long* out8) {
  if (out8 != nullptr)
    for (int i = 0; i < 8; ++i) out8[i] = g_ts_stats_last[i];
  g_ts_stats_on = on != 0 ? 1 : 0;
  return 0;
}
int kamd_triangle_distance_forward_f32(void* stream, int N, int F, const float* points, const float* faces, float* dist,
                                       int64_t* face_idx, int32_t* dist_type, void* workspace) {
  return td_forward_launch<float>((hipStream_t)stream, N, F, points, faces, dist, face_idx, dist_type, workspace);
}
int kamd_triangle_distance_forward_f64(void* stream, int N, int F, const double* points, const double* faces,
                                       double* dist, int64_t* face_idx, int32_t* dist_type, void* workspace) {
  return td_forward_launch<double>((hipStream_t)stream, N, F, points, faces, dist, face_idx, dist_type, workspace);
}
int kamd_triangle_distance_backward_f32(void* stream, int N, int F, const float* grad, const float* points,
                                        const float* faces, const int64_t* face_idx, const int32_t* dist_type,
                                        float* g_points, float* g_faces) {
  return td_backward_launch<float>((hipStream_t)stream, N, F, grad, points, faces, face_idx, dist_type, g_points, g_faces);
}
int kamd_triangle_distance_backward_f64(void* stream, int N, int F, const double* grad, const double* points,
                                        const double* faces, const int64_t* face_idx, const int32_t* dist_type,
                                        double* g_points, double* g_faces) {
  return td_backward_launch<double>((hipStream_t)stream, N, F, grad, points, faces, face_idx, dist_type, g_points,
                                    g_faces);
}

}  // extern "C"
