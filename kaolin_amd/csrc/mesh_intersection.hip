// Ray-parity crossing count behind kaolin.ops.mesh.check_sign, for MI355X (gfx950) -- SURVEY.md 8(f) row 1.
//
// Replaces kaolin/csrc/ops/mesh/mesh_intersection_cuda.cu:101-218.  Semantics (restated in oracle/meshint_oracle.inc):
// per point q1, a ray to q2 = q1 + (10,0,0); a face is counted when q1's (y,z) lies in the face's (y,z) bounding box
// (limits rounded to float, :50-57), the signed volumes of q1 and q2 against the face differ in sign, q1's projection
// lies in the projected triangle (three direction-normalised signed areas, product tests >= 0) and -- if it falls exactly
// on an edge / vertex -- this face is the one designated to own the crossing.  result[j] = number of counted faces; all
// decisions are float comparisons, so every expression keeps the reference's operand order (-ffp-contract=off) and the
// count is integer-exact against the oracle.
//
// MI355X design: an all-pairs scan whose inner test is four compares against a per-face (y,z) box, so faces are staged
// through LDS as {p1,p2,p3, box} records (box computed once per face per tile, not per pair), a wavefront skips a face
// when none of its 64 points is in the box, and only the rare survivors run the volume / area cascade.  The face range
// is split over blockIdx.y when there are too few points to fill 256 CUs (counts are added atomically: exact integers).
#include "common.h"
#include "profile.h"
#include "../../include/kaolin_amd.h"

namespace {

constexpr int MI_TILE = 256;
constexpr int MI_THREADS = 256;

template <typename T>
__device__ __forceinline__ T mi_signed_volume(const T* a, const T* b, const T* c, const T* d) {
  const T bx = b[0] - a[0], by = b[1] - a[1], bz = b[2] - a[2];
  const T cx = c[0] - a[0], cy = c[1] - a[1], cz = c[2] - a[2];
  const T vx = by * cz - bz * cy, vy = bz * cx - bx * cz, vz = bx * cy - by * cx;
  const T dx = d[0] - a[0], dy = d[1] - a[1], dz = d[2] - a[2];
  return vx * dx + vy * dy + vz * dz;
}
// 2-D points are (y, z) pairs
template <typename T>
__device__ __forceinline__ T mi_signed_area(const T* a, const T* b, const T* c) {
  if (c[0] > b[0] || (b[0] == c[0] && c[1] < b[1]))
    return -((b[1] - c[1]) * (a[0] - c[0]) + (c[0] - b[0]) * (a[1] - c[1]));
  return (c[1] - b[1]) * (a[0] - b[0]) + (b[0] - c[0]) * (a[1] - b[1]);
}
template <typename T>
__device__ __forceinline__ bool mi_above(const T* v, const T* l, const T* r) {
  const T v1x = r[0] - l[0], v1y = r[1] - l[1], v2x = v[0] - l[0], v2y = v[1] - l[1];
  return (v1x * v2y - v1y * v2x) > 0.;
}

// does face (p1,p2,p3) own a crossing of the +x ray from q1?  (the box test has been done by the caller)
template <typename T>
__device__ __forceinline__ bool mi_counts(const T* q1, const T* p1, const T* p2, const T* p3) {
  const T q2[3] = {q1[0] + (T)10., q1[1], q1[2]};
  const bool cond_1 = mi_signed_volume<T>(q1, p1, p2, p3) > 0.;
  const bool cond_2 = mi_signed_volume<T>(q2, p1, p2, p3) > 0.;
  if (cond_1 == cond_2) return false;
  const T* q = q1 + 1;
  const T *a = p1 + 1, *b = p2 + 1, *c = p3 + 1;
  const T dist_1 = mi_signed_area<T>(q, a, b);
  const T dist_2 = mi_signed_area<T>(q, b, c);
  if (!(dist_1 * dist_2 >= 0)) return false;
  const T dist_3 = mi_signed_area<T>(q, c, a);
  if (!(dist_3 * dist_1 >= 0 && dist_2 * dist_3 >= 0)) return false;
  bool on_edge = false, on_vertex = false;
  T e1[2] = {0, 0}, e2[2] = {0, 0}, other[2] = {0, 0};
  auto set = [](T* dst, const T* src) {
    dst[0] = src[0];
    dst[1] = src[1];
  };
  if (q[0] == a[0] && q[1] == a[1]) {
    on_vertex = true; set(e1, b); set(e2, c);
  } else if (q[0] == b[0] && q[1] == b[1]) {
    on_vertex = true; set(e1, a); set(e2, c);
  } else if (q[0] == c[0] && q[1] == c[1]) {
    on_vertex = true; set(e1, a); set(e2, b);
  } else if (dist_1 == 0.) {
    on_edge = true; set(e1, a); set(e2, b); set(other, c);
  } else if (dist_2 == 0.) {
    on_edge = true; set(e1, b); set(e2, c); set(other, a);
  } else if (dist_3 == 0.) {
    on_edge = true; set(e1, c); set(e2, a); set(other, b);
  }
  if (e1[0] > e2[0] || (e1[0] == e2[0] && e1[1] > e2[1])) {
    const T t0 = e1[0], t1 = e1[1];
    set(e1, e2);
    e2[0] = t0;
    e2[1] = t1;
  }
  if (on_edge && mi_above<T>(other, e1, e2)) return false;
  if (on_vertex && !(mi_above<T>(q, e1, e2) && (e1[0] < q[0]) && (e2[0] >= q[0]))) return false;
  return true;
}

template <typename T>
__global__ __launch_bounds__(MI_THREADS) void mesh_intersection_kernel(
    int N, int F, int Fs, const T* __restrict__ points, const T* __restrict__ v1s, const T* __restrict__ v2s,
    const T* __restrict__ v3s, T* __restrict__ result, int use_atomic) {
  __shared__ T s_p[MI_TILE * 9];
  __shared__ __attribute__((aligned(16))) float s_box[MI_TILE * 4];  // y_min, y_max, z_min, z_max (float, as the reference)
  const int j = blockIdx.x * MI_THREADS + threadIdx.x;
  const bool live = j < N;
  T q1[3] = {0, 0, 0};
  if (live) {
    q1[0] = points[(size_t)j * 3];
    q1[1] = points[(size_t)j * 3 + 1];
    q1[2] = points[(size_t)j * 3 + 2];
  }
  int count = 0;
  const int f0 = blockIdx.y * Fs, f1 = min(F, f0 + Fs);
  for (int k2 = f0; k2 < f1; k2 += MI_TILE) {
    const int n = min(MI_TILE, f1 - k2);
    __syncthreads();
    for (int i = threadIdx.x; i < n * 3; i += MI_THREADS) {
      const int k = i / 3, c = i % 3;
      s_p[k * 9 + c] = v1s[(size_t)(k2 + k) * 3 + c];
      s_p[k * 9 + 3 + c] = v2s[(size_t)(k2 + k) * 3 + c];
      s_p[k * 9 + 6 + c] = v3s[(size_t)(k2 + k) * 3 + c];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += MI_THREADS) {
      const T* p = s_p + k * 9;
      s_box[k * 4 + 0] = (float)fmin(p[1], fmin(p[4], p[7]));
      s_box[k * 4 + 1] = (float)fmax(p[1], fmax(p[4], p[7]));
      s_box[k * 4 + 2] = (float)fmin(p[2], fmin(p[5], p[8]));
      s_box[k * 4 + 3] = (float)fmax(p[2], fmax(p[5], p[8]));
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      const float y_min = s_box[k * 4], y_max = s_box[k * 4 + 1], z_min = s_box[k * 4 + 2], z_max = s_box[k * 4 + 3];
      const bool outside = q1[1] < y_min || y_max < q1[1] || q1[2] < z_min || z_max < q1[2];
      if (!__any(live && !outside)) continue;
      if (!live || outside) continue;
      const T* p = s_p + k * 9;
      if (mi_counts<T>(q1, p, p + 3, p + 6)) ++count;
    }
  }
  if (live) {
    if (use_atomic)
      kamd_atomic_add(result + j, (T)count);
    else
      result[j] = (T)count;
  }
}

template <typename T>
int mesh_intersection_launch(hipStream_t st, int N, int F, const T* points, const T* v1, const T* v2, const T* v3, T* result) {
  if (N <= 0) return 0;
  const int nx = kamd_cdiv(N, MI_THREADS);
  const int ntiles = kamd_cdiv(F > 0 ? F : 1, MI_TILE);
  int S = (KAMD_NUM_CU * 8 + nx - 1) / nx;  // aim at >= 8 workgroups per CU
  if (S > ntiles / 2) S = ntiles / 2;
  if (S < 1) S = 1;
  const int Fs = kamd_cdiv(ntiles, S) * MI_TILE;
  S = kamd_cdiv(F > 0 ? F : 1, Fs);
  if (S > 1) KAMD_CHECK(kamd_zero_async(result, (size_t)N * sizeof(T), st));
  {
    kamd::ProfScope prof_(kamd::K_MESH_INTERSECTION, st);
    hipLaunchKernelGGL(mesh_intersection_kernel<T>, dim3(nx, S), dim3(MI_THREADS), 0, st, N, F, Fs, points, v1, v2, v3,
                       result, S > 1 ? 1 : 0);
  }
  KAMD_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {
int kamd_mesh_intersection_f32(void* stream, int N, int F, const float* points, const float* v1, const float* v2,
                               const float* v3, float* result) {
  return mesh_intersection_launch<float>((hipStream_t)stream, N, F, points, v1, v2, v3, result);
}
int kamd_mesh_intersection_f64(void* stream, int N, int F, const double* points, const double* v1, const double* v2,
                               const double* v3, double* result) {
  return mesh_intersection_launch<double>((hipStream_t)stream, N, F, points, v1, v2, v3, result);
}
}  // extern "C"
