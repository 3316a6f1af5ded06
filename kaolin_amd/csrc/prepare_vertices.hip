// prepare_vertices forward / backward for MI355X (gfx950) -- SURVEY.md section 8(f) row 2 ("next").
//
// The reference's prepare_vertices (kaolin/render/mesh/utils.py:128-175) is a chain of torch ops:
//   v_cam = (v - t) R^T  or  [v,1] M          (camera/legacy.py:22-38; utils.py:160-170)
//   v_img = (v_cam * proj)[..., :2] / (v_cam * proj)[..., 2:3]      (camera/legacy.py:123-140)
//   face_vertices_camera / _image = per-face gathers                (ops/mesh/mesh.py:54-75)
//   face_normals = unit((c1 - c0) x (c2 - c0)), |n| + 1e-10         (ops/mesh/trianglemesh.py:314-337)
// i.e. ~10 launches forward and, in training, an index_add scatter plus ~15 small launches backward -- around the
// four DIB-R kernels these cost as much as a DIB-R kernel each step.  Here:
//   pv_forward_kernel   one thread per (view, face): gathers the 3 vertices, transforms, projects, writes the three
//                       outputs (no intermediate per-vertex tensors);
//   pv_backward_kernel  one thread per (view, vertex): walks the vertex's incident (face, corner) list (CSR built once per
//                       `faces` tensor by the Python layer), sums the incoming gradients, applies the projection /
//                       normal Jacobians and rotates back: no atomics, deterministic summation order.
// Gradients are produced for the vertices only; if a camera tensor requires grad the Python layer keeps the torch path.
#include "common.h"
#include "profile.h"
#include "../../include/kaolin_amd.h"

namespace {

template <typename T> struct P3 { T x, y, z; };
template <typename T> __device__ __forceinline__ P3<T> p3(T x, T y, T z) { return P3<T>{x, y, z}; }
template <typename T> __device__ __forceinline__ P3<T> operator-(P3<T> a, P3<T> b) { return p3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> __device__ __forceinline__ P3<T> operator+(P3<T> a, P3<T> b) { return p3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> __device__ __forceinline__ P3<T> operator*(P3<T> a, T s) { return p3<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> __device__ __forceinline__ T dot3(P3<T> a, P3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ P3<T> cross3(P3<T> a, P3<T> b) {
  return p3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float pv_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double pv_sqrt(double x) { return sqrt(x); }

template <typename T>
struct Camera {   // one view
  T m[12];        // rot/trans mode: R (row-major 3x3) + t; transform mode: M (4x3 row-major)
  bool affine;    // true: cam_j = v . M[:,j] + M[3][j];  false: cam_j = (v - t) . R[j,:]
};
template <typename T>
__device__ __forceinline__ Camera<T> load_camera(int b, const T* rot, const T* trans, const T* transform) {
  Camera<T> c;
  if (transform != nullptr) {
    c.affine = true;
#pragma unroll
    for (int i = 0; i < 12; ++i) c.m[i] = transform[(size_t)b * 12 + i];
  } else {
    c.affine = false;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.m[i] = rot[(size_t)b * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.m[9 + i] = trans[(size_t)b * 3 + i];
  }
  return c;
}
template <typename T>
__device__ __forceinline__ P3<T> to_camera(const Camera<T>& c, P3<T> v) {
  if (c.affine)
    return p3<T>(v.x * c.m[0] + v.y * c.m[3] + v.z * c.m[6] + c.m[9], v.x * c.m[1] + v.y * c.m[4] + v.z * c.m[7] + c.m[10],
                 v.x * c.m[2] + v.y * c.m[5] + v.z * c.m[8] + c.m[11]);
  const P3<T> t = p3<T>(v.x - c.m[9], v.y - c.m[10], v.z - c.m[11]);
  return p3<T>(t.x * c.m[0] + t.y * c.m[1] + t.z * c.m[2], t.x * c.m[3] + t.y * c.m[4] + t.z * c.m[5],
               t.x * c.m[6] + t.y * c.m[7] + t.z * c.m[8]);
}
// gradient w.r.t. the world vertex of a gradient g on the camera-space point
template <typename T>
__device__ __forceinline__ P3<T> from_camera_grad(const Camera<T>& c, P3<T> g) {
  if (c.affine)
    return p3<T>(g.x * c.m[0] + g.y * c.m[1] + g.z * c.m[2], g.x * c.m[3] + g.y * c.m[4] + g.z * c.m[5],
                 g.x * c.m[6] + g.y * c.m[7] + g.z * c.m[8]);
  return p3<T>(g.x * c.m[0] + g.y * c.m[3] + g.z * c.m[6], g.x * c.m[1] + g.y * c.m[4] + g.z * c.m[7],
               g.x * c.m[2] + g.y * c.m[5] + g.z * c.m[8]);
}

template <typename T>
__global__ __launch_bounds__(256) void pv_forward_kernel(
    int B, int V, int F, const T* __restrict__ vertices, long long vstride, const int64_t* __restrict__ faces,
    const T* __restrict__ proj, const T* __restrict__ rot, const T* __restrict__ trans, const T* __restrict__ transform,
    T* __restrict__ fv_cam, T* __restrict__ fv_img, T* __restrict__ normals, int wide_ok) {
  // Every thread produces 18 scalars in three row-major outputs (9 + 6 + 3 per face): stored lane by lane, each store
  // instruction would scatter 4-byte pieces over a dozen cache lines.  A wavefront's 64 faces are contiguous in all three
  // outputs, so the scalars go through LDS and leave as 16-byte chunks that tile whole lines.
  __shared__ __attribute__((aligned(16))) T s_out[4][64 * 18];
  const long long total = (long long)B * F;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gid0 = gid - lane;  // first face of this wavefront
  const bool live = gid < total;
  const bool full = wide_ok && gid0 + 64 <= total;  // (wave-uniform; wide_ok: the three outputs are 16-byte aligned)
  T* sc = s_out[wave];                   // [64 * 9] camera | [64 * 6] image | [64 * 3] normals
  T* si = sc + 64 * 9;
  T* sn = si + 64 * 6;
  if (live) {
    const int b = (int)(gid / F), f = (int)(gid % F);
    const Camera<T> cam = load_camera<T>(b, rot, trans, transform);
    const T p0 = proj[0], p1 = proj[1], p2 = proj[2];
    P3<T> c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int64_t vi = faces[(size_t)f * 3 + k];
      const T* vp = vertices + (size_t)b * vstride + (size_t)vi * 3;
      c[k] = to_camera<T>(cam, p3<T>(vp[0], vp[1], vp[2]));
      const T px = c[k].x * p0, py = c[k].y * p1, pz = c[k].z * p2;
      const T ix = px / pz, iy = py / pz;
      if (full) {
        sc[lane * 9 + k * 3 + 0] = c[k].x;
        sc[lane * 9 + k * 3 + 1] = c[k].y;
        sc[lane * 9 + k * 3 + 2] = c[k].z;
        si[lane * 6 + k * 2 + 0] = ix;
        si[lane * 6 + k * 2 + 1] = iy;
      } else {
        T* oc = fv_cam + ((size_t)gid * 3 + k) * 3;
        oc[0] = c[k].x;
        oc[1] = c[k].y;
        oc[2] = c[k].z;
        T* oi = fv_img + ((size_t)gid * 3 + k) * 2;
        oi[0] = ix;
        oi[1] = iy;
      }
    }
    const P3<T> n = cross3<T>(c[1] - c[0], c[2] - c[0]);
    const T s = pv_sqrt(dot3<T>(n, n)) + (T)1e-10;
    const T nx = n.x / s, ny = n.y / s, nz = n.z / s;
    if (full) {
      sn[lane * 3 + 0] = nx;
      sn[lane * 3 + 1] = ny;
      sn[lane * 3 + 2] = nz;
    } else {
      T* on = normals + (size_t)gid * 3;
      on[0] = nx;
      on[1] = ny;
      on[2] = nz;
    }
  }
  if (!full) return;
  // (only this wavefront reads what it wrote: wavefront-level ordering is enough)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int PER16 = 16 / (int)sizeof(T);
  const uint4* lc = reinterpret_cast<const uint4*>(sc);
  const uint4* li = reinterpret_cast<const uint4*>(si);
  const uint4* ln = reinterpret_cast<const uint4*>(sn);
  uint4* gc = reinterpret_cast<uint4*>(fv_cam + (size_t)gid0 * 9);
  uint4* gi = reinterpret_cast<uint4*>(fv_img + (size_t)gid0 * 6);
  uint4* gn = reinterpret_cast<uint4*>(normals + (size_t)gid0 * 3);
  for (int i = lane; i < 64 * 9 / PER16; i += 64) gc[i] = lc[i];
  for (int i = lane; i < 64 * 6 / PER16; i += 64) gi[i] = li[i];
  for (int i = lane; i < 64 * 3 / PER16; i += 64) gn[i] = ln[i];
}

template <typename T>
__global__ __launch_bounds__(256) void pv_backward_kernel(
    int B, int V, int F, const T* __restrict__ vertices, long long vstride, const int64_t* __restrict__ faces,
    const T* __restrict__ proj, const T* __restrict__ rot, const T* __restrict__ trans, const T* __restrict__ transform,
    const int* __restrict__ adj_offsets, const int* __restrict__ adj_entries, const T* __restrict__ g_cam,
    const T* __restrict__ g_img, const T* __restrict__ g_nrm, T* __restrict__ g_vertices) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)B * V) return;
  const int b = (int)(gid / V), v = (int)(gid % V);
  const Camera<T> cam = load_camera<T>(b, rot, trans, transform);
  const T p0 = proj[0], p1 = proj[1], p2 = proj[2];
  const T* vb = vertices + (size_t)b * vstride;
  P3<T> gc = p3<T>(0, 0, 0);   // gradient on this vertex's camera-space position
  T gix = 0, giy = 0;          // gradient on its image-plane position
  const int e0 = adj_offsets[v], e1 = adj_offsets[v + 1];
  for (int e = e0; e < e1; ++e) {
    const int fk = adj_entries[e];  // face * 3 + corner
    const int f = fk / 3, k = fk % 3;
    const size_t bf = (size_t)b * F + f;
    if (g_cam != nullptr) {
      const T* g = g_cam + (bf * 3 + k) * 3;
      gc = gc + p3<T>(g[0], g[1], g[2]);
    }
    if (g_img != nullptr) {
      const T* g = g_img + (bf * 3 + k) * 2;
      gix += g[0];
      giy += g[1];
    }
    if (g_nrm != nullptr) {
      // u = n / (|n| + eps), n = (c1 - c0) x (c2 - c0): gradient on the face's three camera-space corners
      P3<T> c[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const T* vp = vb + (size_t)faces[(size_t)f * 3 + j] * 3;
        c[j] = to_camera<T>(cam, p3<T>(vp[0], vp[1], vp[2]));
      }
      const P3<T> a = c[1] - c[0], d = c[2] - c[0];
      const P3<T> n = cross3<T>(a, d);
      const T len = pv_sqrt(dot3<T>(n, n));
      const T s = len + (T)1e-10;
      const P3<T> gu = p3<T>(g_nrm[bf * 3], g_nrm[bf * 3 + 1], g_nrm[bf * 3 + 2]);
      // d(n/s)/dn = I/s - n n^T / (len s^2)
      const T ndg = dot3<T>(n, gu);
      const T coef = len > (T)0 ? ndg / (len * s * s) : (T)0;
      const P3<T> gn = p3<T>(gu.x / s - n.x * coef, gu.y / s - n.y * coef, gu.z / s - n.z * coef);
      const P3<T> ga = cross3<T>(d, gn);   // n = a x d  =>  dL/da = d x gn
      const P3<T> gd = cross3<T>(gn, a);   //                dL/dd = gn x a
      if (k == 0)
        gc = gc - (ga + gd);
      else if (k == 1)
        gc = gc + ga;
      else
        gc = gc + gd;
    }
  }
  if (g_img != nullptr) {
    const P3<T> c = to_camera<T>(cam, p3<T>(vb[(size_t)v * 3], vb[(size_t)v * 3 + 1], vb[(size_t)v * 3 + 2]));
    const T px = c.x * p0, py = c.y * p1, pz = c.z * p2;
    // img = (px/pz, py/pz)
    gc.x += gix * p0 / pz;
    gc.y += giy * p1 / pz;
    gc.z += -(gix * px + giy * py) * p2 / (pz * pz);
  }
  const P3<T> gv = from_camera_grad<T>(cam, gc);
  T* o = g_vertices + (size_t)gid * 3;
  o[0] = gv.x;
  o[1] = gv.y;
  o[2] = gv.z;
}

}  // namespace

extern "C" {
#define KAMD_PV_ENTRY(SFX, T)                                                                                          \
  int kamd_prepare_vertices_forward_##SFX(void* stream, int B, int V, int F, const T* vertices, int64_t vstride,       \
                                          const int64_t* faces, const T* proj, const T* rot, const T* trans,           \
                                          const T* transform, T* fv_cam, T* fv_img, T* normals) {                       \
    if ((long long)B * F <= 0) return 0;                                                                               \
    hipStream_t st = (hipStream_t)stream;                                                                              \
    {                                                                                                                  \
      kamd::ProfScope prof_(kamd::K_PV_FORWARD, st);                                                                   \
      hipLaunchKernelGGL(pv_forward_kernel<T>, dim3(kamd_cdiv((long long)B * F, 256)), dim3(256), 0, st, B, V, F,      \
                         vertices, (long long)vstride, faces, proj, rot, trans, transform, fv_cam, fv_img, normals,    \
                         ((((uintptr_t)fv_cam | (uintptr_t)fv_img | (uintptr_t)normals) & 15) == 0) ? 1 : 0);          \
    }                                                                                                                  \
    KAMD_RETURN_LAST_ERROR();                                                                                          \
  }                                                                                                                    \
  int kamd_prepare_vertices_backward_##SFX(void* stream, int B, int V, int F, const T* vertices, int64_t vstride,      \
                                           const int64_t* faces, const T* proj, const T* rot, const T* trans,          \
                                           const T* transform, const int32_t* adj_offsets,                             \
                                           const int32_t* adj_entries, const T* g_cam, const T* g_img,                 \
                                           const T* g_nrm, T* g_vertices) {                                            \
    if ((long long)B * V <= 0) return 0;                                                                               \
    hipStream_t st = (hipStream_t)stream;                                                                              \
    {                                                                                                                  \
      kamd::ProfScope prof_(kamd::K_PV_BACKWARD, st);                                                                  \
      hipLaunchKernelGGL(pv_backward_kernel<T>, dim3(kamd_cdiv((long long)B * V, 256)), dim3(256), 0, st, B, V, F,     \
                         vertices, (long long)vstride, faces, proj, rot, trans, transform, adj_offsets, adj_entries,   \
                         g_cam, g_img, g_nrm, g_vertices);                                                             \
    }                                                                                                                  \
    KAMD_RETURN_LAST_ERROR();                                                                                          \
  }
KAMD_PV_ENTRY(f32, float)
KAMD_PV_ENTRY(f64, double)
#undef KAMD_PV_ENTRY
}  // extern "C"
