// trianglemeshes_to_voxelgrids (dense) for MI355X (gfx950).
//
// The reference has NO native kernel for this op: it is pure PyTorch
// (kaolin/ops/conversions/trianglemesh.py:29-110, ops/mesh/trianglemesh.py:410-458,
// ops/conversions/pointcloud.py:42-75): repeat { keep triangles whose largest squared edge exceeds
// ((R-1)/R^2)^2; add the three edge midpoints to the vertex set; split into 4 } with a sort-based
// torch.unique and a host sync per round, then round(p*(R-1)), unique again, scatter into a dense grid.
// The result only depends on the SET {original vertices} U {all midpoints}, and marking a voxel is
// idempotent, so the whole op becomes one streaming pass with no sort, no de-duplication and no host sync:
//   vox_clear_extent_kernel : zero-fills the grid (16-byte stores; 67 MB at 256^3: the op's HBM floor) while its first
//                         workgroups reduce the per-mesh minimum / maximum into 32 partials (origin / scale of the
//                         normalisation when the caller gives none);
//   vox_mark_kernel     : the first B*V threads normalise one vertex each ((v - origin) / scale) and mark its voxel; the
//                         rest walk the faces, normalising their three vertices on the fly (same arithmetic, same values).
//                         The subdivision tree of a face is deterministic, so a thread is given
//                         (face, a base-4 path of L0 levels): it re-derives its sub-triangle by descending
//                         the path (the thread whose remaining path digits are all 0 marks the ancestors'
//                         midpoints, exactly once), then finishes the subtree depth-first, stackless, in registers.
//                         L0 is picked on the host from B*F alone so that >= 160k threads exist whatever the mesh (12
//                         huge faces or 10^6 tiny ones); more threads only repeat the shared path prefix.
// Arithmetic as the reference's torch ops, in the tensor's dtype: midpoint (a+b)/2, squared edge
// (dx*dx + dy*dy) + dz*dz (torch.sum's order for 3 elements), threshold rounded to the dtype,
// round-half-even of p*(R-1).  -ffp-contract=off.
#include "common.h"
#include "profile.h"
#include "../../include/kaolin_amd.h"

namespace {

// depth-first levels below L0.  Edges halve per level and subdivision stops below ~1/R, so depth d serves normalised
// edges up to 2^d / R: 20 levels = a triangle 4096x larger than a 256^3 grid's unit cube (reachable only through a
// caller-supplied scale that small; nearly all of such a triangle lies outside the grid).  Beyond the cap the subtree is
// cut where the reference would keep splitting -- documented in trianglemeshes_to_voxelgrids' docstring.
constexpr int VOX_MAXD = 20;
constexpr int VOX_MAXL0 = 10;

__host__ __device__ inline size_t vox_bit_words(int R) { return ((size_t)R * R * R + 31) / 32; }
template <typename T> __device__ __forceinline__ T vox_rint(T x);
template <> __device__ __forceinline__ float vox_rint<float>(float x) { return rintf(x); }
template <> __device__ __forceinline__ double vox_rint<double>(double x) { return rint(x); }

// BITS: `grid` is the mesh's BIT grid (one bit per voxel, 32 per word, voxel ((x R + y) R + z) = bit of that linear index) -- the
// sparse result (return_sparse=True) is compacted from it and never sees R^3 scalars (the reference builds its COO from the unique
// voxel indices, kaolin/ops/conversions/pointcloud.py:66-73; at R = 1024 a dense float grid is 4 GB, the bit grid 128 MB)
template <typename T, bool BITS>
__device__ __forceinline__ void vox_mark(T* __restrict__ grid, int R, T x, T y, T z) {
  const T s = (T)(R - 1);
  const T rx = vox_rint<T>(x * s), ry = vox_rint<T>(y * s), rz = vox_rint<T>(z * s);
  if (rx >= 0 && rx <= s && ry >= 0 && ry <= s && rz >= 0 && rz <= s) {
    const size_t lin = ((size_t)(int)rx * R + (int)ry) * R + (int)rz;
    if (BITS)
      atomicOr(reinterpret_cast<unsigned int*>(grid) + (lin >> 5), 1u << (unsigned int)(lin & 31));
    else
      grid[lin] = (T)1;
  }
}

template <typename T>
__device__ __forceinline__ T vox_edge2(const T* a, const T* b) {
  const T dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}
// t = {v1, v2, v3}; true when the triangle must be subdivided
template <typename T>
__device__ __forceinline__ bool vox_keep(const T* t, T thr) {
  T m = vox_edge2(t, t + 3);
  const T e2 = vox_edge2(t + 3, t + 6), e3 = vox_edge2(t + 6, t);
  if (e2 > m) m = e2;
  if (e3 > m) m = e3;
  return m > thr;
}
// mids = {v4 = (v1+v3)/2, v5 = (v1+v2)/2, v6 = (v2+v3)/2}
template <typename T>
__device__ __forceinline__ void vox_mids(const T* t, T* m) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    m[i] = (t[i] + t[6 + i]) / (T)2;
    m[3 + i] = (t[i] + t[3 + i]) / (T)2;
    m[6 + i] = (t[3 + i] + t[6 + i]) / (T)2;
  }
}
// children: 0 (v1,v4,v5)  1 (v2,v5,v6)  2 (v4,v5,v6)  3 (v3,v4,v6)
template <typename T>
__device__ __forceinline__ void vox_child(const T* t, const T* m, int c, T* out) {
  // value selects, not pointer selects: local arrays whose address is taken conditionally end up in scratch memory
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    out[i] = c == 0 ? t[i] : (c == 1 ? t[3 + i] : (c == 2 ? m[i] : t[6 + i]));
    out[3 + i] = (c == 1 || c == 2) ? m[3 + i] : m[i];
    out[6 + i] = c == 0 ? m[3 + i] : m[6 + i];
  }
}

// origin (B,3) / scale (B) of the reference's normalisation (trianglemesh.py:84-96) when the caller gives none:
// origin = per-mesh minimum, scale = largest extent above the origin.  torch.min / torch.max propagate NaN: so do these.
template <typename T>
__device__ __forceinline__ T vox_nan_min(T a, T b) { return (a != a || a < b) ? a : b; }
template <typename T>
__device__ __forceinline__ T vox_nan_max(T a, T b) { return (a != a || a > b) ? a : b; }
constexpr int VOX_NP = 32;  // partial extents per mesh
// scratch layout (scalars): [B*4] origin xyz + scale | [B*VOX_NP*6] partial min xyz, max xyz
template <typename T>
__global__ __launch_bounds__(256) void vox_clear_extent_kernel(int B, int V, const T* __restrict__ vertices, T* __restrict__ part,
                                                               uint4* __restrict__ grid16, size_t n16) {
  __shared__ T s_lo[3][4], s_hi[3][4];
  if ((int)blockIdx.x < B * VOX_NP && V > 0) {
    const int b = blockIdx.x / VOX_NP, slice = blockIdx.x % VOX_NP, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* vb = vertices + (size_t)b * V * 3;
    T lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = vb[a];  // vertex 0 is neutral for min and max alike
    for (int i = slice * 256 + threadIdx.x; i < V; i += VOX_NP * 256)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const T v = vb[(size_t)i * 3 + a];
        lo[a] = vox_nan_min<T>(lo[a], v);
        hi[a] = vox_nan_max<T>(hi[a], v);
      }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        lo[a] = vox_nan_min<T>(lo[a], __shfl_xor(lo[a], d, 64));
        hi[a] = vox_nan_max<T>(hi[a], __shfl_xor(hi[a], d, 64));
      }
      if (lane == 0) {
        s_lo[a][wave] = lo[a];
        s_hi[a][wave] = hi[a];
      }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      const int a = threadIdx.x;
      T l = s_lo[a][0], h = s_hi[a][0];
      for (int w = 1; w < 4; ++w) {
        l = vox_nan_min<T>(l, s_lo[a][w]);
        h = vox_nan_max<T>(h, s_hi[a][w]);
      }
      T* o = part + (size_t)blockIdx.x * 6;
      o[a] = l;
      o[3 + a] = h;
    }
  }
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) grid16[i] = z;
}
// origin / scale of mesh b from the partials (re-derived by whoever needs them: 32 x 6 scalars from L2)
template <typename T>
__device__ __forceinline__ void vox_norm_of(const T* __restrict__ part, const T* __restrict__ origin_in,
                                            const T* __restrict__ scale_in, int b, T* o, T* sc) {
  T s = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const T* p = part + (size_t)b * VOX_NP * 6;
    T l = p[a], h = p[3 + a];
    for (int k = 1; k < VOX_NP; ++k) {
      l = vox_nan_min<T>(l, p[k * 6 + a]);
      h = vox_nan_max<T>(h, p[k * 6 + 3 + a]);
    }
    o[a] = origin_in ? origin_in[b * 3 + a] : l;
    const T ext = h - o[a];
    s = a == 0 ? ext : vox_nan_max<T>(s, ext);
  }
  *sc = scale_in ? scale_in[b] : s;
}

// origin / scale of the meshes a workgroup touches: wave w < 2 reduces the partials of mesh b0 + w (lane k = partial k)
// into s_n[w]; a thread whose mesh is neither (tiny meshes, huge batches) derives its own
template <typename T>
__device__ __forceinline__ void vox_norm_cache(const T* __restrict__ part, const T* __restrict__ origin_in,
                                               const T* __restrict__ scale_in, int B, int b0, T (*s_n)[4]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 2 && b0 + wave < B) {
    const int b = b0 + wave;
    const T* p = part + ((size_t)b * VOX_NP + (lane < VOX_NP ? lane : 0)) * 6;
    T s = 0, o[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      T l = p[a], h = p[3 + a];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        l = vox_nan_min<T>(l, __shfl_xor(l, d, 64));
        h = vox_nan_max<T>(h, __shfl_xor(h, d, 64));
      }
      o[a] = origin_in ? origin_in[b * 3 + a] : l;
      const T ext = h - o[a];
      s = a == 0 ? ext : vox_nan_max<T>(s, ext);
    }
    if (lane == 0) {
      s_n[wave][0] = o[0];
      s_n[wave][1] = o[1];
      s_n[wave][2] = o[2];
      s_n[wave][3] = scale_in ? scale_in[b] : s;
    }
  }
  __syncthreads();
}

template <typename T, bool BITS>
__global__ __launch_bounds__(256) void vox_mark_kernel(long long nvert, long long total, int B, int V, int F, int R, int L0,
                                                       double thr_d, const T* __restrict__ vertices,
                                                       const int64_t* __restrict__ faces, const T* __restrict__ origin_in,
                                                       const T* __restrict__ scale_in, const T* __restrict__ part,
                                                       T* __restrict__ norm, T* __restrict__ grid_all) {
  __shared__ T s_n[2][4];
  const long long gid0 = (long long)blockIdx.x * 256, gid = gid0 + threadIdx.x;
  const unsigned int npath = 1u << (2 * L0);
  // mesh of the workgroup's first thread
  const int b0 = gid0 < nvert ? (int)(gid0 / V) : (int)((gid0 - nvert) / npath / F);
  vox_norm_cache<T>(part, origin_in, scale_in, B, b0, s_n);
  if (gid >= total) return;
  const bool is_vertex = gid < nvert;
  const long long bf = is_vertex ? 0 : (gid - nvert) / npath;
  const int b = is_vertex ? (int)(gid / V) : (int)(bf / F);
  T o[3], sc;
  if (b == b0 || b == b0 + 1) {
    o[0] = s_n[b - b0][0];
    o[1] = s_n[b - b0][1];
    o[2] = s_n[b - b0][2];
    sc = s_n[b - b0][3];
  } else {
    vox_norm_of<T>(part, origin_in, scale_in, b, o, &sc);
  }
  // (BITS: the meshes' bit grids are vox_bit_words(R) words apart)
  T* grid = BITS ? reinterpret_cast<T*>(reinterpret_cast<unsigned int*>(grid_all) + (size_t)b * vox_bit_words(R))
                 : grid_all + (size_t)b * R * R * R;
  const T* vb = vertices + (size_t)b * V * 3;
  if (is_vertex) {
    // one thread per vertex: (v - origin) / scale, two roundings as the reference's torch ops
    const int i = (int)(gid - (long long)b * V);
    if (i == 0) {
      norm[b * 4 + 0] = o[0];
      norm[b * 4 + 1] = o[1];
      norm[b * 4 + 2] = o[2];
      norm[b * 4 + 3] = sc;
    }
    vox_mark<T, BITS>(grid, R, (vb[(size_t)i * 3] - o[0]) / sc, (vb[(size_t)i * 3 + 1] - o[1]) / sc, (vb[(size_t)i * 3 + 2] - o[2]) / sc);
    return;
  }
  const unsigned int path = (unsigned int)((gid - nvert) % npath);
  const int f = (int)(bf % F);
  const T thr = (T)thr_d;
  // everything below stays in registers: a per-level stack indexed by the depth would live in scratch memory (it did:
  // 1.6 KB per thread, and the kernel ran at a fifth of its present speed)
  T cur[9], mid[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int64_t vi = faces[(size_t)f * 3 + k];
#pragma unroll
    for (int i = 0; i < 3; ++i) cur[k * 3 + i] = (vb[vi * 3 + i] - o[i]) / sc;
  }
  // descend the path prefix
  for (int l = 0; l < L0; ++l) {
    if (!vox_keep<T>(cur, thr)) return;
    vox_mids<T>(cur, mid);
    const int shift = 2 * (L0 - 1 - l);
    if ((path & ((1u << (shift + 2)) - 1u)) == 0u) {
      vox_mark<T, BITS>(grid, R, mid[0], mid[1], mid[2]);
      vox_mark<T, BITS>(grid, R, mid[3], mid[4], mid[5]);
      vox_mark<T, BITS>(grid, R, mid[6], mid[7], mid[8]);
    }
    T child[9];
    vox_child<T>(cur, mid, (int)((path >> shift) & 3u), child);
#pragma unroll
    for (int i = 0; i < 9; ++i) cur[i] = child[i];
  }
  // depth-first below L0 without a stack: a node is its digit string (2 bits per level) and its triangle is re-derived
  // from `cur` by walking that string (its ancestors are known to subdivide).  Trees below L0 are shallow by construction.
  int d = 0;
  unsigned long long digits = 0ull;  // digit of level l in bits [2l, 2l+2)
  while (true) {
    T t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = cur[i];
    for (int l = 0; l < d; ++l) {
      vox_mids<T>(t, mid);
      T child[9];
      vox_child<T>(t, mid, (int)((digits >> (2 * l)) & 3ull), child);
#pragma unroll
      for (int i = 0; i < 9; ++i) t[i] = child[i];
    }
    if (d < VOX_MAXD && vox_keep<T>(t, thr)) {
      vox_mids<T>(t, mid);
      vox_mark<T, BITS>(grid, R, mid[0], mid[1], mid[2]);
      vox_mark<T, BITS>(grid, R, mid[3], mid[4], mid[5]);
      vox_mark<T, BITS>(grid, R, mid[6], mid[7], mid[8]);
      digits &= ~(3ull << (2 * d));  // first child
      ++d;
      continue;
    }
    // leaf (or depth limit): next sibling, climbing while the node was a last child
    while (d > 0 && ((digits >> (2 * (d - 1))) & 3ull) == 3ull) --d;
    if (d == 0) break;
    digits += 1ull << (2 * (d - 1));
  }
}

template <typename T, bool BITS>
int vox_launch(hipStream_t st, int B, int V, int F, int R, const T* vertices, const int64_t* faces, const T* origin,
               const T* scale, T* scratch, T* grid) {
  if (B <= 0 || R <= 1) return 0;
  T* norm = scratch;
  T* part = scratch + (size_t)B * 4;
  const size_t bytes = BITS ? (size_t)B * vox_bit_words(R) * 4 : (size_t)B * R * R * R * sizeof(T);
  // torch's allocations are 512-byte aligned and bytes is a multiple of 16 for every R >= 2 with T = double; float grids of
  // odd R leave a tail of < 16 bytes to the generic fill
  const size_t n16 = ((uintptr_t)grid & 15) == 0 ? bytes / 16 : 0;
  {
    kamd::ProfScope prof_(kamd::K_VOX_VERTICES, st);
    size_t blocks = (n16 + 255) / 256;
    const size_t per_cu = (size_t)kamd_env_int("KAMD_VOX_FILL_PER_CU", 16);  // (measurement knob)
    if (blocks > (size_t)KAMD_NUM_CU * per_cu) blocks = (size_t)KAMD_NUM_CU * per_cu;
    if (V > 0 && blocks < (size_t)B * VOX_NP) blocks = (size_t)B * VOX_NP;
    if (blocks > 0)
      hipLaunchKernelGGL(vox_clear_extent_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, B, V, vertices, part,
                         (uint4*)grid, n16);
    if (n16 * 16 < bytes) KAMD_CHECK(kamd_zero_async((char*)grid + n16 * 16, bytes - n16 * 16, st));
    KAMD_CHECK(hipGetLastError());
  }
  if (V > 0) {
    int L0 = 0;
    // enough threads to fill the GPU, no more: every extra level repeats the shared path prefix in four times the threads
    // (profiles/r02_vox_threads.txt, 256^3: 50k faces 23.7 / 16.2 / 24.5 / 70 us with 50k / 200k / 800k / 3.2M threads; 80 faces
    // 88 / 29 / 25 / 51 us with 20k / 82k / 330k / 1.3M)
    const long long target = kamd_env_int("KAMD_VOX_THREADS", 160000);
    while (F > 0 && L0 < VOX_MAXL0 && (long long)B * F * (1ll << (2 * L0)) < target) ++L0;
    const long long nvert = (long long)B * V;
    const long long total = nvert + (long long)B * F * (1ll << (2 * L0));
    const double thr = ((double)(R - 1) / ((double)R * (double)R)) * ((double)(R - 1) / ((double)R * (double)R));
    {
      kamd::ProfScope prof_(kamd::K_VOX_FACES, st);
      hipLaunchKernelGGL((vox_mark_kernel<T, BITS>), dim3(kamd_cdiv(total, 256)), dim3(256), 0, st, nvert, total, B, V, F > 0 ? F : 1,
                         R, L0, thr, vertices, faces, origin, scale, (const T*)part, norm, grid);
    }
    KAMD_CHECK(hipGetLastError());
  }
  return 0;
}

}  // namespace

extern "C" {
size_t kamd_trianglemeshes_to_voxelgrids_workspace(int B, int V, int elem_size) {
  if (B <= 0) return 0;
  (void)V;
  return ((size_t)B * 4 + (size_t)B * VOX_NP * 6) * (size_t)elem_size;
}
int kamd_trianglemeshes_to_voxelgrids_f32(void* stream, int B, int V, int F, int R, const float* vertices,
                                          const int64_t* faces, const float* origin, const float* scale, float* norm,
                                          float* grid) {
  return vox_launch<float, false>((hipStream_t)stream, B, V, F, R, vertices, faces, origin, scale, norm, grid);
}
int kamd_trianglemeshes_to_voxelgrids_f64(void* stream, int B, int V, int F, int R, const double* vertices,
                                          const int64_t* faces, const double* origin, const double* scale, double* norm,
                                          double* grid) {
  return vox_launch<double, false>((hipStream_t)stream, B, V, F, R, vertices, faces, origin, scale, norm, grid);
}
// the same marks as ONE BIT per voxel: bits[b * words + (lin >> 5)] bit (lin & 31), words = kamd_trianglemeshes_to_voxelbits_words(R)
size_t kamd_trianglemeshes_to_voxelbits_words(int R) { return R > 1 ? vox_bit_words(R) : 0; }
int kamd_trianglemeshes_to_voxelbits_f32(void* stream, int B, int V, int F, int R, const float* vertices, const int64_t* faces,
                                         const float* origin, const float* scale, float* norm, uint32_t* bits) {
  return vox_launch<float, true>((hipStream_t)stream, B, V, F, R, vertices, faces, origin, scale, norm, reinterpret_cast<float*>(bits));
}
int kamd_trianglemeshes_to_voxelbits_f64(void* stream, int B, int V, int F, int R, const double* vertices, const int64_t* faces,
                                         const double* origin, const double* scale, double* norm, uint32_t* bits) {
  return vox_launch<double, true>((hipStream_t)stream, B, V, F, R, vertices, faces, origin, scale, norm, reinterpret_cast<double*>(bits));
}
}  // extern "C"
