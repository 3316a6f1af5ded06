// unbatched_mesh_to_spc: conservative voxelization of a triangle soup in [-1,1]^3 into a structured point cloud (SPC)
// octree.
//
// Replaces kaolin/csrc/ops/conversions/mesh_to_spc/mesh_to_spc_cuda.cu (:98-162 the 13-axis triangle / voxel test,
// :168-253 decide / subdivide / compactify, :239-297 barycentric weights, :299-467 the host loop) and
// kaolin/csrc/ops/spc/spc_cuda.cu:46-160 (morton_to_octree).
//
// The reference advances ONE octree level per round trip: decide -> scan -> read the count on the host -> allocate ->
// write 8 children per survivor (16 bytes each) -> next level, then sorts, and builds the octree with one more host
// read per level.  Here:
//   * eight lanes take a proposal and finish THREE levels of its subtree in registers (the eight children of a node
//     are tested at once, the survivors of each depth kept as bits of one word), so proposals are materialised every
//     third level only; each stage is a count
//     pass, a scan and an emit pass (the host reads one number per stage: output sizes are data-dependent);
//   * pairs leave a stage ordered by (triangle, Morton code); a stable radix sort on the 3*level key bits (the kernels
//     below, 8 bits a pass; the reference calls thrust) then makes "first of every run" the smallest triangle of every voxel;
//   * run heads, compaction, ALL octree levels and their sizes are produced on the device into one workspace; the host
//     reads the level sizes once, allocates the three results and a last kernel gathers them.
// Voxel tests use the reference's expressions in its operand order (float differences, double projections, comparison
// on float-rounded values); 1/sqrt stands for rsqrt (arithmetic contract).  Bit-exact vs
// oracle/mesh_to_spc_oracle.inc.
#include <string.h>
#include <hip/hip_runtime.h>
#include "common.h"
#include "profile.h"
#include "../../include/kaolin_amd.h"

namespace {

constexpr int MS_STAGE_LEVELS = 3;   // levels finished per stage below the proposal's own
constexpr int MS_MAX_LEVEL = 15;     // KAOLIN_SPC_MAX_LEVELS (spc_math.h:38)

// ---- Morton layout of spc_math.h:98-126: bit 3i = z_i, 3i+1 = y_i, 3i+2 = x_i ----------------------------------------
__device__ __forceinline__ uint64_t ms_spread3(uint64_t v) {  // 15 bits -> every third bit
  v &= 0x7FFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__device__ __forceinline__ uint64_t ms_to_morton(int x, int y, int z) {
  return (ms_spread3((uint64_t)x) << 2) | (ms_spread3((uint64_t)y) << 1) | ms_spread3((uint64_t)z);
}
__device__ __forceinline__ int ms_compact3(uint64_t v) {  // every third bit -> 15 bits
  v &= 0x1249249249249249ull;
  v = (v | (v >> 2)) & 0x10C30C30C30C30C3ull;
  v = (v | (v >> 4)) & 0x100F00F00F00F00Full;
  v = (v | (v >> 8)) & 0x1F0000FF0000FFull;
  v = (v | (v >> 16)) & 0x1F00000000FFFFull;
  v = (v | (v >> 32)) & 0x7FFFull;
  return (int)v;
}
__device__ __forceinline__ void ms_to_point(uint64_t m, int* x, int* y, int* z) {
  *x = ms_compact3(m >> 2);
  *y = ms_compact3(m >> 1);
  *z = ms_compact3(m);
}

struct MsTri {
  float a[3], b[3], c[3];
};
__device__ __forceinline__ MsTri ms_load_tri(const float* __restrict__ fv, int64_t t) {
  MsTri r;
  const float* p = fv + (size_t)t * 9;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    r.a[k] = p[k];
    r.b[k] = p[3 + k];
    r.c[k] = p[6 + k];
  }
  return r;
}
// voxel centre of grid position g at `level`: fmaf(g, size, half - 1) (mesh_to_spc_cuda.cu:186-195)
__device__ __forceinline__ void ms_centre(int x, int y, int z, int level, float* c, float* half) {
  const float two_level = (float)(1 << level);
  const float size = 2.0f / two_level;
  const float h = (float)(0.5 * size);
  c[0] = fmaf((float)x, size, h - 1.0f);
  c[1] = fmaf((float)y, size, h - 1.0f);
  c[2] = fmaf((float)z, size, h - 1.0f);
  *half = h;
}
__device__ __forceinline__ bool ms_sat(const double* v0, const double* v1, const double* v2, float half, double ax, double ay,
                                       double az) {
  const double d0 = v0[0] * ax + v0[1] * ay + v0[2] * az;
  const double d1 = v1[0] * ax + v1[1] * ay + v1[2] * az;
  const double d2 = v2[0] * ax + v2[1] * ay + v2[2] * az;
  const double maxd = fmax(d0, fmax(d1, d2)), mind = fmin(d0, fmin(d1, d2));
  const double r = half * (fabs(ax) + fabs(ay) + fabs(az));
  const float fd = (float)fmax(-maxd, mind);
  const float fr = (float)r;
  return fd <= fr;
}
__device__ __forceinline__ void ms_normalize(const double* a, const double* b, double* o) {  // normalize(b - a)
  const double x = b[0] - a[0], y = b[1] - a[1], z = b[2] - a[2];
  const double inv = 1.0 / sqrt(x * x + y * y + z * z);
  o[0] = inv * x;
  o[1] = inv * y;
  o[2] = inv * z;
}
// TriangleVoxelTest (mesh_to_spc_cuda.cu:119-162), same axis order, early exit
__device__ bool ms_triangle_voxel(const MsTri& t, const float* c, float half) {
  const double va[3] = {(double)(t.a[0] - c[0]), (double)(t.a[1] - c[1]), (double)(t.a[2] - c[2])};
  const double vb[3] = {(double)(t.b[0] - c[0]), (double)(t.b[1] - c[1]), (double)(t.b[2] - c[2])};
  const double vc[3] = {(double)(t.c[0] - c[0]), (double)(t.c[1] - c[1]), (double)(t.c[2] - c[2])};
  double ab[3], bc[3], ca[3];
  ms_normalize(va, vb, ab);
  ms_normalize(vb, vc, bc);
  ms_normalize(vc, va, ca);
  if (!ms_sat(va, vb, vc, half, 0.0, -ab[2], ab[1])) return false;
  if (!ms_sat(va, vb, vc, half, 0.0, -bc[2], bc[1])) return false;
  if (!ms_sat(va, vb, vc, half, 0.0, -ca[2], ca[1])) return false;
  if (!ms_sat(va, vb, vc, half, ab[2], 0.0, -ab[0])) return false;
  if (!ms_sat(va, vb, vc, half, bc[2], 0.0, -bc[0])) return false;
  if (!ms_sat(va, vb, vc, half, ca[2], 0.0, -ca[0])) return false;
  if (!ms_sat(va, vb, vc, half, -ab[1], ab[0], 0.0)) return false;
  if (!ms_sat(va, vb, vc, half, -bc[1], bc[0], 0.0)) return false;
  if (!ms_sat(va, vb, vc, half, -ca[1], ca[0], 0.0)) return false;
  if (!ms_sat(va, vb, vc, half, 1.0, 0.0, 0.0)) return false;
  if (!ms_sat(va, vb, vc, half, 0.0, 1.0, 0.0)) return false;
  if (!ms_sat(va, vb, vc, half, 0.0, 0.0, 1.0)) return false;
  return ms_sat(va, vb, vc, half, ab[1] * bc[2] - ab[2] * bc[1], ab[2] * bc[0] - ab[0] * bc[2], ab[0] * bc[1] - ab[1] * bc[0]);
}
__device__ __forceinline__ bool ms_test(const MsTri& t, int x, int y, int z, int level) {
  float c[3], half;
  ms_centre(x, y, z, level, c, &half);
  return ms_triangle_voxel(t, c, half);
}

// ---- a stage: every proposal (voxel at level_from, triangle) walks its subtree down to level_to -------------------------
// EMIT = false: counts[i] = number of voxels at level_to that pass with all their ancestors;
// EMIT = true : writes them (ascending Morton code) at offsets[i].  `tested` = the proposal's own level passed already.
// Eight lanes share a proposal: they test the eight children of the current node at once (one ballot), then the group
// descends into the survivors in child order, keeping the not-yet-visited children of each depth as 8 bits of one word.
// All groups of a wavefront run the same loop; a group that is done idles until the last one finishes.
template <bool EMIT>
__global__ __launch_bounds__(256) void ms_stage_kernel(int64_t n, const float* __restrict__ fv,
                                                       const int64_t* __restrict__ morton, const int64_t* __restrict__ tri,
                                                       int level_from, int level_to, int tested, int* __restrict__ counts,
                                                       const int64_t* __restrict__ offsets, int64_t* __restrict__ morton_out,
                                                       int64_t* __restrict__ tri_out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = gid >> 3;
  const int k = (int)(gid & 7), grp = (threadIdx.x & 63) >> 3;
  bool active = i < n;
  int64_t t = 0, o = 0;
  MsTri T = {};
  int cx = 0, cy = 0, cz = 0, count = 0, d = 0;
  unsigned rem = 0;  // 8 bits per depth: children still to visit
  bool expand = false;
  if (active) {
    t = tri[i];
    T = ms_load_tri(fv, t);
    ms_to_point((uint64_t)morton[i], &cx, &cy, &cz);
    if (EMIT) o = offsets[i];
    const bool self = tested || ms_test(T, cx, cy, cz, level_from);
    if (!self) {
      active = false;
    } else if (level_from == level_to) {
      if (EMIT && k == 0) {
        morton_out[o] = morton[i];
        tri_out[o] = t;
      }
      count = 1;
      active = false;
    } else {
      expand = true;
    }
  }
  while (__any(active)) {
    bool pass = false;
    int nx = 0, ny = 0, nz = 0;
    const int nl = level_from + d + 1;
    if (active && expand) {
      nx = 2 * cx + (k >> 2);
      ny = 2 * cy + ((k >> 1) & 1);
      nz = 2 * cz + (k & 1);
      pass = ms_test(T, nx, ny, nz, nl);
    }
    const unsigned long long bal = __ballot(pass);
    if (active) {
      if (expand) {
        unsigned mask = (unsigned)((bal >> (8 * grp)) & 0xFFull);
        if (nl == level_to) {
          if (EMIT && pass) {
            const int64_t q = o + count + __popc(mask & ((1u << k) - 1u));
            morton_out[q] = (int64_t)ms_to_morton(nx, ny, nz);
            tri_out[q] = t;
          }
          count += __popc(mask);
          mask = 0;
        }
        rem = (rem & ~(0xFFu << (8 * d))) | (mask << (8 * d));
        expand = false;
      }
      const unsigned todo = (rem >> (8 * d)) & 0xFFu;
      if (todo != 0u) {  // descend into the next surviving child
        const int c = __builtin_ctz(todo);
        rem &= ~(1u << (8 * d + c));
        cx = 2 * cx + (c >> 2);
        cy = 2 * cy + ((c >> 1) & 1);
        cz = 2 * cz + (c & 1);
        ++d;
        expand = true;
      } else if (d == 0) {
        active = false;
      } else {  // back to the parent
        --d;
        cx >>= 1;
        cy >>= 1;
        cz >>= 1;
      }
    }
  }
  if (!EMIT && i < n && k == 0) counts[i] = count;
}

// ---- exclusive scan of n ints into n + 1 int64 offsets (offsets[n] = total): sums of 1024-blocks, then apply ----------
__device__ __forceinline__ long long ms_block_inclusive(long long v, long long* s_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  long long woff = 0;
  for (int k = 0; k < wave; ++k) woff += s_wave[k];
  return woff + inc;
}
// n may live on the device (n_ptr != nullptr): entries at or beyond it count as 0
__global__ __launch_bounds__(1024) void ms_scan_sums_kernel(int64_t n_host, const int64_t* __restrict__ n_ptr,
                                                            const int* __restrict__ in, int64_t* __restrict__ sums) {
  __shared__ long long s_wave[16];
  const int64_t n = n_ptr ? *n_ptr : n_host;
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const long long tot = ms_block_inclusive(i < n ? in[i] : 0, s_wave);
  if (threadIdx.x == 1023) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void ms_scan_apply_kernel(int64_t n_host, const int64_t* __restrict__ n_ptr,
                                                             const int* __restrict__ in, const int64_t* __restrict__ sums,
                                                             int64_t* __restrict__ out) {
  __shared__ long long s_wave[16];
  __shared__ long long s_off;
  const int64_t n = n_ptr ? *n_ptr : n_host;
  long long part = 0;
  for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) part += sums[k];
  const long long before = ms_block_inclusive(part, s_wave);
  if (threadIdx.x == 1023) s_off = before;
  __syncthreads();
  const long long off = s_off;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const long long v = i < n ? in[i] : 0;
  const long long inc = ms_block_inclusive(v, s_wave);
  if (i < n) out[i] = off + inc - v;
  if (i == n - 1) out[n] = off + inc;
  if (n == 0 && i == 0) out[0] = 0;
}
int ms_scan(hipStream_t st, int64_t n_bound, const int64_t* n_ptr, const int* in, int64_t* out, int64_t* sums) {
  const unsigned nb = (unsigned)(n_bound > 0 ? (n_bound + 1023) / 1024 : 1);
  hipLaunchKernelGGL(ms_scan_sums_kernel, dim3(nb), dim3(1024), 0, st, n_bound, n_ptr, in, sums);
  hipLaunchKernelGGL(ms_scan_apply_kernel, dim3(nb), dim3(1024), 0, st, n_bound, n_ptr, in, sums, out);
  return (int)hipGetLastError();
}

// ---- stable LSD radix sort of (Morton code, triangle) pairs by code, 8 bits a pass ---------------------------------------
// The reference sorts with thrust (mesh_to_spc_cuda.cu:388-392); here: per pass a digit histogram per block of 2 048 pairs, ONE
// exclusive scan over the (digit, block) counts (ms_scan) and a scatter that ranks the pairs of a block in their original order
// (wavefront by wavefront: the lanes sharing a digit are found with eight ballots, lower lanes first; wavefronts and rounds of 256
// in order through LDS counters), so equal codes keep their order -- triangles ascending inside a voxel, which is what picks a
// voxel's face.  No library, nothing but launches on the stream.
constexpr int MS_SORT_ITEMS = 8, MS_SORT_BLOCK = 256 * MS_SORT_ITEMS;
__global__ __launch_bounds__(256) void ms_sort_hist_kernel(int64_t n, const uint64_t* __restrict__ keys, int shift, int nblk,
                                                           int* __restrict__ hist) {
  __shared__ int s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * MS_SORT_BLOCK;
#pragma unroll
  for (int r = 0; r < MS_SORT_ITEMS; ++r) {
    const int64_t e = base + r * 256 + threadIdx.x;
    if (e < n) atomicAdd(&s_h[(int)((keys[e] >> shift) & 255u)], 1);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblk + blockIdx.x] = s_h[threadIdx.x];
}
__global__ __launch_bounds__(256) void ms_sort_scatter_kernel(int64_t n, const uint64_t* __restrict__ keys,
                                                              const int64_t* __restrict__ vals, int shift, int nblk,
                                                              const int64_t* __restrict__ offs, uint64_t* __restrict__ keys_out,
                                                              int64_t* __restrict__ vals_out) {
  __shared__ long long s_run[256];
  __shared__ int s_wc[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  s_run[tid] = offs[(size_t)tid * nblk + blockIdx.x];
#pragma unroll
  for (int w = 0; w < 4; ++w) s_wc[w][tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * MS_SORT_BLOCK;
  for (int r = 0; r < MS_SORT_ITEMS; ++r) {
    const int64_t e = base + r * 256 + tid;
    const bool on = e < n;
    const uint64_t key = on ? keys[e] : 0ull;
    const int64_t val = on ? vals[e] : 0;
    const int digit = (int)((key >> shift) & 255u);
    // the lanes of this wavefront that hold the same digit (and a pair at all)
    unsigned long long peers = __ballot(on);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long m = __ballot((digit >> bit) & 1);
      peers &= ((digit >> bit) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (on && rank == 0) s_wc[wave][digit] = __popcll(peers);
    __syncthreads();
    if (on) {
      long long pos = s_run[digit] + rank;
      for (int w = 0; w < wave; ++w) pos += s_wc[w][digit];
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __syncthreads();
    s_run[tid] += (s_wc[0][tid] + s_wc[1][tid]) + (s_wc[2][tid] + s_wc[3][tid]);
#pragma unroll
    for (int w = 0; w < 4; ++w) s_wc[w][tid] = 0;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void ms_copy_words_kernel(int64_t n, const int64_t* __restrict__ a, int64_t* __restrict__ a_out,
                                                            const int64_t* __restrict__ b, int64_t* __restrict__ b_out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    a_out[i] = a[i];
    if (b != nullptr) b_out[i] = b[i];
  }
}
inline int ms_sort_blocks(int64_t n) { return (int)((n > 0 ? n : 1) + MS_SORT_BLOCK - 1) / MS_SORT_BLOCK; }

// ---- after the sort: one voxel per run of equal Morton codes, then the octree bottom-up --------------------------------
// sizes[0] = voxels, sizes[1 + l] = nodes of octree level l (root = level 0)
__global__ __launch_bounds__(256) void ms_heads_kernel(int64_t n_host, const int64_t* __restrict__ n_ptr,
                                                       const int64_t* __restrict__ m, int shift, int* __restrict__ flag) {
  const int64_t n = n_ptr ? *n_ptr : n_host;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || ((uint64_t)m[i - 1] >> shift) != ((uint64_t)m[i] >> shift)) ? 1 : 0;
}
__global__ __launch_bounds__(256) void ms_unique_kernel(int64_t n, const int64_t* __restrict__ m, const int64_t* __restrict__ t,
                                                        const int* __restrict__ flag, const int64_t* __restrict__ pos,
                                                        int64_t* __restrict__ um, int64_t* __restrict__ ut,
                                                        int64_t* __restrict__ sizes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && flag[i]) {
    um[pos[i]] = m[i];
    ut[pos[i]] = t[i];
  }
  if (i == 0) sizes[0] = pos[n];
}
// nodes of the level above: parent code + children bitmap of every run of siblings (spc_cuda.cu:64-90)
__global__ __launch_bounds__(256) void ms_parents_kernel(const int64_t* __restrict__ n_ptr, const int64_t* __restrict__ m,
                                                         const int* __restrict__ flag, const int64_t* __restrict__ pos,
                                                         int64_t* __restrict__ m_out, unsigned char* __restrict__ bytes,
                                                         int64_t* __restrict__ size_out) {
  const int64_t n = *n_ptr;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && flag[i]) {
    unsigned code = 0;
    int64_t j = i;
    do {
      code |= 1u << ((unsigned)m[j] & 7u);
      ++j;
    } while (j < n && !flag[j]);
    m_out[pos[i]] = (int64_t)((uint64_t)m[i] >> 3);
    bytes[pos[i]] = (unsigned char)code;
  }
  if (i == 0) *size_out = n > 0 ? pos[n] : 0;
}

// ---- float helpers of spc_math.h:366-457, left-to-right sums -----------------------------------------------------------
struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float project_edge(V3 v, V3 e, V3 p) { return vdot(p - v, e) / vdot(e, e); }
__device__ __forceinline__ bool not_above(V3 v, V3 e, V3 n, V3 p) { return vdot(vcross(n, e), p - v) <= 0; }
__device__ __forceinline__ V3 point_at(V3 v, V3 e, float t) { return V3{v.x + e.x * t, v.y + e.y * t, v.z + e.z * t}; }
__device__ V3 closest_point(V3 v1, V3 v2, V3 v3_, V3 p) {
  const V3 e12 = v2 - v1, e23 = v3_ - v2, e31 = v1 - v3_;
  const V3 n = vcross(v1 - v2, e31);
  const float uab = project_edge(v1, e12, p), uca = project_edge(v3_, e31, p);
  if (uca > 1 && uab < 0) return v1;
  const float ubc = project_edge(v2, e23, p);
  if (uab > 1 && ubc < 0) return v2;
  if (ubc > 1 && uca < 0) return v3_;
  if (uab <= 1. && uab >= 0. && not_above(v1, e12, n, p)) return point_at(v1, e12, uab);
  if (ubc <= 1. && ubc >= 0. && not_above(v2, e23, n, p)) return point_at(v2, e23, ubc);
  if (uca <= 1. && uca >= 0. && not_above(v3_, e31, n, p)) return point_at(v3_, e31, uca);
  const float inv = 1.0f / sqrtf(vdot(n, n));
  const V3 un = V3{n.x * inv, n.y * inv, n.z * inv};
  const float dist = (p.x - v1.x) * un.x + (p.y - v1.y) * un.y + (p.z - v1.z) * un.z;
  return V3{p.x - un.x * dist, p.y - un.y * dist, p.z - un.z * dist};
}

// results: face ids, barycentric weights of the closest point (mesh_to_spc_cuda.cu:239-297), octree bytes root first
__global__ __launch_bounds__(256) void ms_results_kernel(int64_t nvox, int level, const float* __restrict__ fv,
                                                         const int64_t* __restrict__ um, const int64_t* __restrict__ ut,
                                                         int64_t* __restrict__ face_ids, float* __restrict__ bary) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvox) return;
  const int64_t t = ut[i];
  face_ids[i] = t;
  int x, y, z;
  ms_to_point((uint64_t)um[i], &x, &y, &z);
  float c[3], half;
  ms_centre(x, y, z, level, c, &half);
  const float* f = fv + (size_t)t * 9;
  const V3 v1 = v3(f), v2 = v3(f + 3), v3_ = v3(f + 6), p = v3(c);
  const V3 cp = closest_point(v1, v2, v3_, p);
  const V3 nn = vcross(v1 - v2, v1 - v3_);
  const float delta = vdot(nn, nn);
  const V3 d1 = cp - v1, d2 = cp - v2, d3 = cp - v3_;
  const V3 ca = vcross(d2, d3), cb = vcross(d1, d3), cc = vcross(d1, d2);
  const float da = sqrtf(vdot(ca, ca)), db = sqrtf(vdot(cb, cb)), dc = sqrtf(vdot(cc, cc));
  const float rs = 1.0f / sqrtf(delta);
  float bx = da * rs, by = db * rs, bz = dc * rs;
  if (bx < 0.0f) bx = 0.f;
  if (by < 0.0f) by = 0.f;
  if (bz < 0.0f) bz = 0.f;
  const float k = (float)(1. / (double)(bx + by + bz));
  bary[i * 2] = bx * k;
  bary[i * 2 + 1] = by * k;
}
__global__ __launch_bounds__(256) void ms_gather_octree_kernel(int level, int64_t stride, const int64_t* __restrict__ sizes,
                                                               const unsigned char* __restrict__ level_bytes,
                                                               unsigned char* __restrict__ octree) {
  int64_t off = 0;
  for (int l = 0; l < level; ++l) {
    const int64_t nl = sizes[1 + l];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nl; i += (int64_t)gridDim.x * 256)
      octree[off + i] = level_bytes[(size_t)l * stride + i];
    off += nl;
  }
}

// workspace of kamd_mesh_to_spc_build for n pairs at `level` (8-byte words unless noted):
//   sizes (16) | sorted morton (n) | sorted tri (n) | unique morton A (n) | unique tri (n) | morton B (n) | pos (n + 1)
//   | scan sums (n / 1024 + 2) | flag (n ints) | level bytes (level * n)
//   | the sort's (digit, block) counts (256 * blocks ints), their offsets (256 * blocks + 1) and scan sums
struct MsWs {
  int64_t *sizes, *sm, *st, *um, *ut, *mb, *pos, *sums;
  int* flag;
  unsigned char* level_bytes;
  int* sort_hist;
  int64_t *sort_offs, *sort_sums;
  size_t total_bytes;
};
MsWs ms_ws(void* base, int64_t n, int level) {
  MsWs w;
  char* p = (char*)base;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += (bytes + 255) & ~(size_t)255;
    return r;
  };
  const size_t n8 = (size_t)(n > 0 ? n : 1) * 8;
  w.sizes = (int64_t*)take(16 * 8);
  w.sm = (int64_t*)take(n8);
  w.st = (int64_t*)take(n8);
  w.um = (int64_t*)take(n8);
  w.ut = (int64_t*)take(n8);
  w.mb = (int64_t*)take(n8);
  w.pos = (int64_t*)take(n8 + 8);
  w.sums = (int64_t*)take(((size_t)n / 1024 + 2) * 8);
  w.flag = (int*)take((size_t)(n > 0 ? n : 1) * 4);
  w.level_bytes = (unsigned char*)take((size_t)(level > 0 ? level : 1) * (size_t)(n > 0 ? n : 1));
  const size_t hn = (size_t)256 * (size_t)ms_sort_blocks(n);
  w.sort_hist = (int*)take(hn * 4);
  w.sort_offs = (int64_t*)take((hn + 1) * 8);
  w.sort_sums = (int64_t*)take((hn / 1024 + 2) * 8);
  w.total_bytes = (size_t)(p - (char*)base);
  return w;
}

}  // namespace

extern "C" {

int kamd_mesh_to_spc_stage_levels(void) { return MS_STAGE_LEVELS; }

size_t kamd_mesh_to_spc_scan_workspace(int64_t n) { return ((size_t)(n > 0 ? n : 0) / 1024 + 2) * 8; }

int kamd_mesh_to_spc_stage_count(void* stream, int64_t n, const float* face_vertices, const int64_t* morton,
                                 const int64_t* triangle_id, int level_from, int level_to, int tested, int32_t* counts,
                                 int64_t* offsets, void* scan_workspace) {
  if (n <= 0) return 0;
  if (level_from < 0 || level_to < level_from || level_to > MS_MAX_LEVEL) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  kamd::ProfScope prof_(kamd::K_SPC_STAGE, st);
  hipLaunchKernelGGL(ms_stage_kernel<false>, dim3(kamd_cdiv(n * 8, 256)), dim3(256), 0, st, n, face_vertices, morton, triangle_id,
                     level_from, level_to, tested, counts, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr);
  return ms_scan(st, n, nullptr, counts, offsets, (int64_t*)scan_workspace);
}

int kamd_mesh_to_spc_stage_emit(void* stream, int64_t n, const float* face_vertices, const int64_t* morton,
                                const int64_t* triangle_id, int level_from, int level_to, int tested, const int64_t* offsets,
                                int64_t* morton_out, int64_t* triangle_id_out) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  kamd::ProfScope prof_(kamd::K_SPC_STAGE, st);
  hipLaunchKernelGGL(ms_stage_kernel<true>, dim3(kamd_cdiv(n * 8, 256)), dim3(256), 0, st, n, face_vertices, morton, triangle_id,
                     level_from, level_to, tested, (int*)nullptr, offsets, morton_out, triangle_id_out);
  KAMD_RETURN_LAST_ERROR();
}

size_t kamd_mesh_to_spc_build_workspace(int64_t n, int level) { return ms_ws(nullptr, n, level).total_bytes; }

// pairs (any order inside a triangle, triangles ascending) -> sorted unique voxels + all octree levels, in the workspace;
// sizes (device, int64[1 + level]): voxels, then nodes per octree level root first
int kamd_mesh_to_spc_build(void* stream, int64_t n, int level, const int64_t* morton, const int64_t* triangle_id,
                           void* workspace, size_t workspace_bytes, int64_t* sizes) {
  if (n <= 0 || level < 0 || level > MS_MAX_LEVEL) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const MsWs w = ms_ws(workspace, n, level);
  if (workspace == nullptr || workspace_bytes < w.total_bytes) return (int)hipErrorInvalidValue;
  kamd::ProfScope prof_(kamd::K_SPC_BUILD, st);
  {
    // (level 0: a single voxel, every key is 0 and the order is already by triangle -- one pass over no bits would do; a copy)
    const int passes = (3 * level + 7) / 8;
    const int nblk = ms_sort_blocks(n);
    const uint64_t* src_k = (const uint64_t*)morton;
    const int64_t* src_v = triangle_id;
    if (passes == 0) {
      unsigned cg = (unsigned)kamd_cdiv(n, 256);
      if (cg > (unsigned)KAMD_NUM_CU * 8u) cg = (unsigned)KAMD_NUM_CU * 8u;
      hipLaunchKernelGGL(ms_copy_words_kernel, dim3(cg), dim3(256), 0, st, n, morton, w.sm, triangle_id, w.st);
    }
    for (int k = 0; k < passes; ++k) {  // the last pass lands in (sm, st); (um, ut) is the other side
      const bool to_sorted = ((passes - 1 - k) & 1) == 0;
      uint64_t* dst_k = (uint64_t*)(to_sorted ? w.sm : w.um);
      int64_t* dst_v = to_sorted ? w.st : w.ut;
      hipLaunchKernelGGL(ms_sort_hist_kernel, dim3(nblk), dim3(256), 0, st, n, src_k, 8 * k, nblk, w.sort_hist);
      KAMD_CHECK(ms_scan(st, (int64_t)256 * nblk, nullptr, w.sort_hist, w.sort_offs, w.sort_sums));
      hipLaunchKernelGGL(ms_sort_scatter_kernel, dim3(nblk), dim3(256), 0, st, n, src_k, src_v, 8 * k, nblk,
                         (const int64_t*)w.sort_offs, dst_k, dst_v);
      src_k = dst_k;
      src_v = dst_v;
    }
    KAMD_CHECK(hipGetLastError());
  }
  const unsigned g = (unsigned)kamd_cdiv(n, 256);
  hipLaunchKernelGGL(ms_heads_kernel, dim3(g), dim3(256), 0, st, n, (const int64_t*)nullptr, (const int64_t*)w.sm, 0, w.flag);
  KAMD_CHECK(ms_scan(st, n, nullptr, w.flag, w.pos, w.sums));
  hipLaunchKernelGGL(ms_unique_kernel, dim3(g), dim3(256), 0, st, n, (const int64_t*)w.sm, (const int64_t*)w.st,
                     (const int*)w.flag, (const int64_t*)w.pos, w.um, w.ut, w.sizes);
  // octree: level l nodes from the codes of level l + 1; counts stay on the device (grids sized by the bound n)
  int64_t* cur = w.um;
  int64_t* nxt = w.mb;
  for (int l = level; l > 0; --l) {
    const int64_t* n_ptr = (l == level) ? w.sizes : w.sizes + 1 + l;  // nodes of level l (voxels at the deepest)
    hipLaunchKernelGGL(ms_heads_kernel, dim3(g), dim3(256), 0, st, (int64_t)0, n_ptr, (const int64_t*)cur, 3, w.flag);
    KAMD_CHECK(ms_scan(st, n, n_ptr, w.flag, w.pos, w.sums));
    hipLaunchKernelGGL(ms_parents_kernel, dim3(g), dim3(256), 0, st, n_ptr, (const int64_t*)cur, (const int*)w.flag,
                       (const int64_t*)w.pos, nxt, w.level_bytes + (size_t)(l - 1) * (size_t)n, w.sizes + 1 + (l - 1));
    // the first swap must not overwrite the unique voxel codes (needed by the results kernel): A -> B -> sorted -> B ...
    int64_t* t = cur;
    cur = nxt;
    nxt = (t == w.um) ? w.sm : t;
  }
  hipLaunchKernelGGL(ms_copy_words_kernel, dim3(1), dim3(256), 0, st, (int64_t)(1 + level), (const int64_t*)w.sizes, sizes,
                     (const int64_t*)nullptr, (int64_t*)nullptr);
  KAMD_RETURN_LAST_ERROR();
}

int kamd_mesh_to_spc_results(void* stream, int64_t n, int level, const float* face_vertices, const void* workspace,
                             int64_t num_voxels, int64_t octree_bytes, uint8_t* octree, int64_t* face_ids,
                             float* barycoords) {
  if (n <= 0 || num_voxels <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const MsWs w = ms_ws(const_cast<void*>(workspace), n, level);
  kamd::ProfScope prof_(kamd::K_SPC_BUILD, st);
  hipLaunchKernelGGL(ms_results_kernel, dim3(kamd_cdiv(num_voxels, 256)), dim3(256), 0, st, num_voxels, level, face_vertices,
                     (const int64_t*)w.um, (const int64_t*)w.ut, face_ids, barycoords);
  if (octree_bytes > 0) {
    int blocks = kamd_cdiv(octree_bytes, 256);
    if (blocks > KAMD_NUM_CU * 8) blocks = KAMD_NUM_CU * 8;
    hipLaunchKernelGGL(ms_gather_octree_kernel, dim3(blocks), dim3(256), 0, st, level, n, (const int64_t*)w.sizes,
                       (const unsigned char*)w.level_bytes, octree);
  }
  KAMD_RETURN_LAST_ERROR();
}

}  // extern "C"
