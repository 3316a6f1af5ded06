/*
 * kaolin_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A single-threaded (optionally OpenMP over the outermost independent loop) CPU
 * restatement of the reference's CUDA kernels on the DIB-R / 3D-metrics hot path.
 * It exists to CHECK libkaolin_amd.so; nothing in the product path (kaolin_amd/)
 * may import, link or call it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it.
 *
 * Every function states the reference file:line it restates and walks the data in
 * the reference's own order (thread -> loop iteration), including its tiling, its
 * seeds and its tie-breaks, so that integer outputs are defined by the same rules.
 *
 * Floating-point contract (shared with kaolin_amd/csrc, see DESIGN.md "Arithmetic"):
 *   compiled with -ffp-contract=off; expressions are evaluated operation by
 *   operation in the reference's source order, in the reference's operand types
 *   (including its float/double promotions), EXCEPT the two all-pairs distance
 *   sums, which are pinned to the fused form nvcc's default (-fmad=true) gives:
 *     sided_distance:  d = fma(dz,dz, fma(dy,dy, dx*dx))
 *   Half (fp16) follows c10::Half: float operation, round to half after each one.
 *
 * Pinning: the `-m "not gpu"` tests check this file against the reference's own
 * pure-PyTorch oracles and known-answer tests -- tests/test_sided_distance.py,
 * tests/test_dibr_oracle.py, tests/test_triangle_distance.py, tests/test_check_sign.py,
 * tests/test_deftet.py, tests/test_mesh_to_spc.py, tests/test_voxelgrid.py -- on
 * fixtures under tests/golden/, generated from /root/reference by
 * tests/golden/make_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* (bench.py's cpu_baseline: every hardware thread of the host, whatever the OpenMP runtime's default) */
ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- fp16 <-> fp32 (round to nearest even), software, no F16C needed ------- */
static uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t mant = x & 0x007fffffu;
  int32_t exp = (int32_t)((x >> 23) & 0xff);
  if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? (0x200u | (mant >> 13)) : 0));
  int32_t e = exp - 127 + 15;
  if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    mant |= 0x00800000u;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t half_m = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1))) half_m++;
    return (uint16_t)(sign | half_m);
  }
  uint32_t half_m = mant >> 13;
  uint32_t rem = mant & 0x1fffu;
  uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | half_m);
  if (rem > 0x1000u || (rem == 0x1000u && (half_m & 1))) h++;
  return h;
}
static float f16_bits_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t mant = h & 0x3ffu;
  uint32_t x;
  if (exp == 0) {
    if (mant == 0) {
      x = sign;
    } else {
      int e = -1;
      do {
        e++;
        mant <<= 1;
      } while (!(mant & 0x400u));
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3ffu) << 13);
    }
  } else if (exp == 0x1f) {
    x = sign | 0x7f800000u | (mant << 13);
  } else {
    x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}
static inline float hround(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

/* ========================================================================== */
/* K5  sided_distance_forward_cuda_kernel                                     */
/* reference: kaolin/csrc/metrics/sided_distance_cuda.cu:52-201               */
/*   tile of 512 targets (:57), per-thread best seeded by `k == 0 ||` at the  */
/*   start of every tile (:88,136,180), strict `<` inside a tile, merge across*/
/*   tiles with `k2 == 0 || result > best` (:193); outputs pre-zeroed by the  */
/*   wrapper (sided_distance.cpp:80-81) so M == 0 leaves zeros.               */
/* ========================================================================== */
#define SD_TILE 512

#define DEFINE_SIDED_FWD(NAME, T, FMA)                                                    \
  ORACLE_API void NAME(int B, int N, int M, const T* p1, const T* p2, T* dist,            \
                       int64_t* idx) {                                                    \
    _Pragma("omp parallel for collapse(2) schedule(static)")                              \
    for (int i = 0; i < B; ++i) {                                                         \
      for (int j = 0; j < N; ++j) {                                                       \
        const T x1 = p1[((size_t)i * N + j) * 3 + 0];                                     \
        const T y1 = p1[((size_t)i * N + j) * 3 + 1];                                     \
        const T z1 = p1[((size_t)i * N + j) * 3 + 2];                                     \
        T result = 0;                                                                     \
        int64_t result_i = 0;                                                             \
        for (int k2 = 0; k2 < M; k2 += SD_TILE) {                                         \
          const int end_k = (M < k2 + SD_TILE ? M : k2 + SD_TILE) - k2;                   \
          const T* buf = p2 + ((size_t)i * M + k2) * 3;                                   \
          int64_t best_i = 0;                                                             \
          T best = 0;                                                                     \
          for (int k = 0; k < end_k; ++k) {                                               \
            const T x2 = buf[k * 3 + 0] - x1;                                             \
            const T y2 = buf[k * 3 + 1] - y1;                                             \
            const T z2 = buf[k * 3 + 2] - z1;                                             \
            const T d = FMA(z2, z2, FMA(y2, y2, x2 * x2));                                \
            if (k == 0 || d < best) {                                                     \
              best = d;                                                                   \
              best_i = k + k2;                                                            \
            }                                                                             \
          }                                                                               \
          if (k2 == 0 || result > best) {                                                 \
            result = best;                                                                \
            result_i = best_i;                                                            \
          }                                                                               \
        }                                                                                 \
        if (M > 0) {                                                                      \
          dist[(size_t)i * N + j] = result;                                               \
          idx[(size_t)i * N + j] = result_i;                                              \
        }                                                                                 \
      }                                                                                   \
    }                                                                                     \
  }
DEFINE_SIDED_FWD(oracle_sided_distance_forward_f32, float, fmaf)
DEFINE_SIDED_FWD(oracle_sided_distance_forward_f64, double, fma)

/* at::Half instantiation: every operator rounds to half (c10/util/Half-inl.h). */
ORACLE_API void oracle_sided_distance_forward_f16(int B, int N, int M, const uint16_t* p1,
                                                  const uint16_t* p2, uint16_t* dist,
                                                  int64_t* idx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int i = 0; i < B; ++i) {
    for (int j = 0; j < N; ++j) {
      const float x1 = f16_bits_to_f32(p1[((size_t)i * N + j) * 3 + 0]);
      const float y1 = f16_bits_to_f32(p1[((size_t)i * N + j) * 3 + 1]);
      const float z1 = f16_bits_to_f32(p1[((size_t)i * N + j) * 3 + 2]);
      float result = 0;
      int64_t result_i = 0;
      for (int k2 = 0; k2 < M; k2 += SD_TILE) {
        const int end_k = (M < k2 + SD_TILE ? M : k2 + SD_TILE) - k2;
        const uint16_t* buf = p2 + ((size_t)i * M + k2) * 3;
        int64_t best_i = 0;
        float best = 0;
        for (int k = 0; k < end_k; ++k) {
          const float x2 = hround(f16_bits_to_f32(buf[k * 3 + 0]) - x1);
          const float y2 = hround(f16_bits_to_f32(buf[k * 3 + 1]) - y1);
          const float z2 = hround(f16_bits_to_f32(buf[k * 3 + 2]) - z1);
          const float d = hround(hround(hround(x2 * x2) + hround(y2 * y2)) + hround(z2 * z2));
          if (k == 0 || d < best) {
            best = d;
            best_i = k + k2;
          }
        }
        if (k2 == 0 || result > best) {
          result = best;
          result_i = best_i;
        }
      }
      if (M > 0) {
        dist[(size_t)i * N + j] = f32_to_f16_bits(result);
        idx[(size_t)i * N + j] = result_i;
      }
    }
  }
}

/* ========================================================================== */
/* K6  sided_distance_backward_cuda_kernel                                    */
/* reference: sided_distance_cuda.cu:203-242; g1 stored, g2 accumulated       */
/* (atomicAdd on the device: summation order is unspecified there; the oracle */
/* accumulates in ascending point order).                                     */
/* ========================================================================== */
#define DEFINE_SIDED_BWD(NAME, T)                                                         \
  ORACLE_API void NAME(int B, int N, int M, const T* grad, const T* p1, const T* p2,      \
                       const int64_t* idx, T* g1, T* g2) {                                \
    for (int b = 0; b < B; ++b) {                                                         \
      for (int pt = 0; pt < N; ++pt) {                                                    \
        const size_t main_id = (size_t)pt + (size_t)b * N;                                \
        const T x1 = p1[main_id * 3], y1 = p1[main_id * 3 + 1], z1 = p1[main_id * 3 + 2]; \
        const size_t p2_idx = ((size_t)idx[main_id] + (size_t)b * M) * 3;                 \
        const T x2 = p2[p2_idx], y2 = p2[p2_idx + 1], z2 = p2[p2_idx + 2];                \
        const T g = grad[main_id];                                                        \
        g1[main_id * 3] = 2 * (x1 - x2) * g;                                              \
        g1[main_id * 3 + 1] = 2 * (y1 - y2) * g;                                          \
        g1[main_id * 3 + 2] = 2 * (z1 - z2) * g;                                          \
        g2[p2_idx] += 2 * (x2 - x1) * g;                                                  \
        g2[p2_idx + 1] += 2 * (y2 - y1) * g;                                              \
        g2[p2_idx + 2] += 2 * (z2 - z1) * g;                                              \
      }                                                                                   \
    }                                                                                     \
  }
DEFINE_SIDED_BWD(oracle_sided_distance_backward_f32, float)
DEFINE_SIDED_BWD(oracle_sided_distance_backward_f64, double)

/* ---- DIB-R kernels K1-K4, instantiated for float and double ----------------- */
#define T float
#define FN(n) n##_f32
#define EXPFN expf
#include "dibr_oracle.inc"
#undef T
#undef FN
#undef EXPFN
#define T double
#define FN(n) n##_f64
#define EXPFN exp
#include "dibr_oracle.inc"
#undef T
#undef FN
#undef EXPFN

/* ---- triangle distance K7/K8, instantiated for float and double ----------------- */
#define T float
#define FN(n) n##_f32
#define SQRTFN sqrtf
#define FMAFN fmaf
#define REF_TILE 512  /* the FORWARD launcher instantiates BLOCK_SIZE = 512 for float AND double (unbatched_triangle_distance_cuda.cu:424-433); the 1024 / 512 of :37-38 is the backward's thread count */
#include "tridist_oracle.inc"
#undef T
#undef FN
#undef SQRTFN
#undef FMAFN
#undef REF_TILE
#define T double
#define FN(n) n##_f64
#define SQRTFN sqrt
#define FMAFN fma
#define REF_TILE 512
#include "tridist_oracle.inc"
#undef T
#undef FN
#undef SQRTFN
#undef FMAFN
#undef REF_TILE
/* The same restatement WITHOUT the contraction pin -- every multiply and add rounded on its own, exactly the source
 * expressions of unbatched_triangle_distance_cuda.cu:56-131 -- as oracle_triangle_distance_{forward,backward}_{f32,f64}_unfused.
 * The pinned build above is what the HIP kernel is bit-compared with; this one bounds how far that pin can sit from the
 * reference whichever contraction nvcc chose (tests/test_triangle_distance.py::test_gpu_within_ulps_of_both_contraction_variants). */
#define ORACLE_UNFUSED_FMA(a, b, c) ((a) * (b) + (c))
#define T float
#define FN(n) n##_f32_unfused
#define SQRTFN sqrtf
#define FMAFN ORACLE_UNFUSED_FMA
#define REF_TILE 512  /* the FORWARD launcher instantiates BLOCK_SIZE = 512 for float AND double (unbatched_triangle_distance_cuda.cu:424-433); the 1024 / 512 of :37-38 is the backward's thread count */
#include "tridist_oracle.inc"
#undef T
#undef FN
#undef SQRTFN
#undef FMAFN
#undef REF_TILE
#define T double
#define FN(n) n##_f64_unfused
#define SQRTFN sqrt
#define FMAFN ORACLE_UNFUSED_FMA
#define REF_TILE 512
#include "tridist_oracle.inc"
#undef T
#undef FN
#undef SQRTFN
#undef FMAFN
#undef REF_TILE

/* ---- mesh intersection (check_sign), instantiated for float and double ----------------- */
#define T float
#define FN(n) n##_f32
#include "meshint_oracle.inc"
#undef T
#undef FN
#define T double
#define FN(n) n##_f64
#include "meshint_oracle.inc"
#undef T
#undef FN

/* ---- deftet sparse render ---- */
#define T float
#define FN(n) n##_f32
#include "deftet_oracle.inc"
#undef T
#undef FN
#define T double
#define FN(n) n##_f64
#include "deftet_oracle.inc"
#undef T
#undef FN

/* ---- unbatched_mesh_to_spc (float only) ---- */
#include "mesh_to_spc_oracle.inc"
