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

#include "raster_backward.inc"

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
                              const unsigned char* tile_cov = nullptr, const unsigned int* row_centre = nullptr) {
  const long long total = (long long)B * H * W;
  if (total <= 0 || F <= 0) return 0;
  const dim3 grid((unsigned)(B * ((W + 15) / 16) * ((H + 15) / 16)));
  kamd::ProfScope prof_(kamd::K_RASTER_BACKWARD, st);
  const RasterBwdArgs<T> ra{B, H, W, F, D, grad, face_idx, weights, img, feat, eps, g_img, g_feat, tile_cov, row_centre};
#define KAMD_RB(DT)                                                                                  \
  if (g_feat != nullptr)                                                                             \
    hipLaunchKernelGGL((raster_backward_kernel<T, DT, true>), grid, dim3(256), 0, st, ra);           \
  else                                                                                               \
    hipLaunchKernelGGL((raster_backward_kernel<T, DT, false>), grid, dim3(256), 0, st, ra)
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

}  // namespace

namespace kamd {
template <typename T>
int raster2_draw(hipStream_t st, int B, int H, int W, int D, int F_dense, float multiplier, float eps, const T* rec,
                 const tl::Lists& LR, const T* feat, T* interp, int64_t* sel_idx, T* weights, const tl::ClassifyOut& co,
                 bool weights_internal) {
  kamd::ProfScope prof_(kamd::K_RASTER_TILE, st);
  hipLaunchKernelGGL((raster_tile_kernel2<T, true>), dim3(LR.ntiles * B), dim3(256), 0, st, B, F_dense,
                     (const int64_t*)nullptr, H, W, D, pixel_scale(multiplier, H, W), eps,
                     raster2_wide_ok(W, interp, sel_idx, weights, co.soft_mask) | (weights_internal ? 2 : 0),
                     kamd_env_int("KAMD_RASTER_MODE", 0), rec, LR, feat, interp, sel_idx, weights, co);
  return (int)hipGetLastError();
}
template <typename T>
int raster_backward_draw(hipStream_t st, int B, int H, int W, int F, int D, const T* grad, const int64_t* face_idx, const T* weights,
                         const T* img, const T* feat, float eps, T* g_img, T* g_feat, const unsigned char* tile_cov,
                         const unsigned int* row_centre) {
  return rasterize_backward_launch<T>(st, B, H, W, F, D, grad, face_idx, weights, img, feat, eps, g_img, g_feat, tile_cov, row_centre);
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
extern "C" int kamd_debug_phase_cycles_raster(unsigned long long* out16, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_raster), 16 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[16] = {0};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_raster), z, sizeof(z));
  }
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
