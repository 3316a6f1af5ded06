// TEMPORARY: entry points not implemented yet return hipErrorNotSupported (801).
#include "common.h"
#include "../../include/kaolin_amd.h"
extern "C" {
size_t kamd_rasterize_forward_workspace(int, int, int, int64_t) { return 0; }
size_t kamd_dibr_soft_mask_forward_workspace(int, int, int, int) { return 0; }
size_t kamd_triangle_distance_forward_workspace(int, int, int) { return 0; }
int kamd_packed_rasterize_forward_f32(void* stream, int B, int H, int W, int D,
                                      int64_t total_faces,
                                      const float* z, const float* img, const float* bbox,
                                      const float* feat, const int64_t* first_idx,
                                      float multiplier, float eps,
                                      float* interp, int64_t* sel_idx, float* weights,
                                      void* workspace) { return 801; }
int kamd_packed_rasterize_forward_f64(void* stream, int B, int H, int W, int D,
                                      int64_t total_faces,
                                      const double* z, const double* img, const double* bbox,
                                      const double* feat, const int64_t* first_idx,
                                      float multiplier, float eps,
                                      double* interp, int64_t* sel_idx, double* weights,
                                      void* workspace) { return 801; }
int kamd_rasterize_backward_f32(void* stream, int B, int H, int W, int F, int D,
                                const float* grad, const int64_t* face_idx,
                                const float* weights, const float* img, const float* feat,
                                float eps, float* g_img, float* g_feat) { return 801; }
int kamd_rasterize_backward_f64(void* stream, int B, int H, int W, int F, int D,
                                const double* grad, const int64_t* face_idx,
                                const double* weights, const double* img, const double* feat,
                                float eps, double* g_img, double* g_feat) { return 801; }
int kamd_dibr_soft_mask_forward_f32(void* stream, int B, int H, int W, int F, int K,
                                    const float* img, const float* large_bbox,
                                    const int64_t* sel_idx, float sigmainv, float multiplier,
                                    float* soft_mask, float* prob, int64_t* idx, uint8_t* type,
                                    void* workspace) { return 801; }
int kamd_dibr_soft_mask_forward_f64(void* stream, int B, int H, int W, int F, int K,
                                    const double* img, const double* large_bbox,
                                    const int64_t* sel_idx, float sigmainv, float multiplier,
                                    double* soft_mask, double* prob, int64_t* idx, uint8_t* type,
                                    void* workspace) { return 801; }
int kamd_dibr_soft_mask_backward_f32(void* stream, int B, int H, int W, int F, int K,
                                     const float* grad, const float* soft_mask,
                                     const int64_t* sel_idx, const float* prob,
                                     const int64_t* idx, const uint8_t* type, const float* img,
                                     float sigmainv, float multiplier, float* g_img) { return 801; }
int kamd_dibr_soft_mask_backward_f64(void* stream, int B, int H, int W, int F, int K,
                                     const double* grad, const double* soft_mask,
                                     const int64_t* sel_idx, const double* prob,
                                     const int64_t* idx, const uint8_t* type, const double* img,
                                     float sigmainv, float multiplier, double* g_img) { return 801; }
int kamd_triangle_distance_forward_f32(void* stream, int N, int F,
                                       const float* points, const float* faces,
                                       float* dist, int64_t* face_idx, int32_t* dist_type,
                                       void* workspace) { return 801; }
int kamd_triangle_distance_forward_f64(void* stream, int N, int F,
                                       const double* points, const double* faces,
                                       double* dist, int64_t* face_idx, int32_t* dist_type,
                                       void* workspace) { return 801; }
int kamd_triangle_distance_backward_f32(void* stream, int N, int F,
                                        const float* grad, const float* points,
                                        const float* faces, const int64_t* face_idx,
                                        const int32_t* dist_type, float* g_points,
                                        float* g_faces) { return 801; }
int kamd_triangle_distance_backward_f64(void* stream, int N, int F,
                                        const double* grad, const double* points,
                                        const double* faces, const int64_t* face_idx,
                                        const int32_t* dist_type, double* g_points,
                                        double* g_faces) { return 801; }
int kamd_trianglemeshes_to_voxelgrids_f32(void* stream, int B, int V, int F, int R,
                                          const float* vertices, const int64_t* faces,
                                          float* grid) { return 801; }
int kamd_trianglemeshes_to_voxelgrids_f64(void* stream, int B, int V, int F, int R,
                                          const double* vertices, const int64_t* faces,
                                          double* grid) { return 801; }
}
