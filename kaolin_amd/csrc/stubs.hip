// TEMPORARY: entry points not implemented yet return hipErrorNotSupported (801).
#include "common.h"
#include "../../include/kaolin_amd.h"
extern "C" {
size_t kamd_triangle_distance_forward_workspace(int, int, int) { return 0; }
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
