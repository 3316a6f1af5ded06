// Exact uniform-grid nearest-neighbour search for sided_distance (fp32 and fp64), see sided_distance_grid.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace kamd {
bool sdgrid_applicable(int B, int N, int M);
size_t sdgrid_workspace_bytes(int B, int N, int M, int elem_size = 4);
int sdgrid_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist, int64_t* idx,
                       void* workspace);
int sdgrid_forward_f64(hipStream_t st, int B, int N, int M, const double* p1, const double* p2, double* dist, int64_t* idx,
                       void* workspace);
// both directions of a chamfer distance, each cloud binned once (dist1/idx1: p1 -> p2, dist2/idx2: p2 -> p1)
// at::Half clouds (raw 16-bit storage): the same exact search with c10::Half's arithmetic (sided_distance_cuda.cu:252)
int sdgrid_forward_f16(hipStream_t st, int B, int N, int M, const void* p1, const void* p2, void* dist, int64_t* idx, void* workspace);
bool sdgrid_pair_applicable(int B, int N, int M);
size_t sdgrid_pair_workspace_bytes(int B, int N, int M, int elem_size = 4);
int sdgrid_pair_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist1,
                            int64_t* idx1, float* dist2, int64_t* idx2, void* workspace);
int sdgrid_pair_forward_f64(hipStream_t st, int B, int N, int M, const double* p1, const double* p2, double* dist1,
                            int64_t* idx1, double* dist2, int64_t* idx2, void* workspace);
// chamfer_distance as a whole: the query launch also produces the value and, with_grad, the gradient pieces kept in the
// workspace (which the caller holds on to until sdgrid_chamfer_backward_f32); dist / idx outputs may be null
size_t sdgrid_chamfer_workspace_bytes(int B, int N, int M, bool with_grad);
int sdgrid_chamfer_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float w1, float w2,
                               int squared, bool with_grad, float* out, float* dist1, int64_t* idx1, float* dist2,
                               int64_t* idx2, void* workspace);
int sdgrid_chamfer_backward_f32(hipStream_t st, int B, int N, int M, const float* grad, void* workspace, float* g1,
                                float* g2);
}  // namespace kamd
