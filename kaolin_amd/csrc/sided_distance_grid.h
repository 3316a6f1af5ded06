// Exact uniform-grid nearest-neighbour search for sided_distance (fp32), see sided_distance_grid.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace kamd {
bool sdgrid_applicable(int B, int N, int M);
size_t sdgrid_workspace_bytes(int B, int N, int M);
int sdgrid_forward_f32(hipStream_t st, int B, int N, int M, const float* p1, const float* p2, float* dist, int64_t* idx,
                       void* workspace);
}  // namespace kamd
