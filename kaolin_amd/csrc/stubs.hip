// TEMPORARY: entry points not implemented yet return hipErrorNotSupported (801).
#include "common.h"
#include "../../include/kaolin_amd.h"
extern "C" {
int kamd_trianglemeshes_to_voxelgrids_f32(void* stream, int B, int V, int F, int R,
                                          const float* vertices, const int64_t* faces,
                                          float* grid) { return 801; }
int kamd_trianglemeshes_to_voxelgrids_f64(void* stream, int B, int V, int F, int R,
                                          const double* vertices, const int64_t* faces,
                                          double* grid) { return 801; }
}
