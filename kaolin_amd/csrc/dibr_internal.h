// Internal (non-ABI) entry points shared by the DIB-R translation units of libkaolin_amd.so.
#pragma once
#include "tile_lists.h"

namespace kamd {
// rasterize.hip: the rasterizer's tile kernel with the soft mask's classification attached (fused dibr_rasterization)
template <typename T>
int raster2_draw(hipStream_t st, int B, int H, int W, int D, int F_dense, float multiplier, float eps, const T* rec,
                 const tl::Lists& LR, const T* feat, T* interp, int64_t* sel_idx, T* weights, const tl::ClassifyOut& co,
                 bool weights_internal);  // background tiles leave `weights` unwritten (read only where face_idx >= 0)
}  // namespace kamd
