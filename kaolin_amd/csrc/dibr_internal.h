// Internal (non-ABI) entry points shared by the DIB-R translation units of libkaolin_amd.so.
#pragma once
#include "tile_lists.h"

namespace kamd {
// rasterize.hip: the rasterizer's tile kernel with the soft mask's classification attached (fused dibr_rasterization)
template <typename T>
int raster2_draw(hipStream_t st, int B, int H, int W, int D, int F_dense, float multiplier, float eps, const T* rec,
                 const tl::Lists& LR, const T* feat, T* interp, int64_t* sel_idx, T* weights, const tl::ClassifyOut& co,
                 bool weights_internal);  // background tiles leave `weights` unwritten (read only where face_idx >= 0)
// rasterize.hip: the rasterizer's backward kernel; tile_cov (one byte per (mesh, 16 x 16 tile), tl::work_cov_offset_words)
// lets workgroups of tiles without a covered pixel leave at once (nullptr: found out from face_idx); row_span: the forward's
// covered-row spans (tl::work_span_offset_words; nullptr: start from the middle of the image)
template <typename T>
int raster_backward_draw(hipStream_t st, int B, int H, int W, int F, int D, const T* grad, const int64_t* face_idx, const T* weights,
                         const T* img, const T* feat, float eps, T* g_img, T* g_feat, const unsigned char* tile_cov,
                         const unsigned int* row_span);
// rasterize.hip: the same over the forward's list of covered tiles (tl::work_covlist_offset_words; counters at
// work[tl::WORK_COV_WORD + s * COUNTER_STRIDE]): a persistent grid, no workgroup for a tile without a covered pixel
template <typename T>
int raster_backward_draw_list(hipStream_t st, int B, int H, int W, int F, int D, const T* grad, const int64_t* face_idx, const T* weights,
                              const T* img, const T* feat, float eps, T* g_img, T* g_feat, const unsigned int* cov_counts,
                              const unsigned int* cov_list, unsigned int cov_cap, const unsigned int* magic_word,
                              unsigned int* bigwork /* work + tl::WORK_BIGHASH_WORD: the hot faces' partial sums are folded in and cleared; or nullptr */);
}  // namespace kamd
