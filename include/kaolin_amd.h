/*
 * kaolin_amd.h -- C ABI of libkaolin_amd.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for Kaolin's DIB-R / 3D-metrics hot path.  Every
 * entry point replaces one function of the reference's pybind11 module
 * `kaolin._C` (kaolin/csrc/bindings.cpp:103-115); the reference binding each one
 * stands in for is cited next to it.  No torch / ATen type crosses this ABI:
 *
 *   - all pointers are DEVICE pointers into memory the caller owns (the Python
 *     shim allocates every output and workspace with torch, so the caching
 *     allocator and stream semantics of the reference are unchanged);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); kernels
 *     are only enqueued, no entry point synchronises the host;
 *   - every function returns a hipError_t as int (0 = hipSuccess) and never
 *     throws; argument-shape checking (the ATen `checkSize` strings the
 *     reference's tests match) is done by the host shim above this ABI;
 *   - functions are re-entrant and may be called concurrently from the forward
 *     thread and autograd's backward thread;
 *   - "accumulate" outputs must be zero-initialised by the caller exactly where
 *     the reference requires it (at::zeros / zeros_like in the .cpp wrappers).
 *
 * dtype suffixes: _f32 float, _f64 double, _f16 IEEE half (passed as uint16_t*).
 * Index tensors are int64_t as in the reference (at::kLong).
 */
#ifndef KAOLIN_AMD_H_
#define KAOLIN_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library identification. Returns a static string "kaolin_amd <ver> gfx950". */
const char* kamd_version(void);

/* ------------------------------------------------------------------------- */
/* metrics.sided_distance_forward_cuda(p1, p2) -> [dist, idx]                 */
/* reference: kaolin/csrc/metrics/sided_distance.cpp:65-89,                   */
/*            kaolin/csrc/metrics/sided_distance_cuda.cu:52-201,244-267       */
/* p1 (B,N,3), p2 (B,M,3) contiguous; dist (B,N); idx (B,N) int64.            */
/* dist[b,i] = min_j |p1[b,i]-p2[b,j]|^2, idx = lowest j attaining it.        */
/* `workspace` must hold kamd_sided_distance_forward_workspace(B,N,M,esize)   */
/* bytes (may be NULL when that size is 0).                                   */
/* ------------------------------------------------------------------------- */
size_t kamd_sided_distance_forward_workspace(int B, int N, int M, int elem_size);
int kamd_sided_distance_forward_f32(void* stream, int B, int N, int M,
                                    const float* p1, const float* p2,
                                    float* dist, int64_t* idx, void* workspace);
int kamd_sided_distance_forward_f64(void* stream, int B, int N, int M,
                                    const double* p1, const double* p2,
                                    double* dist, int64_t* idx, void* workspace);
int kamd_sided_distance_forward_f16(void* stream, int B, int N, int M,
                                    const uint16_t* p1, const uint16_t* p2,
                                    uint16_t* dist, int64_t* idx, void* workspace);
/* integer clouds (the reference dispatches Byte / Short / Int / Long as well, */
/* kaolin/csrc/utils.h:50-64): arithmetic in the element type, C promotions,   */
/* truncation at every assignment as in sided_distance_cuda.cu:83-86           */
int kamd_sided_distance_forward_u8(void* stream, int B, int N, int M, const uint8_t* p1, const uint8_t* p2,
                                   uint8_t* dist, int64_t* idx, void* workspace);
int kamd_sided_distance_forward_i16(void* stream, int B, int N, int M, const int16_t* p1, const int16_t* p2,
                                    int16_t* dist, int64_t* idx, void* workspace);
int kamd_sided_distance_forward_i32(void* stream, int B, int N, int M, const int32_t* p1, const int32_t* p2,
                                    int32_t* dist, int64_t* idx, void* workspace);
int kamd_sided_distance_forward_i64(void* stream, int B, int N, int M, const int64_t* p1, const int64_t* p2,
                                    int64_t* dist, int64_t* idx, void* workspace);

/* Both directions of chamfer_distance in one pass (kaolin/metrics/pointcloud.py */
/* :89-136 calls sided_distance(p1, p2) and sided_distance(p2, p1)): both clouds */
/* are binned once, each on its own grid, and serve as targets of one direction  */
/* and as queries of the other.  dist1/idx1 (B,N) are p1 -> p2, dist2/idx2 (B,M) */
/* are p2 -> p1, bit-identical to two kamd_sided_distance_forward calls.         */
/* The workspace query returns 0 when the shapes do not qualify (a cloud below   */
/* 8192 points, non-fp32): callers then issue the two calls.                     */
size_t kamd_sided_distance_pair_forward_workspace(int B, int N, int M, int elem_size);
int kamd_sided_distance_pair_forward_f32(void* stream, int B, int N, int M,
                                         const float* p1, const float* p2,
                                         float* dist1, int64_t* idx1,
                                         float* dist2, int64_t* idx2, void* workspace);
int kamd_sided_distance_pair_forward_f64(void* stream, int B, int N, int M,
                                         const double* p1, const double* p2,
                                         double* dist1, int64_t* idx1,
                                         double* dist2, int64_t* idx2, void* workspace);

/* Gradient of chamfer_distance (kaolin/metrics/pointcloud.py:120-136) w.r.t.   */
/* both clouds in one launch: grad (B) is the gradient of the (B) result; the    */
/* autograd chain weight -> mean -> [sqrt] -> sided_distance backward            */
/* (sided_distance_cuda.cu:203-242) of both directions is evaluated per point.  */
/* g1 (B,N,3) and g2 (B,M,3) are overwritten (no initialisation needed).         */
int kamd_chamfer_distance_backward_f32(void* stream, int B, int N, int M,
                                       const float* grad, float w1, float w2, int squared,
                                       const float* p1, const float* p2,
                                       const int64_t* idx1, const int64_t* idx2,
                                       const float* dist1, const float* dist2,
                                       float* g1, float* g2);

/* chamfer_distance as ONE operator (kaolin/metrics/pointcloud.py:89-136, fp32):  */
/* out (B) = w1 * mean_i f(dist1_i) + w2 * mean_j f(dist2_j), f = identity        */
/* (squared != 0) or sqrt, from workspace fill + build + search = 3 launches; the */
/* means are accumulated in double inside the search launch.  dist1/idx1/dist2/  */
/* idx2 are optional outputs (NULL = not wanted).  with_grad != 0: the search     */
/* also leaves d out / d p (per unit of upstream gradient) in the workspace,      */
/* which the caller then keeps alive until kamd_chamfer_distance_backward_fused   */
/* (one launch: g1 (B,N,3), g2 (B,M,3) = grad[b] * those, overwritten).           */
/* ..._workspace returns 0 when the shapes do not qualify (see pair_forward).     */
size_t kamd_chamfer_distance_forward_workspace(int B, int N, int M, int with_grad);
int kamd_chamfer_distance_forward_f32(void* stream, int B, int N, int M,
                                      const float* p1, const float* p2, float w1, float w2,
                                      int squared, int with_grad, float* out,
                                      float* dist1, int64_t* idx1, float* dist2, int64_t* idx2,
                                      void* workspace);
int kamd_chamfer_distance_backward_fused_f32(void* stream, int B, int N, int M, const float* grad,
                                             void* workspace, float* g1, float* g2);

/* metrics.sided_distance_backward_cuda(grad, p1, p2, idx) -> [g1, g2]        */
/* reference: sided_distance.cpp:91-122, sided_distance_cuda.cu:203-242       */
/* g1 (B,N,3) is overwritten; g2 (B,M,3) is accumulated (caller zeroes it).   */
int kamd_sided_distance_backward_f32(void* stream, int B, int N, int M,
                                     const float* grad, const float* p1, const float* p2,
                                     const int64_t* idx, float* g1, float* g2);
int kamd_sided_distance_backward_f64(void* stream, int B, int N, int M,
                                     const double* grad, const double* p1, const double* p2,
                                     const int64_t* idx, double* g1, double* g2);
int kamd_sided_distance_backward_f16(void* stream, int B, int N, int M,
                                     const uint16_t* grad, const uint16_t* p1, const uint16_t* p2,
                                     const int64_t* idx, uint16_t* g1, uint16_t* g2);
int kamd_sided_distance_backward_u8(void* stream, int B, int N, int M, const uint8_t* grad, const uint8_t* p1,
                                    const uint8_t* p2, const int64_t* idx, uint8_t* g1, uint8_t* g2);
int kamd_sided_distance_backward_i16(void* stream, int B, int N, int M, const int16_t* grad, const int16_t* p1,
                                     const int16_t* p2, const int64_t* idx, int16_t* g1, int16_t* g2);
int kamd_sided_distance_backward_i32(void* stream, int B, int N, int M, const int32_t* grad, const int32_t* p1,
                                     const int32_t* p2, const int64_t* idx, int32_t* g1, int32_t* g2);
int kamd_sided_distance_backward_i64(void* stream, int B, int N, int M, const int64_t* grad, const int64_t* p1,
                                     const int64_t* p2, const int64_t* idx, int64_t* g1, int64_t* g2);

/* ------------------------------------------------------------------------- */
/* render.mesh.packed_rasterize_forward_cuda(H, W, z, img, bbox, feat,        */
/*     first_idx_face_per_mesh, multiplier, eps) -> [feat, sel_idx, weights]  */
/* reference: kaolin/csrc/render/mesh/rasterization.cpp:49-104,               */
/*            rasterization_cuda.cu:43-236                                    */
/* Packed inputs over F' = first_idx[B] faces: z (F',3), img (F',3,2) already */
/* multiplied by `multiplier`, bbox (F',4) = [xmin,ymin,xmax,ymax],           */
/* feat (F',3,D); first_idx (B+1) int64 ON DEVICE.                            */
/* Outputs (fully written, no pre-fill needed): interp (B,H,W,D),             */
/* sel_idx (B,H,W) int64 relative to the mesh's first packed face (-1 none),  */
/* weights (B,H,W,3).  `workspace`: kamd_rasterize_forward_workspace bytes.   */
/* ------------------------------------------------------------------------- */
size_t kamd_rasterize_forward_workspace(int B, int H, int W, int64_t total_faces, int elem_size);
int kamd_packed_rasterize_forward_f32(void* stream, int B, int H, int W, int D,
                                      int64_t total_faces,
                                      const float* z, const float* img, const float* bbox,
                                      const float* feat, const int64_t* first_idx,
                                      float multiplier, float eps,
                                      float* interp, int64_t* sel_idx, float* weights,
                                      void* workspace);
int kamd_packed_rasterize_forward_f64(void* stream, int B, int H, int W, int D,
                                      int64_t total_faces,
                                      const double* z, const double* img, const double* bbox,
                                      const double* feat, const int64_t* first_idx,
                                      float multiplier, float eps,
                                      double* interp, int64_t* sel_idx, double* weights,
                                      void* workspace);

/* render.mesh.rasterize_backward_cuda(grad, interp, sel_idx, weights, img,   */
/*     feat, eps) -> [g_img, g_feat]                                          */
/* reference: rasterization.cpp:106-168, rasterization_cuda.cu:238-442        */
/* img (B,F,3,2) UNSCALED, feat (B,F,3,D), face_idx (B,H,W) mesh-relative.    */
/* g_img (B,F,3,2), g_feat (B,F,3,D) are accumulated (caller zeroes them).    */
/* Works for any B*H*W >= 1 (the reference launches 0 blocks below 512 px).   */
/* g_feat may be NULL when the caller does not need the feature gradient      */
/* (autograd's needs_input_grad): it is then not computed.                     */
int kamd_rasterize_backward_f32(void* stream, int B, int H, int W, int F, int D,
                                const float* grad, const int64_t* face_idx,
                                const float* weights, const float* img, const float* feat,
                                float eps, float* g_img, float* g_feat);
int kamd_rasterize_backward_f64(void* stream, int B, int H, int W, int F, int D,
                                const double* grad, const int64_t* face_idx,
                                const double* weights, const double* img, const double* feat,
                                float eps, double* g_img, double* g_feat);

/* ------------------------------------------------------------------------- */
/* render.mesh.dibr_soft_mask_forward_cuda(img*mult, large_bbox, sel_idx,     */
/*     sigmainv, knum, multiplier) -> [soft_mask, prob, idx, type]            */
/* reference: kaolin/csrc/render/mesh/dibr_soft_mask.cpp:48-108,              */
/*            dibr_soft_mask_cuda.cu:27-228                                   */
/* img (B,F,3,2) scaled, large_bbox (B,F,4), sel_idx (B,H,W) int64.           */
/* Outputs fully written: soft_mask (B,H,W), prob (B,H,W,K) (0 fill),         */
/* idx (B,H,W,K) int64 (-1 fill), type (B,H,W,K) uint8 (0 fill).              */
/* hit_count (B,H,W) uint8 is OPTIONAL (NULL = not produced; not part of the  */
/* reference's interface): number of K-buffer entries written per pixel,      */
/* saturated at 255; the backward uses it to skip pixels without hits.        */
/* work: kamd_dibr_soft_mask_work_words(B,H,W) 32-bit words receiving the      */
/* search's worklist (scratch for this operator).                             */
/* ------------------------------------------------------------------------- */
size_t kamd_dibr_soft_mask_forward_workspace(int B, int H, int W, int F, int K, int elem_size);
int kamd_dibr_soft_mask_forward_f32(void* stream, int B, int H, int W, int F, int K,
                                    const float* img, const float* large_bbox,
                                    const int64_t* sel_idx, float sigmainv, float multiplier,
                                    float* soft_mask, float* prob, int64_t* idx, uint8_t* type,
                                    void* workspace, uint8_t* hit_count, uint32_t* work);
int kamd_dibr_soft_mask_forward_f64(void* stream, int B, int H, int W, int F, int K,
                                    const double* img, const double* large_bbox,
                                    const int64_t* sel_idx, float sigmainv, float multiplier,
                                    double* soft_mask, double* prob, int64_t* idx, uint8_t* type,
                                    void* workspace, uint8_t* hit_count, uint32_t* work);

/* render.mesh.dibr_soft_mask_backward_cuda(grad, soft_mask, sel_idx, prob,   */
/*     idx, type, img*mult, sigmainv, multiplier) -> g_img                    */
/* reference: dibr_soft_mask.cpp:110-183, dibr_soft_mask_cuda.cu:230-402      */
/* g_img (B,F,3,2) is accumulated (caller zeroes it).                         */
int kamd_dibr_soft_mask_backward_f32(void* stream, int B, int H, int W, int F, int K,
                                     const float* grad, const float* soft_mask,
                                     const int64_t* sel_idx, const float* prob,
                                     const int64_t* idx, const uint8_t* type, const float* img,
                                     float sigmainv, float multiplier, float* g_img,
                                     const uint8_t* hit_count);
int kamd_dibr_soft_mask_backward_f64(void* stream, int B, int H, int W, int F, int K,
                                     const double* grad, const double* soft_mask,
                                     const int64_t* sel_idx, const double* prob,
                                     const int64_t* idx, const uint8_t* type, const double* img,
                                     float sigmainv, float multiplier, double* g_img,
                                     const uint8_t* hit_count);

/* Compact-list variant used by this package's own autograd Function (not part  */
/* of the reference's interface): same search, same soft_mask, but instead of   */
/* the (B,H,W,K) K-buffers the accepted (pixel, face) hits are kept as lists:   */
/*  hit_pair  (two int32 per record: the mesh-relative face, and (pixel of the  */
/*            sub-tile 0..63) << 16 | rank of the hit among the pixel's hits):  */
/*            what the search leaves for the evaluation, SEGMENTED: a work item */
/*            (a 16x4-pixel sub-tile that has hits to search), item = (32x32    */
/*            tile * B + b) * 16 + sub-tile, owns the records [item*64*K,        */
/*            item*64*K + item_count[item]), FACE-MAJOR (the pixels of one face  */
/*            are consecutive);                                                   */
/*  hit_rec / hit_prob  the evaluated hits as a flat list in 64 shards of equal    */
/*            capacity (capacity / 64 records each; shard s holds work[256 + 32 s]   */
/*            records; an item appends one contiguous range to shard item % 64):    */
/*            hit_rec = two int32 {(b*F + face) | which-of-six << 29, row << 16 |   */
/*            col}, hit_prob the probability -- all the backward pass reads (it     */
/*            streams the shards in rounds of 256 records, no items).               */
/*            Requires B*F < 2^29, H, W < 2^16.                                     */
/* item_count holds ceil(W/32)*ceil(H/32)*16*B ints.  `work`                      */
/* (kamd_dibr_soft_mask_work_words 32-bit words) receives the worklist: 8 sharded */
/* item counters (words 32 s), the flat list's 64 shard counts and the covered-tile */
/* list's 32, one per 128-byte line (same-line device atomics serialise), in a    */
/* 3328-word header, then the items {item, uncovered-pixel mask (2 words), pairs  */
/* the search accepted}.  The fused dibr_rasterization forward appends one byte   */
/* per (mesh, 16x16-pixel tile): does the tile hold a covered pixel; a copy of    */
/* the covered tile rows per view (2 words each); and the list of the tiles that  */
/* hold a covered pixel (32 shards of ceil(B/8)*ceil(tiles/4) words), which the   */
/* rasterizer's backward walks.  Word 1 of the header receives a signature of the  */
/* layout (a hash of B, H, W) from the forward's last launch: the fused backward  */
/* walks the covered-tile list only when it finds it there and visits every tile  */
/* otherwise.  Requires B*H*W < 2^31.                                              */
/* Records each of the three hit arrays must hold (64*K per sub-tile slot).       */
size_t kamd_dibr_soft_mask_lean_capacity(int B, int H, int W, int K);
size_t kamd_dibr_soft_mask_work_words(int B, int H, int W);
int kamd_dibr_soft_mask_forward_lean_f32(void* stream, int B, int H, int W, int F, int K,
                                         const float* img, const float* large_bbox,
                                         const int64_t* sel_idx, float sigmainv, float multiplier,
                                         float* soft_mask, int32_t* hit_pair,
                                         float* hit_prob, int32_t* hit_rec, int32_t* item_count,
                                         uint32_t* work, void* workspace);
int kamd_dibr_soft_mask_forward_lean_f64(void* stream, int B, int H, int W, int F, int K,
                                         const double* img, const double* large_bbox,
                                         const int64_t* sel_idx, float sigmainv, float multiplier,
                                         double* soft_mask, int32_t* hit_pair,
                                         double* hit_prob, int32_t* hit_rec, int32_t* item_count,
                                         uint32_t* work, void* workspace);
int kamd_dibr_soft_mask_backward_lean_f32(void* stream, int B, int H, int W, int F, int K,
                                          const float* grad, const float* soft_mask,
                                          const int32_t* hit_pair,
                                          const float* hit_prob, const int32_t* hit_rec,
                                          const int32_t* item_count, const uint32_t* work,
                                          const float* img,
                                          double img_scale, float sigmainv, float multiplier,
                                          float* g_img);
int kamd_dibr_soft_mask_backward_lean_f64(void* stream, int B, int H, int W, int F, int K,
                                          const double* grad, const double* soft_mask,
                                          const int32_t* hit_pair,
                                          const double* hit_prob, const int32_t* hit_rec,
                                          const int32_t* item_count, const uint32_t* work,
                                          const double* img,
                                          double img_scale, float sigmainv, float multiplier,
                                          double* g_img);
/* Fused front doors (ours): take the RAW operator inputs of the Python layer and */
/* fold its torch glue into the bin kernel -- rasterization.py:292-327 (packing   */
/* of valid faces with torch.where = a host sync, x multiplier, per-face min/max) */
/* and dibr.py:31-39 (x multiplier, boxes enlarged by margin = boxlen*multiplier).*/
/* valid: (B,F) bytes, NULL = all faces.  face_idx comes out mesh-relative.       */
/* In the lean backward, img_scale multiplies img on the fly (pass the multiplier */
/* with the unscaled vertices, or 1 with already scaled ones).                    */
int kamd_rasterize_forward_fused_f32(void* stream, int B, int H, int W, int F, int D,
                                     const float* z, const float* img, const float* feat,
                                     const uint8_t* valid, double multiplier, float eps,
                                     float* interp, int64_t* face_idx, float* weights,
                                     void* workspace);
int kamd_rasterize_forward_fused_f64(void* stream, int B, int H, int W, int F, int D,
                                     const double* z, const double* img, const double* feat,
                                     const uint8_t* valid, double multiplier, float eps,
                                     double* interp, int64_t* face_idx, double* weights,
                                     void* workspace);
/* _strided: z (B,F,3) and the optional per-face scalar `front` (B,F) are read in place through ELEMENT strides    */
/* (z[f * z_face_stride + k * z_vertex_stride], front[f * front_stride]; batch items must be F faces apart), e.g. */
/* the [..., 2] views of the camera-space vertices / face normals; a face is kept when valid[f] != 0 (if given)   */
/* and front[f] >= 0 (if given) -- the mask `face_normals_z >= 0` of dibr_rasterization (dibr.py:188).           */
int kamd_rasterize_forward_fused_strided_f32(void* stream, int B, int H, int W, int F, int D, const float* z,
                                             int64_t z_face_stride, int64_t z_vertex_stride,
                                             const float* img, const float* feat, const uint8_t* valid,
                                             const float* front, int64_t front_stride, double multiplier,
                                             float eps, float* interp, int64_t* face_idx, float* weights,
                                             void* workspace);
int kamd_rasterize_forward_fused_strided_f64(void* stream, int B, int H, int W, int F, int D, const double* z,
                                             int64_t z_face_stride, int64_t z_vertex_stride,
                                             const double* img, const double* feat, const uint8_t* valid,
                                             const double* front, int64_t front_stride, double multiplier,
                                             float eps, double* interp, int64_t* face_idx, double* weights,
                                             void* workspace);
int kamd_dibr_soft_mask_forward_fused_f32(void* stream, int B, int H, int W, int F, int K,
                                          const float* img, double multiplier, double margin,
                                          const int64_t* sel_idx, float sigmainv,
                                          float* soft_mask, int32_t* hit_pair,
                                          float* hit_prob, int32_t* hit_rec, int32_t* item_count,
                                          uint32_t* work, void* workspace);
int kamd_dibr_soft_mask_forward_fused_f64(void* stream, int B, int H, int W, int F, int K,
                                          const double* img, double multiplier, double margin,
                                          const int64_t* sel_idx, float sigmainv,
                                          double* soft_mask, int32_t* hit_pair,
                                          double* hit_prob, int32_t* hit_rec, int32_t* item_count,
                                          uint32_t* work, void* workspace);

/* ------------------------------------------------------------------------- */
/* kaolin.metrics.render.mask_iou(lhs, rhs) (kaolin/metrics/render.py:18-40),  */
/* fused: forward = one pass over both (B, H, W) masks, P = H * W (+ a one-     */
/* workgroup finish): sums (B, 2) doubles receive {sum(l*r), sum(l+r-l*r)} per  */
/* item, loss the scalar 1 - mean(I / (U + 1e-10)); backward = one elementwise  */
/* pass giving d loss / d(one mask) from the OTHER mask and the sums.           */
/* workspace: kamd_mask_iou_workspace(B) bytes.                                 */
/* ------------------------------------------------------------------------- */
size_t kamd_mask_iou_workspace(int B);
int kamd_mask_iou_forward_f32(void* stream, int B, int64_t P, const float* lhs, const float* rhs, void* workspace,
                              double* sums, float* loss);
int kamd_mask_iou_forward_f64(void* stream, int B, int64_t P, const double* lhs, const double* rhs, void* workspace,
                              double* sums, double* loss);
int kamd_mask_iou_backward_f32(void* stream, int B, int64_t P, const float* grad_loss, const float* other,
                               const double* sums, float* grad);
int kamd_mask_iou_backward_f64(void* stream, int B, int64_t P, const double* grad_loss, const double* other,
                               const double* sums, double* grad);
/* The linear loss sum(x1 * w1) + sum(x2 * w2) over two G-buffers of one render  */
/* (image features and soft mask against fixed weights; no reference operator -- */
/* in torch: two dots, an add, two full-size products backward).  Forward: one   */
/* pass over the four arrays + a one-workgroup finish, out = the scalar; n2 = 0  */
/* for a single pair.  Backward: g1 = grad_out[0] * w1, g2 = grad_out[0] * w2 in */
/* one pass (a NULL gradient is skipped).  workspace: kamd_weighted_sum2_workspace() bytes. */
size_t kamd_weighted_sum2_workspace(void);
int kamd_weighted_sum2_forward_f32(void* stream, int64_t n1, const float* x1, const float* w1, int64_t n2,
                                   const float* x2, const float* w2, void* workspace, float* out);
int kamd_weighted_sum2_forward_f64(void* stream, int64_t n1, const double* x1, const double* w1, int64_t n2,
                                   const double* x2, const double* w2, void* workspace, double* out);
int kamd_weighted_sum2_backward_f32(void* stream, const float* grad_out, int64_t n1, const float* w1, float* g1,
                                    int64_t n2, const float* w2, float* g2);
int kamd_weighted_sum2_backward_f64(void* stream, const double* grad_out, int64_t n1, const double* w1, double* g1,
                                    int64_t n2, const double* w2, double* g2);
/* kaolin.render.mesh.texture_mapping (kaolin/render/mesh/utils.py:23-76), one  */
/* gather kernel each way: uv (B, N, 2) OpenGL-style in [0, 1] (clamped), tex   */
/* (B, C, TH, TW), out (B, N, C); grid_sample arithmetic (align_corners=False,  */
/* border padding), bilinear != 0 or nearest.  Backward accumulates into the    */
/* caller-zeroed g_tex (NULL: not needed) and writes g_uv (B, N, 2) (NULL: not  */
/* needed; zero for nearest).                                                   */
int kamd_texture_mapping_forward_f32(void* stream, int B, int64_t N, int C, int TH, int TW, int bilinear,
                                     const float* uv, const float* tex, float* out);
int kamd_texture_mapping_forward_f64(void* stream, int B, int64_t N, int C, int TH, int TW, int bilinear,
                                     const double* uv, const double* tex, double* out);
int kamd_texture_mapping_backward_f32(void* stream, int B, int64_t N, int C, int TH, int TW, int bilinear,
                                      const float* uv, const float* tex, const float* grad_out, float* g_tex,
                                      float* g_uv);
int kamd_texture_mapping_backward_f64(void* stream, int B, int64_t N, int C, int TH, int TW, int bilinear,
                                      const double* uv, const double* tex, const double* grad_out, double* g_tex,
                                      double* g_uv);

/* ------------------------------------------------------------------------- */
/* dibr_rasterization in one call (ours; kaolin/render/mesh/dibr.py:119-209   */
/* = rasterize with valid faces + dibr_soft_mask over all faces).  Same       */
/* kernels as the separate entry points, sharing one binning pass: the faces  */
/* are binned for both operators in one launch per phase, and the rasterizer's */
/* tile kernel classifies the pixels for the soft mask.  The two backward      */
/* kernels do not depend on each other; they are enqueued one after the other  */
/* on `stream` (with KAMD_BWD_SIDE_STREAM=1 the soft mask's goes to an         */
/* internal side stream forked from / joined to `stream` with events: still    */
/* stream-ordered for the caller).  g_img (zeroed by the caller)               */
/* receives BOTH gradient contributions; g_feat may be NULL (feature gradient  */
/* not needed: not computed).  workspace: kamd_dibr_rasterization_workspace.   */
/* `weights` is meaningful only where face_idx >= 0 (background tiles do not write */
/* it: the backward reads it only there).                                         */
/* `work` of ..._backward is the forward's, unchanged in between; the backward   */
/* uses a part of it as scratch (per-XCD partial sums of the gradients of faces  */
/* whose enlarged box spans more than 8 x 8 soft tiles; left cleared).           */
/* grad_img_to_zero (optional, (B,F,3,2)): cleared by the forward's one fill      */
/* launch so that the caller can hand it to ..._backward as g_img without a fill */
/* launch of its own.                                                            */
/* ------------------------------------------------------------------------- */
size_t kamd_dibr_rasterization_workspace(int B, int H, int W, int F, int K, int elem_size);
int kamd_dibr_rasterization_forward_f32(void* stream, int B, int H, int W, int F, int D, int K,
                                        const float* z, int64_t z_face_stride, int64_t z_vertex_stride,
                                        const float* img, const float* feat, const uint8_t* valid,
                                        const float* front, int64_t front_stride, double multiplier,
                                        float eps, float sigmainv, double margin, float* interp,
                                        int64_t* face_idx, float* weights, float* soft_mask,
                                        int32_t* hit_pair, float* hit_prob,
                                        int32_t* hit_rec, int32_t* item_count, uint32_t* work,
                                        void* workspace, float* grad_img_to_zero);
int kamd_dibr_rasterization_forward_f64(void* stream, int B, int H, int W, int F, int D, int K,
                                        const double* z, int64_t z_face_stride, int64_t z_vertex_stride,
                                        const double* img, const double* feat, const uint8_t* valid,
                                        const double* front, int64_t front_stride, double multiplier,
                                        float eps, float sigmainv, double margin, double* interp,
                                        int64_t* face_idx, double* weights, double* soft_mask,
                                        int32_t* hit_pair, double* hit_prob,
                                        int32_t* hit_rec, int32_t* item_count, uint32_t* work,
                                        void* workspace, double* grad_img_to_zero);
int kamd_dibr_rasterization_backward_f32(void* stream, int B, int H, int W, int F, int D, int K,
                                         const float* grad_feat, const float* grad_soft,
                                         const int64_t* face_idx, const float* weights,
                                         const float* soft_mask, const int32_t* hit_pair,
                                         const float* hit_prob,
                                         const int32_t* hit_rec, const int32_t* item_count,
                                         uint32_t* work, const float* img, const float* feat,
                                         double multiplier, float eps, float sigmainv,
                                         float* g_img, float* g_feat);
int kamd_dibr_rasterization_backward_f64(void* stream, int B, int H, int W, int F, int D, int K,
                                         const double* grad_feat, const double* grad_soft,
                                         const int64_t* face_idx, const double* weights,
                                         const double* soft_mask, const int32_t* hit_pair,
                                         const double* hit_prob,
                                         const int32_t* hit_rec, const int32_t* item_count,
                                         uint32_t* work, const double* img, const double* feat,
                                         double multiplier, float eps, float sigmainv,
                                         double* g_img, double* g_feat);

/* ------------------------------------------------------------------------- */
/* render.mesh.prepare_vertices, fused (the reference is     */
/* pure torch: kaolin/render/mesh/utils.py:128-175).  vertices (Bv,V,3) with    */
/* batch stride `vstride` elements (0 = one mesh shared by all B views),        */
/* faces (F,3) int64, proj (3); camera either rot (B,3,3) + trans (B,3) or      */
/* transform (B,4,3) (the other pointers NULL).  Outputs fv_cam (B,F,3,3),      */
/* fv_img (B,F,3,2), unit normals (B,F,3).  Backward: gradient w.r.t. the       */
/* vertices only, g_vertices (B,V,3) fully written; any of g_cam / g_img /      */
/* g_nrm may be NULL; adj_offsets (V+1) / adj_entries (3F, values face*3+k)     */
/* list each vertex's incident face corners.                                    */
/* ------------------------------------------------------------------------- */
int kamd_prepare_vertices_forward_f32(void* stream, int B, int V, int F, const float* vertices,
                                      int64_t vstride, const int64_t* faces, const float* proj,
                                      const float* rot, const float* trans, const float* transform,
                                      float* fv_cam, float* fv_img, float* normals);
int kamd_prepare_vertices_forward_f64(void* stream, int B, int V, int F, const double* vertices,
                                      int64_t vstride, const int64_t* faces, const double* proj,
                                      const double* rot, const double* trans, const double* transform,
                                      double* fv_cam, double* fv_img, double* normals);
int kamd_prepare_vertices_backward_f32(void* stream, int B, int V, int F, const float* vertices,
                                       int64_t vstride, const int64_t* faces, const float* proj,
                                       const float* rot, const float* trans, const float* transform,
                                       const int32_t* adj_offsets, const int32_t* adj_entries,
                                       const float* g_cam, const float* g_img, const float* g_nrm,
                                       float* g_vertices);
int kamd_prepare_vertices_backward_f64(void* stream, int B, int V, int F, const double* vertices,
                                       int64_t vstride, const int64_t* faces, const double* proj,
                                       const double* rot, const double* trans, const double* transform,
                                       const int32_t* adj_offsets, const int32_t* adj_entries,
                                       const double* g_cam, const double* g_img, const double* g_nrm,
                                       double* g_vertices);

/* ------------------------------------------------------------------------- */
/* render.mesh.deftet_sparse_render_forward_cuda(face_vertices_z,             */
/*     face_vertices_image, face_bboxes, pixel_coords, pixel_depth_ranges,    */
/*     knum, eps) -> [face_idx, pixel_depths, w0, w1]                          */
/* (reference: kaolin/csrc/render/mesh/deftet.cpp:47-108,   */
/* deftet_cuda.cu:31-232).  face_vertices_z (B,F,3), face_vertices_image      */
/* (B,F,3,2), face_bboxes (B,F,4) = (xmin,ymin,xmax,ymax), pixel_coords and    */
/* pixel_depth_ranges (B,P,2).  Outputs (B,P,K), FULLY written here (the       */
/* reference pre-fills them: -1 / -inf / 0 / 0): per pixel the first K faces   */
/* in mesh order whose box [min,max) and triangle contain the pixel with       */
/* min_depth <= depth < max_depth, unsorted.  workspace: sorted pixels, cell    */
/* table, per-pixel counters and two work lists,                               */
/* kamd_deftet_forward_workspace(B,F,P,elem_size) bytes (elem_size = 4 | 8),   */
/* data-independent, no initialisation needed.                                 */
/*                                                                             */
/* _forward_fused = the whole of DeftetSparseRenderer.forward                  */
/* (kaolin/render/mesh/deftet.py:269-315): the operator above, then hits       */
/* sorted by depth (descending, equal depths keep mesh order),                 */
/* w2 = 1 - (w0 + w1), corner features weighted and summed.  tmp_* (B,P,K) and */
/* hit_count (B,P) int32 are uninitialised scratch; sorted_face_idx (B,P,K),   */
/* weights (B,P,K,3), interpolated_features (B,P,K,D) are fully written.       */
/*                                                                             */
/* render.mesh.deftet_sparse_render_backward_cuda(grad, face_idx, weights,    */
/*     face_vertices_image, face_features, eps) -> [g_image, g_features]       */
/* (deftet.cpp:110-161, deftet_cuda.cu:240-449): grad (B,P,K,D), face_idx      */
/* (B,P,K), weights (B,P,K,3); both gradients ACCUMULATE: caller zero-fills    */
/* (the reference wrapper allocates them with zeros_like).                     */
/* ------------------------------------------------------------------------- */
size_t kamd_deftet_forward_workspace(int B, int F, int P, int elem_size);
int kamd_deftet_sparse_render_forward_f32(void* stream, int B, int F, int P, int K, const float* face_vertices_z,
        const float* face_vertices_image, const float* face_bboxes, const float* pixel_coords,
        const float* pixel_depth_ranges, float eps, int64_t* face_idx, float* pixel_depths, float* w0, float* w1,
        void* workspace, size_t workspace_bytes);
int kamd_deftet_sparse_render_forward_fused_f32(void* stream, int B, int F, int P, int K, int D,
        const float* face_vertices_z, const float* face_vertices_image, const float* face_bboxes,
        const float* pixel_coords, const float* pixel_depth_ranges, const float* face_features, float eps,
        int64_t* tmp_face_idx, float* tmp_depths, float* tmp_w0, float* tmp_w1, int32_t* hit_count,
        int64_t* sorted_face_idx, float* weights, float* interpolated_features, void* workspace,
        size_t workspace_bytes);
int kamd_deftet_sparse_render_backward_f32(void* stream, int B, int F, int P, int K, int D,
        const float* grad_interpolated_features, const int64_t* face_idx, const float* weights,
        const float* face_vertices_image, const float* face_features, float eps,
        float* grad_face_vertices_image, float* grad_face_features);
int kamd_deftet_sparse_render_forward_f64(void* stream, int B, int F, int P, int K, const double* face_vertices_z,
        const double* face_vertices_image, const double* face_bboxes, const double* pixel_coords,
        const double* pixel_depth_ranges, float eps, int64_t* face_idx, double* pixel_depths, double* w0, double* w1,
        void* workspace, size_t workspace_bytes);
int kamd_deftet_sparse_render_forward_fused_f64(void* stream, int B, int F, int P, int K, int D,
        const double* face_vertices_z, const double* face_vertices_image, const double* face_bboxes,
        const double* pixel_coords, const double* pixel_depth_ranges, const double* face_features, float eps,
        int64_t* tmp_face_idx, double* tmp_depths, double* tmp_w0, double* tmp_w1, int32_t* hit_count,
        int64_t* sorted_face_idx, double* weights, double* interpolated_features, void* workspace,
        size_t workspace_bytes);
int kamd_deftet_sparse_render_backward_f64(void* stream, int B, int F, int P, int K, int D,
        const double* grad_interpolated_features, const int64_t* face_idx, const double* weights,
        const double* face_vertices_image, const double* face_features, float eps,
        double* grad_face_vertices_image, double* grad_face_features);

/* ------------------------------------------------------------------------- */
/* ops.conversions.mesh_to_spc_cuda(face_vertices, level)                      */
/*     -> [octree uint8 (num_nodes), face_ids int64 (num_voxels),              */
/*         barycoords float (num_voxels, 2)]                                   */
/* (reference: kaolin/csrc/ops/conversions/mesh_to_spc/      */
/* mesh_to_spc.cpp:27-42, mesh_to_spc_cuda.cu:98-467, ops/spc/spc_cuda.cu:46-160)*/
/* face_vertices (F,3,3) float32 in [-1,1]^3.  Result sizes depend on the data, */
/* the library never allocates, so the operator is a short host sequence:      */
/*  1. proposals = (Morton code 0, triangle f) for every f; level_from = 0.     */
/*  2. stage_count(level_from, level_to = min(level_from + stage_levels(),     */
/*     level)): counts + offsets (n + 1, offsets[n] = total, read by the host); */
/*     stage_emit writes the `total` surviving (code, triangle) pairs at        */
/*     level_to; repeat from level_to with tested = 1 until level_to == level.  */
/*  3. build: stable sort by code, first triangle of every voxel, all octree    */
/*     levels, into the workspace; sizes[0] = voxels, sizes[1 + l] = nodes of   */
/*     octree level l (device int64[1 + level], read by the host).              */
/*  4. results: face_ids, barycoords of the point of the triangle closest to    */
/*     the voxel centre, octree bytes (levels concatenated root first).         */
/* ------------------------------------------------------------------------- */
int kamd_mesh_to_spc_stage_levels(void);
size_t kamd_mesh_to_spc_scan_workspace(int64_t n);
int kamd_mesh_to_spc_stage_count(void* stream, int64_t n, const float* face_vertices, const int64_t* morton,
                                 const int64_t* triangle_id, int level_from, int level_to, int tested,
                                 int32_t* counts, int64_t* offsets, void* scan_workspace);
int kamd_mesh_to_spc_stage_emit(void* stream, int64_t n, const float* face_vertices, const int64_t* morton,
                                const int64_t* triangle_id, int level_from, int level_to, int tested,
                                const int64_t* offsets, int64_t* morton_out, int64_t* triangle_id_out);
size_t kamd_mesh_to_spc_build_workspace(int64_t n, int level);
int kamd_mesh_to_spc_build(void* stream, int64_t n, int level, const int64_t* morton,
                           const int64_t* triangle_id, void* workspace, size_t workspace_bytes,
                           int64_t* sizes);
int kamd_mesh_to_spc_results(void* stream, int64_t n, int level, const float* face_vertices,
                             const void* workspace, int64_t num_voxels, int64_t octree_bytes,
                             uint8_t* octree, int64_t* face_ids, float* barycoords);

/* ------------------------------------------------------------------------- */
/* ops.mesh.unbatched_mesh_intersection_cuda(points, v1, v2, v3) -> result     */
/* (reference: kaolin/csrc/ops/mesh/mesh_intersection.cpp,  */
/* mesh_intersection_cuda.cu:101-253).  points (N,3); v1,v2,v3 (F,3) = the      */
/* faces' three vertices; result (N) fully written = number of faces the +x ray */
/* from each point crosses (check_sign takes its parity).                       */
/* ------------------------------------------------------------------------- */
int kamd_mesh_intersection_f32(void* stream, int N, int F, const float* points, const float* v1,
                               const float* v2, const float* v3, float* result);
int kamd_mesh_intersection_f64(void* stream, int N, int F, const double* points, const double* v1,
                               const double* v2, const double* v3, double* result);

/* ------------------------------------------------------------------------- */
/* metrics.unbatched_triangle_distance_forward_cuda(points, faces, dist,      */
/*     face_idx, dist_type) -> void                                           */
/* reference: kaolin/csrc/metrics/unbatched_triangle_distance.cpp:43-72,      */
/*            unbatched_triangle_distance_cuda.cu:237-317,418-443             */
/* points (N,3), faces (F,3,3); caller-allocated dist (N), face_idx (N) int64,*/
/* dist_type (N) int32 (0 plane, 1-3 vertex, 4-6 edge).                       */
/* ------------------------------------------------------------------------- */
size_t kamd_triangle_distance_forward_workspace(int N, int F, int elem_size);
int kamd_triangle_distance_forward_f32(void* stream, int N, int F,
                                       const float* points, const float* faces,
                                       float* dist, int64_t* face_idx, int32_t* dist_type,
                                       void* workspace);
int kamd_triangle_distance_forward_f64(void* stream, int N, int F,
                                       const double* points, const double* faces,
                                       double* dist, int64_t* face_idx, int32_t* dist_type,
                                       void* workspace);

/* metrics.unbatched_triangle_distance_backward_cuda(grad, points, faces,     */
/*     face_idx, dist_type, g_points, g_faces) -> void                        */
/* reference: unbatched_triangle_distance.cpp:74-114, .cu:319-416,445-474     */
/* g_points (N,3) overwritten; g_faces (F,3,3) accumulated (caller zeroes).   */
int kamd_triangle_distance_backward_f32(void* stream, int N, int F,
                                        const float* grad, const float* points,
                                        const float* faces, const int64_t* face_idx,
                                        const int32_t* dist_type, float* g_points,
                                        float* g_faces);
int kamd_triangle_distance_backward_f64(void* stream, int N, int F,
                                        const double* grad, const double* points,
                                        const double* faces, const int64_t* face_idx,
                                        const int32_t* dist_type, double* g_points,
                                        double* g_faces);

/* ------------------------------------------------------------------------- */
/* ops.conversions.trianglemeshes_to_voxelgrids (dense) -- the reference has  */
/* NO native kernel here (pure torch: kaolin/ops/conversions/trianglemesh.py: */
/* 29-110, ops/mesh/trianglemesh.py:410-458, ops/conversions/pointcloud.py:   */
/* 42-75); this entry point fuses subdivide-until-dense + point binning.      */
/* vertices (B,V,3) RAW, faces (F,3) int64 shared by the batch.  origin (B,3)  */
/* and scale (B) of the normalisation (v - origin) / scale, either may be NULL:*/
/* then origin = per-mesh minimum, scale = largest extent above the origin     */
/* (trianglemesh.py:84-96).  norm: scratch of                                 */
/* kamd_trianglemeshes_to_voxelgrids_workspace(B,V,elem_size) bytes (its first */
/* 4*B scalars return origin xyz + scale per mesh).  grid (B,R,R,R) is fully   */
/* written (0/1 in dtype).                                                     */
/* ------------------------------------------------------------------------- */
size_t kamd_trianglemeshes_to_voxelgrids_workspace(int B, int V, int elem_size);
int kamd_trianglemeshes_to_voxelgrids_f32(void* stream, int B, int V, int F, int R,
                                          const float* vertices, const int64_t* faces,
                                          const float* origin, const float* scale, float* norm,
                                          float* grid);
int kamd_trianglemeshes_to_voxelgrids_f64(void* stream, int B, int V, int F, int R,
                                          const double* vertices, const int64_t* faces,
                                          const double* origin, const double* scale, double* norm,
                                          double* grid);

/* The same marks as one BIT per voxel (the sparse result of trianglemeshes_to_voxelgrids(return_sparse=True): the reference builds  */
/* its COO tensor from the unique voxel indices and never allocates R^3 scalars, kaolin/ops/conversions/pointcloud.py:66-73).       */
/* bits: B * kamd_trianglemeshes_to_voxelbits_words(R) uint32 words, cleared by the call; voxel ((x R + y) R + z) of mesh b = bit     */
/* (lin & 31) of word b * words + (lin >> 5).  norm: as kamd_trianglemeshes_to_voxelgrids_*.                                          */
size_t kamd_trianglemeshes_to_voxelbits_words(int R);
int kamd_trianglemeshes_to_voxelbits_f32(void* stream, int B, int V, int F, int R, const float* vertices,
                                         const int64_t* faces, const float* origin, const float* scale, float* norm,
                                         uint32_t* bits);
int kamd_trianglemeshes_to_voxelbits_f64(void* stream, int B, int V, int F, int R, const double* vertices,
                                         const int64_t* faces, const double* origin, const double* scale, double* norm,
                                         uint32_t* bits);

/* ------------------------------------------------------------------------- */
/* Optional per-kernel timing (HIP events recorded on the launch stream).      */
/* Not part of the reference's interface: used by bench.py for its roofline    */
/* line; off by default.  kamd_profile_read synchronises the pending events.  */
/* ------------------------------------------------------------------------- */
int kamd_profile_enable(int on);
/* id >= 0: only that kernel is timed while profiling is on; -1: every kernel */
int kamd_profile_select(int id);
int kamd_profile_reset(void);
int kamd_profile_num_kernels(void);
const char* kamd_profile_kernel_name(int id);
int kamd_profile_read(int id, double* total_ms, int64_t* launches);

/* Work counters of the exact triangle-distance search (no reference counterpart; bench.py's VALU figure for C5).  While `on`,      */
/* every kamd_triangle_distance_forward_* call that takes the sweep counts its work and SYNCHRONISES to read the counts back;        */
/* out8 (optional) receives the last call's: tiles staged, tile walks (per wavefront), (query, tile) sphere tests, face steps (per   */
/* wavefront of 64 lanes), closest-point evaluations (unbatched_triangle_distance_cuda.cu:237-317, one per lane), sum of tiles       */
/* needed per query, queries needing > 48 tiles, hard queries.  Off by default: the product path counts nothing.                     */
int kamd_triangle_distance_work_counters(int on, unsigned long long* out8);

/* ---- self-test hook (no reference counterpart) ------------------------------------------------------------------ */
/* The 64 x 64 bit transpose of the soft mask's select kernel (csrc/tile_bins.h wave_transpose64: gfx950 lane-swap and DPP      */
/* instructions, no LDS) applied to `n_matrices` matrices of 64 uint64 rows each: out[m][j] = column j of matrix m.  With       */
/* reference != 0 the same through __shfl_xor, the form the fast one replaced.  tests/test_dibr_gpu.py checks both against numpy. */
int kamd_debug_transpose64(void* stream, int n_matrices, const uint64_t* in, uint64_t* out, int reference);

#ifdef __cplusplus
}
#endif
#endif  /* KAOLIN_AMD_H_ */
