"""Host shims for ``kaolin._C.render.mesh`` (bindings.cpp:109-115): same names, argument order,
allocation rules and ATen check strings; the work is done by libkaolin_amd.so."""
import torch

from ... import _lib
from ..._checks import Arg, check_all_same_gpu, check_all_contiguous, check_size


def packed_rasterize_forward_cuda(height, width, face_vertices_z, face_vertices_image, face_bboxes,
                                  face_features, first_idx_face_per_mesh, multiplier, eps):
    """reference: kaolin/csrc/render/mesh/rasterization.cpp:49-104
    -> [interpolated_features (B,H,W,D), selected_face_idx (B,H,W) int64, output_weights (B,H,W,3)]"""
    fn = 'packed_rasterize_forward_cuda'
    args = [Arg(face_vertices_z, 'face_vertices_z', 3), Arg(face_vertices_image, 'face_vertices_image', 4),
            Arg(face_bboxes, 'face_bboxes', 5), Arg(face_features, 'face_features', 6),
            Arg(first_idx_face_per_mesh, 'first_idx_face_per_mesh', 7)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    num_faces = face_vertices_z.size(0)
    batch_size = first_idx_face_per_mesh.size(0) - 1
    feat_dim = face_features.size(2)
    check_size(fn, args[0], [num_faces, 3])
    check_size(fn, args[1], [num_faces, 3, 2])
    check_size(fn, args[2], [num_faces, 4])
    check_size(fn, args[3], [num_faces, 3, feat_dim])
    check_size(fn, args[4], [batch_size + 1])
    dtype, device = face_vertices_z.dtype, face_vertices_z.device
    sfx = _lib.dtype_suffix(dtype, 'packed_rasterize_forward_cuda')
    for a in args[1:4]:
        if a.t.dtype != dtype:
            raise RuntimeError(f'expected scalar type {_lib._PRETTY[dtype]} but found {_lib._PRETTY.get(a.t.dtype, a.t.dtype)}')
    if first_idx_face_per_mesh.dtype != torch.long:
        raise RuntimeError(f'expected scalar type Long but found {_lib._PRETTY.get(first_idx_face_per_mesh.dtype)}')
    lib = _lib.load()
    with _lib.on_device(device):
        # every element of the three outputs is written by the kernel (uncovered pixels: -1 / 0 / 0),
        # which is what at::full(-1) / at::zeros give in the reference
        sel = torch.empty((batch_size, height, width), dtype=torch.long, device=device)
        wts = torch.empty((batch_size, height, width, 3), dtype=dtype, device=device)
        interp = torch.empty((batch_size, height, width, feat_dim), dtype=dtype, device=device)
        ws = _lib.workspace(lib.kamd_rasterize_forward_workspace(batch_size, height, width, num_faces,
                                                                 face_vertices_z.element_size()), device)
        st = getattr(lib, f'kamd_packed_rasterize_forward_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, feat_dim, num_faces,
            _lib.ptr(face_vertices_z), _lib.ptr(face_vertices_image), _lib.ptr(face_bboxes), _lib.ptr(face_features),
            _lib.ptr(first_idx_face_per_mesh), float(multiplier), float(eps),
            _lib.ptr(interp), _lib.ptr(sel), _lib.ptr(wts), _lib.ptr(ws))
    _lib.check(st, fn)
    return [interp, sel, wts]


def rasterize_backward_cuda(grad_interpolated_features, interpolated_features, selected_face_idx, output_weights,
                            face_vertices_image, face_features, eps, need_feature_grad=True):
    """reference: rasterization.cpp:106-168 -> [grad_face_vertices_image (B,F,3,2), grad_face_features (B,F,3,D)].
    ``need_feature_grad=False`` (not in the reference; autograd's ``needs_input_grad``) skips the second one -> None."""
    fn = 'rasterize_backward_cuda'
    args = [Arg(grad_interpolated_features, 'grad_interpolated_features', 1),
            Arg(interpolated_features, 'interpolated_features', 2),
            Arg(selected_face_idx, 'selected_face_idx', 3), Arg(output_weights, 'output_weights', 4),
            Arg(face_vertices_image, 'face_vertices_image', 5), Arg(face_features, 'face_features', 6)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, height, width, feat_dim = (grad_interpolated_features.size(0), grad_interpolated_features.size(1),
                                           grad_interpolated_features.size(2), grad_interpolated_features.size(3))
    num_faces = face_vertices_image.size(1)
    check_size(fn, args[0], [batch_size, height, width, feat_dim])
    check_size(fn, args[1], [batch_size, height, width, feat_dim])
    check_size(fn, args[2], [batch_size, height, width])
    check_size(fn, args[3], [batch_size, height, width, 3])
    check_size(fn, args[4], [batch_size, num_faces, 3, 2])
    check_size(fn, args[5], [batch_size, num_faces, 3, feat_dim])
    dtype, device = grad_interpolated_features.dtype, grad_interpolated_features.device
    sfx = _lib.dtype_suffix(dtype, 'rasterize_backward_cuda')
    lib = _lib.load()
    with _lib.on_device(device):
        g_img = torch.zeros_like(face_vertices_image)
        g_feat = torch.zeros_like(face_features) if need_feature_grad else None
        st = getattr(lib, f'kamd_rasterize_backward_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, feat_dim,
            _lib.ptr(grad_interpolated_features), _lib.ptr(selected_face_idx), _lib.ptr(output_weights),
            _lib.ptr(face_vertices_image), _lib.ptr(face_features), float(eps), _lib.ptr(g_img), _lib.ptr(g_feat))
    _lib.check(st, fn)
    return [g_img, g_feat]


def dibr_soft_mask_forward_cuda(face_vertices_image, face_large_bboxes, selected_face_idx, sigmainv, knum,
                                multiplier, _with_hit_count=False):
    """reference: kaolin/csrc/render/mesh/dibr_soft_mask.cpp:48-108
    -> [soft_mask (B,H,W), close_face_prob (B,H,W,K), close_face_idx (B,H,W,K) int64,
        close_face_dist_type (B,H,W,K) uint8]; face_vertices_image is already scaled by `multiplier`.
    `_with_hit_count` (ours, not in the reference) appends a (B,H,W) uint8 tensor holding the number of K-buffer
    entries written per pixel, which lets the backward skip pixels without hits."""
    fn = 'dibr_soft_mask_forward_cuda'
    args = [Arg(face_vertices_image, 'face_vertices_image', 1), Arg(face_large_bboxes, 'face_bboxes', 2),
            Arg(selected_face_idx, 'selected_face_idx', 3)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, num_faces = face_vertices_image.size(0), face_vertices_image.size(1)
    height, width = selected_face_idx.size(1), selected_face_idx.size(2)
    check_size(fn, args[0], [batch_size, num_faces, 3, 2])
    check_size(fn, args[1], [batch_size, num_faces, 4])
    check_size(fn, args[2], [batch_size, height, width])
    dtype, device = face_vertices_image.dtype, face_vertices_image.device
    sfx = _lib.dtype_suffix(dtype, 'dibr_soft_mask_forward_cuda')
    knum = int(knum)
    lib = _lib.load()
    with _lib.on_device(device):
        # the library initialises the K-buffers itself (prob 0, idx -1, type 0) in one streaming pass
        soft_mask = torch.empty((batch_size, height, width), dtype=dtype, device=device)
        prob = torch.empty((batch_size, height, width, knum), dtype=dtype, device=device)
        idx = torch.empty((batch_size, height, width, knum), dtype=torch.long, device=device)
        typ = torch.empty((batch_size, height, width, knum), dtype=torch.uint8, device=device)
        hits = torch.empty((batch_size, height, width), dtype=torch.uint8, device=device) if _with_hit_count else None
        ws = _lib.workspace(lib.kamd_dibr_soft_mask_forward_workspace(batch_size, height, width, num_faces, knum,
                                                                      face_vertices_image.element_size()), device)
        work = _work_buffer(batch_size, height, width, device)
        st = getattr(lib, f'kamd_dibr_soft_mask_forward_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, knum,
            _lib.ptr(face_vertices_image), _lib.ptr(face_large_bboxes), _lib.ptr(selected_face_idx),
            float(sigmainv), float(multiplier), _lib.ptr(soft_mask), _lib.ptr(prob), _lib.ptr(idx), _lib.ptr(typ),
            _lib.ptr(ws), _lib.ptr(hits), _lib.ptr(work))
    _lib.check(st, fn)
    return [soft_mask, prob, idx, typ, hits] if _with_hit_count else [soft_mask, prob, idx, typ]


def dibr_soft_mask_backward_cuda(grad_soft_mask, soft_mask, selected_face_idx, close_face_prob, close_face_idx,
                                 close_face_dist_type, face_vertices_image, sigmainv, multiplier, _hit_count=None):
    """reference: dibr_soft_mask.cpp:110-183 -> grad_face_vertices_image (B,F,3,2) (w.r.t. the UNSCALED input)"""
    fn = 'dibr_soft_mask_backward_cuda'
    args = [Arg(grad_soft_mask, 'grad_soft_mask', 1), Arg(soft_mask, 'soft_mask', 2),
            Arg(selected_face_idx, 'selected_face_idx', 3), Arg(close_face_prob, 'close_face_prob', 4),
            Arg(close_face_idx, 'close_face_idx', 5), Arg(close_face_dist_type, 'close_face_dist_type', 6),
            Arg(face_vertices_image, 'face_vertices_image', 7)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, num_faces = face_vertices_image.size(0), face_vertices_image.size(1)
    height, width = selected_face_idx.size(1), selected_face_idx.size(2)
    knum = close_face_idx.size(-1)
    check_size(fn, args[0], [batch_size, height, width])
    check_size(fn, args[1], [batch_size, height, width])
    check_size(fn, args[2], [batch_size, height, width])
    check_size(fn, args[3], [batch_size, height, width, knum])
    check_size(fn, args[4], [batch_size, height, width, knum])
    check_size(fn, args[5], [batch_size, height, width, knum])
    check_size(fn, args[6], [batch_size, num_faces, 3, 2])
    dtype, device = face_vertices_image.dtype, face_vertices_image.device
    sfx = _lib.dtype_suffix(dtype, 'dibr_soft_mask_backward_cuda')
    lib = _lib.load()
    with _lib.on_device(device):
        g_img = torch.zeros_like(face_vertices_image)
        st = getattr(lib, f'kamd_dibr_soft_mask_backward_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, knum,
            _lib.ptr(grad_soft_mask), _lib.ptr(soft_mask), _lib.ptr(selected_face_idx), _lib.ptr(close_face_prob),
            _lib.ptr(close_face_idx), _lib.ptr(close_face_dist_type), _lib.ptr(face_vertices_image),
            float(sigmainv), float(multiplier), _lib.ptr(g_img), _lib.ptr(_hit_count))
    _lib.check(st, fn)
    return g_img


def dibr_soft_mask_forward_lean(face_vertices_image, face_large_bboxes, selected_face_idx, sigmainv, knum, multiplier):
    """Our autograd path's variant of ``dibr_soft_mask_forward_cuda`` (no counterpart in the reference): the same
    search and the same ``soft_mask``, but the per-pixel K-buffers (13*knum bytes per pixel, initialised for every
    pixel) are replaced by a compact, segmented list of the actual (pixel, face, prob, type) hits.
    -> (soft_mask, hits); ``hit_list_entries(hits, knum)`` flattens the list."""
    fn = 'dibr_soft_mask_forward_lean'
    args = [Arg(face_vertices_image, 'face_vertices_image', 1), Arg(face_large_bboxes, 'face_bboxes', 2),
            Arg(selected_face_idx, 'selected_face_idx', 3)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, num_faces = face_vertices_image.size(0), face_vertices_image.size(1)
    height, width = selected_face_idx.size(1), selected_face_idx.size(2)
    check_size(fn, args[0], [batch_size, num_faces, 3, 2])
    check_size(fn, args[1], [batch_size, num_faces, 4])
    check_size(fn, args[2], [batch_size, height, width])
    dtype, device = face_vertices_image.dtype, face_vertices_image.device
    sfx = _lib.dtype_suffix(dtype, fn)
    lib = _lib.load()
    with _lib.on_device(device):
        soft_mask = torch.empty((batch_size, height, width), dtype=dtype, device=device)
        hits = _hit_list(batch_size, height, width, knum, dtype, device, num_faces)
        ws = _lib.workspace(lib.kamd_dibr_soft_mask_forward_workspace(batch_size, height, width, num_faces, int(knum),
                                                                      face_vertices_image.element_size()), device)
        st = getattr(lib, f'kamd_dibr_soft_mask_forward_lean_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, int(knum),
            _lib.ptr(face_vertices_image), _lib.ptr(face_large_bboxes), _lib.ptr(selected_face_idx),
            float(sigmainv), float(multiplier), _lib.ptr(soft_mask), _lib.ptr(hits[0]), _lib.ptr(hits[1]),
            _lib.ptr(hits[2]), _lib.ptr(hits[3]), _lib.ptr(hits[4]), _lib.ptr(ws))
    _lib.check(st, fn)
    return soft_mask, hits


def dibr_soft_mask_backward_lean(grad_soft_mask, soft_mask, hits, face_vertices_image, sigmainv, knum, multiplier,
                                 img_scale=1.0):
    """Backward of :func:`dibr_soft_mask_forward_lean` / ``_fused`` -> grad_face_vertices_image (B,F,3,2), w.r.t. the
    unscaled input.  ``face_vertices_image * img_scale`` must be the scaled vertices the forward searched with."""
    fn = 'dibr_soft_mask_backward_lean'
    hit_pair, hit_prob, hit_rec, item_count, work = hits
    knum = int(knum)
    args = [Arg(grad_soft_mask, 'grad_soft_mask', 1), Arg(soft_mask, 'soft_mask', 2),
            Arg(face_vertices_image, 'face_vertices_image', 4)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, num_faces = face_vertices_image.size(0), face_vertices_image.size(1)
    height, width = soft_mask.size(1), soft_mask.size(2)
    check_size(fn, args[0], [batch_size, height, width])
    dtype, device = face_vertices_image.dtype, face_vertices_image.device
    sfx = _lib.dtype_suffix(dtype, fn)
    lib = _lib.load()
    with _lib.on_device(device):
        g_img = torch.zeros_like(face_vertices_image)
        st = getattr(lib, f'kamd_dibr_soft_mask_backward_lean_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, knum,
            _lib.ptr(grad_soft_mask), _lib.ptr(soft_mask), _lib.ptr(hit_pair), _lib.ptr(hit_prob),
            _lib.ptr(hit_rec), _lib.ptr(item_count), _lib.ptr(work), _lib.ptr(face_vertices_image), float(img_scale), float(sigmainv),
            float(multiplier), _lib.ptr(g_img))
    _lib.check(st, fn)
    return g_img


_SIZES = {}   # (query, shape...) -> size: the library's size queries are pure functions of the shape (a ctypes call each)


def _size(query, *shape):
    key = (query,) + shape
    n = _SIZES.get(key)
    if n is None:
        if len(_SIZES) > 256:
            _SIZES.clear()
        n = _SIZES[key] = int(getattr(_lib.load(), query)(*shape))
    return n


def _work_buffer(batch_size, height, width, device):
    """The search's worklist: 8 sharded item counters (16-word header), then the items {item id, uncovered-pixel mask}."""
    n = max(_size('kamd_dibr_soft_mask_work_words', batch_size, height, width), WORK_HEADER)
    return torch.empty(n, dtype=torch.int32, device=device)


def _hit_list(batch_size, height, width, knum, dtype, device, num_faces=0):
    """Storage of the hit lists: 64*K record slots per 16x4-pixel sub-tile slot (just the used parts are ever touched) for
    the segmented pair records the search leaves {face, pixel << 16 | rank} as (cap, 2) int32, and for the flat list of the
    evaluated hits -- probabilities and (cap, 2) int32 records {(b*F + face) | which << 29, row << 16 | col} --, one count
    per sub-tile slot, and the worklist of the sub-tiles that were searched (its header also holds the flat list's lengths)."""
    cap = max(_size('kamd_dibr_soft_mask_lean_capacity', batch_size, height, width, int(knum)), 1)
    n_sub = ((width + 31) // 32) * ((height + 31) // 32) * 16 * batch_size
    return (torch.empty((cap, 2), dtype=torch.int32, device=device),
            torch.empty(cap, dtype=dtype, device=device), torch.empty((cap, 2), dtype=torch.int32, device=device),
            torch.empty(max(n_sub, 1), dtype=torch.int32, device=device),
            _work_buffer(batch_size, height, width, device))


COUNTER_STRIDE = 32                              # tile_lists.h: append counters sit one per 128-byte line
WORK_FLAT_WORD = 8 * COUNTER_STRIDE              # the flat hit list's shard counters follow the worklist's 8
FLAT_SHARDS = 64                                 # soft2.inc
COV_SHARDS = 32                                  # the covered-tile list's shard counters follow the flat list's
WORK_COV_WORD = WORK_FLAT_WORD + FLAT_SHARDS * COUNTER_STRIDE
WORK_BIGHASH_WORD = WORK_COV_WORD + COV_SHARDS * COUNTER_STRIDE   # the hot faces of the soft backward: count, 4096 bits, 4096 tags, 8 x 4096 x 8 partial sums
WORK_HEADER = WORK_BIGHASH_WORD + 32 + 4096 // 32 + 4096 + 8 * 4096 * 8


def covered_tiles(work, batch_size, height, width):
    """Indices b * ntiles + tile (sorted, 1-D int64) of the 16 x 16 tiles the forward pass found a covered pixel in: the list the
    rasterizer's backward pass walks (tile_lists.h work_covlist_offset_words; tests / debugging; synchronises)."""
    n_groups = batch_size * ((height + 15) // 16) * ((width + 15) // 16)
    cap = ((batch_size + 7) // 8) * ((n_groups // batch_size + 3) // 4)
    off = WORK_HEADER + 8 * (4 * ((n_groups + 7) // 8)) * 4 + (n_groups + 3) // 4 + 2 * batch_size
    counts = work[WORK_COV_WORD:WORK_COV_WORD + COV_SHARDS * COUNTER_STRIDE:COUNTER_STRIDE].tolist()
    lists = work[off:off + COV_SHARDS * cap].view(COV_SHARDS, cap)
    out = [lists[s, :c] for s, c in enumerate(counts) if c]
    return torch.sort(torch.cat(out).long())[0] if out else torch.zeros(0, dtype=torch.long)


def covered_row_spans(work, batch_size, height, width):
    """(B, 2) int tensor: per view (last covered tile row + 1, tiles_y - first covered tile row) as the binning launch reported them
    (0 = nothing reported); the copy the forward leaves next to the worklist for the backward pass (tile_lists.h
    work_span_offset_words; tests / debugging; synchronises)."""
    n_groups = batch_size * ((height + 15) // 16) * ((width + 15) // 16)
    off = WORK_HEADER + 8 * (4 * ((n_groups + 7) // 8)) * 4 + (n_groups + 3) // 4
    return work[off:off + 2 * batch_size].view(batch_size, 2)


def work_items(work, batch_size, height, width):
    """Item ids (int64, 1-D) recorded in a worklist buffer (tests / debugging; synchronises).  Layout (tile_lists.h): the
    header, 8 shards x shard_cap items of 4 words, one coverage byte per 16 x 16 tile, the tile kernels' row order."""
    counts = work[:8 * COUNTER_STRIDE:COUNTER_STRIDE].tolist()
    n_groups = batch_size * ((width + 15) // 16) * ((height + 15) // 16)
    shard_cap = 4 * ((n_groups + 7) // 8)
    items = work[WORK_HEADER:WORK_HEADER + 8 * shard_cap * 4].view(8, shard_cap, 4)
    return torch.cat([items[s, :min(c, shard_cap), 0] for s, c in enumerate(counts)]).long()


def hit_list_entries(hits, num_faces, batch_size, height, width):
    """The flat hit list -> (pix, face, prob, type) 1-D tensors of the recorded hits, pix = flat (b, row, col) index, face
    mesh-relative, type = which-of-six + 1 as in the K-buffers (tests / debugging; synchronises).  The list is FLAT_SHARDS
    lists of equal capacity; shard s holds work[WORK_FLAT_WORD + s * COUNTER_STRIDE] records."""
    hit_pair, hit_prob, hit_rec, item_count, work = hits
    counts = work[WORK_FLAT_WORD:WORK_FLAT_WORD + FLAT_SHARDS * COUNTER_STRIDE:COUNTER_STRIDE].tolist()
    shard_cap = hit_rec.shape[0] // FLAT_SHARDS
    sel = torch.cat([torch.arange(s * shard_cap, s * shard_cap + c, device=hit_rec.device) for s, c in enumerate(counts)])
    rec = hit_rec[sel].long()
    key = rec[:, 0] & ((1 << 29) - 1)
    which = (rec[:, 0] >> 29) & 7
    b, face = key // int(num_faces), key % int(num_faces)
    row, col = (rec[:, 1] >> 16) & 0xFFFF, rec[:, 1] & 0xFFFF
    pix = (b * height + row) * width + col
    return pix, face, hit_prob[sel], which + 1


def dibr_soft_mask_forward_fused(face_vertices_image, selected_face_idx, sigmainv, boxlen, knum, multiplier):
    """``dibr_soft_mask_forward_lean`` taking the Python layer's RAW inputs: the scaling by ``multiplier`` and the
    boxes enlarged by ``boxlen * multiplier`` (dibr.py:31-39) are computed inside the bin kernel.
    -> (soft_mask, hits)"""
    fn = 'dibr_soft_mask_forward_fused'
    args = [Arg(face_vertices_image, 'face_vertices_image', 1), Arg(selected_face_idx, 'selected_face_idx', 2)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, num_faces = face_vertices_image.size(0), face_vertices_image.size(1)
    height, width = selected_face_idx.size(1), selected_face_idx.size(2)
    check_size(fn, args[0], [batch_size, num_faces, 3, 2])
    check_size(fn, args[1], [batch_size, height, width])
    dtype, device = face_vertices_image.dtype, face_vertices_image.device
    sfx = _lib.dtype_suffix(dtype, fn)
    lib = _lib.load()
    with _lib.on_device(device):
        soft_mask = torch.empty((batch_size, height, width), dtype=dtype, device=device)
        hits = _hit_list(batch_size, height, width, knum, dtype, device, num_faces)
        ws = _lib.workspace(lib.kamd_dibr_soft_mask_forward_workspace(batch_size, height, width, num_faces, int(knum),
                                                                      face_vertices_image.element_size()), device)
        st = getattr(lib, f'kamd_dibr_soft_mask_forward_fused_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, int(knum), _lib.ptr(face_vertices_image),
            float(multiplier), float(boxlen * multiplier), _lib.ptr(selected_face_idx), float(sigmainv),
            _lib.ptr(soft_mask), _lib.ptr(hits[0]), _lib.ptr(hits[1]), _lib.ptr(hits[2]), _lib.ptr(hits[3]),
            _lib.ptr(hits[4]), _lib.ptr(ws))
    _lib.check(st, fn)
    return soft_mask, hits


def rasterize_forward_fused(height, width, face_vertices_z, face_vertices_image, face_features, valid_faces,
                            multiplier, eps):
    """``packed_rasterize_forward_cuda`` taking the Python layer's RAW (B,F,...) inputs and the optional valid-face
    mask: packing (torch.where = a host sync + three gathers), scaling and per-face bounding boxes
    (rasterization.py:292-327) happen inside the bin kernel.
    -> [interpolated_features (B,H,W,D), face_idx (B,H,W) int64 mesh-relative, output_weights (B,H,W,3)]"""
    fn = 'rasterize_forward_fused'
    args = [Arg(face_vertices_z, 'face_vertices_z', 3), Arg(face_vertices_image, 'face_vertices_image', 4),
            Arg(face_features, 'face_features', 5)]
    if valid_faces is not None:
        args.append(Arg(valid_faces, 'valid_faces', 6))
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, num_faces, feat_dim = face_vertices_z.size(0), face_vertices_z.size(1), face_features.size(3)
    check_size(fn, args[0], [batch_size, num_faces, 3])
    check_size(fn, args[1], [batch_size, num_faces, 3, 2])
    check_size(fn, args[2], [batch_size, num_faces, 3, feat_dim])
    if valid_faces is not None:
        check_size(fn, args[3], [batch_size, num_faces])
        if valid_faces.dtype not in (torch.bool, torch.uint8):
            raise RuntimeError('valid_faces must be a bool tensor')
    dtype, device = face_vertices_z.dtype, face_vertices_z.device
    sfx = _lib.dtype_suffix(dtype, fn)
    for a in args[1:3]:
        if a.t.dtype != dtype:
            raise RuntimeError(f'expected scalar type {_lib._PRETTY[dtype]} but found {_lib._PRETTY.get(a.t.dtype, a.t.dtype)}')
    lib = _lib.load()
    with _lib.on_device(device):
        face_idx = torch.empty((batch_size, height, width), dtype=torch.long, device=device)
        wts = torch.empty((batch_size, height, width, 3), dtype=dtype, device=device)
        interp = torch.empty((batch_size, height, width, feat_dim), dtype=dtype, device=device)
        ws = _lib.workspace(lib.kamd_rasterize_forward_workspace(batch_size, height, width, batch_size * num_faces,
                                                                 face_vertices_z.element_size()), device)
        st = getattr(lib, f'kamd_rasterize_forward_fused_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, feat_dim, _lib.ptr(face_vertices_z),
            _lib.ptr(face_vertices_image), _lib.ptr(face_features), _lib.ptr(valid_faces), float(multiplier), float(eps),
            _lib.ptr(interp), _lib.ptr(face_idx), _lib.ptr(wts), _lib.ptr(ws))
    _lib.check(st, fn)
    return [interp, face_idx, wts]


def _face_strides(t, inner):
    """(tensor to pass, face stride, inner stride) in ELEMENTS when `t` (B, F[, 3]) can be read in place -- dense, or a
    last-index view such as ``face_vertices_camera[..., 2]`` -- else of its contiguous copy."""
    if inner:
        ok = t.dim() == 3 and t.size(0) * t.size(1) > 0 and t.stride(0) == t.size(1) * t.stride(1) and t.stride(2) > 0
        if not ok:
            t = t.contiguous()
        return t, t.stride(1), t.stride(2)
    ok = t.dim() == 2 and t.numel() > 0 and t.stride(0) == t.size(1) * t.stride(1) and t.stride(1) > 0
    if not ok:
        t = t.contiguous()
    return t, t.stride(1), 1


def dibr_rasterization_forward_fused(height, width, face_vertices_z, face_vertices_image, face_features, valid_faces,
                                     sigmainv, boxlen, knum, multiplier, eps, prepare_grad=False):
    """``rasterize_forward_fused`` + ``dibr_soft_mask_forward_fused`` in one library call (ours): the same kernels
    sharing one binning pass, the rasterizer's tile kernel classifying the pixels for the soft mask.  ``valid_faces`` is either
    the bool mask of the faces to rasterize or a FLOAT (B, F) tensor ``n`` standing for the mask ``n >= 0`` (the face
    normals' z of ``dibr_rasterization``): that one, and ``face_vertices_z``, may be last-index views and are read in
    place.  -> (interpolated_features, face_idx, output_weights, soft_mask, hits, grad_buffer): with ``prepare_grad`` the
    last one is a zeroed ``grad_face_vertices_image`` for :func:`dibr_rasterization_backward_fused`, cleared by the same fill
    launch as the operator's own list heads (a fill launch of its own in the backward costs ~5 us); else ``None``.
    ``output_weights`` is INTERNAL to the autograd node (the backward reads it only where ``face_idx >= 0``): where
    ``face_idx == -1`` its content is UNDEFINED (background tiles do not write it: 12 of their 36 bytes per pixel) -- unlike
    :func:`rasterize_forward_fused` / the reference operator, which store zeros there.  Mask with ``face_idx >= 0`` before
    looking at it."""
    fn = 'dibr_rasterization_forward_fused'
    front = None
    if valid_faces is not None and valid_faces.is_floating_point():
        front, valid_faces = valid_faces, None
    batch_size, num_faces, feat_dim = face_vertices_z.size(0), face_vertices_z.size(1), face_features.size(3)
    extra = valid_faces if valid_faces is not None else front
    # (a hot host path -- the DIB-R step is enqueued from here: one inline test for the usual case, the reference-style checks
    # with their messages only when it fails)
    dev = face_vertices_z.device
    if not (face_vertices_z.is_cuda and face_vertices_image.device == dev and face_features.device == dev and
            face_vertices_image.is_contiguous() and face_features.is_contiguous() and
            face_vertices_z.shape == (batch_size, num_faces, 3) and face_vertices_image.shape == (batch_size, num_faces, 3, 2) and
            face_features.shape == (batch_size, num_faces, 3, feat_dim) and
            (extra is None or (extra.device == dev and extra.shape == (batch_size, num_faces) and
                               (front is not None or extra.is_contiguous())))):
        args = [Arg(face_vertices_z, 'face_vertices_z', 3), Arg(face_vertices_image, 'face_vertices_image', 4),
                Arg(face_features, 'face_features', 5)]
        if valid_faces is not None:
            args.append(Arg(valid_faces, 'valid_faces', 6))
        if front is not None:
            args.append(Arg(front, 'face_normals_z', 6))
        check_all_same_gpu(fn, args)
        check_all_contiguous(fn, args[1:3] + ([args[3]] if valid_faces is not None else []))
        check_size(fn, args[0], [batch_size, num_faces, 3])
        check_size(fn, args[1], [batch_size, num_faces, 3, 2])
        check_size(fn, args[2], [batch_size, num_faces, 3, feat_dim])
        if len(args) > 3:
            check_size(fn, args[3], [batch_size, num_faces])
    dtype, device = face_vertices_z.dtype, face_vertices_z.device
    sfx = _lib.dtype_suffix(dtype, fn)
    for t in (face_vertices_image, face_features, front):
        if t is not None and t.dtype != dtype:
            raise RuntimeError(f'expected scalar type {_lib._PRETTY[dtype]} but found {_lib._PRETTY.get(t.dtype, t.dtype)}')
    lib = _lib.load()
    esz = face_vertices_z.element_size()
    z, z_face, z_vertex = _face_strides(face_vertices_z, True)
    front_stride = 1
    if front is not None:
        front, front_stride, _ = _face_strides(front, False)
    with _lib.on_device(device):
        face_idx = torch.empty((batch_size, height, width), dtype=torch.long, device=device)
        wts = torch.empty((batch_size, height, width, 3), dtype=dtype, device=device)
        interp = torch.empty((batch_size, height, width, feat_dim), dtype=dtype, device=device)
        soft_mask = torch.empty((batch_size, height, width), dtype=dtype, device=device)
        hits = _hit_list(batch_size, height, width, knum, dtype, device, num_faces)
        g_img = torch.empty_like(face_vertices_image) if prepare_grad else None
        ws = _lib.workspace(_size('kamd_dibr_rasterization_workspace', batch_size, height, width, num_faces, int(knum), esz), device)
        st = getattr(lib, f'kamd_dibr_rasterization_forward_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, feat_dim, int(knum),
            _lib.ptr(z), int(z_face), int(z_vertex), _lib.ptr(face_vertices_image), _lib.ptr(face_features),
            _lib.ptr(valid_faces), _lib.ptr(front), int(front_stride),
            float(multiplier), float(eps), float(sigmainv), float(boxlen * multiplier),
            _lib.ptr(interp), _lib.ptr(face_idx), _lib.ptr(wts), _lib.ptr(soft_mask),
            _lib.ptr(hits[0]), _lib.ptr(hits[1]), _lib.ptr(hits[2]), _lib.ptr(hits[3]), _lib.ptr(hits[4]),
            _lib.ptr(ws), _lib.ptr(g_img))
    _lib.check(st, fn)
    return interp, face_idx, wts, soft_mask, hits, g_img


def dibr_rasterization_backward_fused(grad_features, grad_soft_mask, face_idx, output_weights, soft_mask, hits,
                                      face_vertices_image, face_features, sigmainv, knum, multiplier, eps,
                                      need_feature_grad=True, zeroed_grad_image=None):
    """Backward of :func:`dibr_rasterization_forward_fused`: the rasterizer's and the soft mask's backward kernels run
    concurrently and accumulate into ONE grad_face_vertices_image. -> (grad_face_vertices_image, grad_face_features);
    ``need_feature_grad=False`` (static features: autograd's ``needs_input_grad``) skips the second one -> None.
    ``zeroed_grad_image``: the buffer the forward cleared for this call (accumulated into and returned)."""
    fn = 'dibr_rasterization_backward_fused'
    batch_size, height, width, feat_dim = grad_features.shape
    num_faces = face_vertices_image.size(1)
    dev = face_vertices_image.device
    tensors = (grad_features, grad_soft_mask, face_idx, output_weights, soft_mask, face_vertices_image, face_features)
    if not (face_vertices_image.is_cuda and all(t.device == dev and t.is_contiguous() for t in tensors) and
            grad_soft_mask.shape == (batch_size, height, width) and face_idx.shape == (batch_size, height, width) and
            output_weights.shape == (batch_size, height, width, 3) and
            face_vertices_image.shape == (batch_size, num_faces, 3, 2) and
            face_features.shape == (batch_size, num_faces, 3, feat_dim)):
        args = [Arg(grad_features, 'grad_features', 1), Arg(grad_soft_mask, 'grad_soft_mask', 2), Arg(face_idx, 'face_idx', 3),
                Arg(output_weights, 'output_weights', 4), Arg(soft_mask, 'soft_mask', 5),
                Arg(face_vertices_image, 'face_vertices_image', 7), Arg(face_features, 'face_features', 8)]
        check_all_same_gpu(fn, args)
        check_all_contiguous(fn, args)
        check_size(fn, args[1], [batch_size, height, width])
        check_size(fn, args[2], [batch_size, height, width])
        check_size(fn, args[3], [batch_size, height, width, 3])
        check_size(fn, args[5], [batch_size, num_faces, 3, 2])
        check_size(fn, args[6], [batch_size, num_faces, 3, feat_dim])
    dtype, device = face_vertices_image.dtype, face_vertices_image.device
    sfx = _lib.dtype_suffix(dtype, fn)
    lib = _lib.load()
    with _lib.on_device(device):
        g_img = zeroed_grad_image if zeroed_grad_image is not None else torch.zeros_like(face_vertices_image)
        g_feat = torch.zeros_like(face_features) if need_feature_grad else None
        st = getattr(lib, f'kamd_dibr_rasterization_backward_{sfx}')(
            _lib.stream_ptr(device), batch_size, height, width, num_faces, feat_dim, int(knum),
            _lib.ptr(grad_features), _lib.ptr(grad_soft_mask), _lib.ptr(face_idx), _lib.ptr(output_weights),
            _lib.ptr(soft_mask), _lib.ptr(hits[0]), _lib.ptr(hits[1]), _lib.ptr(hits[2]), _lib.ptr(hits[3]),
            _lib.ptr(hits[4]), _lib.ptr(face_vertices_image), _lib.ptr(face_features),
            float(multiplier), float(eps), float(sigmainv), _lib.ptr(g_img), _lib.ptr(g_feat))
    _lib.check(st, fn)
    return g_img, g_feat


def _storage_key(t):
    """What identifies the DATA a tensor shows: a fresh tensor object over the same storage (`faces[:]`, `.view(...)`, a property
    that re-wraps) has a new id() but the same key.  Sound only while a cache entry pins a tensor of that storage (the address
    cannot be recycled) -- every entry below does; an in-place edit bumps `_version` (shared by all views of a storage)."""
    return (t.data_ptr(), tuple(t.shape), t.stride(), t.device, t.dtype, t._version)


class _TopologyCache:
    """Per-`faces` results (mesh topology is static in training).  Two doors: id(faces) -- one dict probe and an `is`, the hot
    path when the caller passes the same tensor object every step -- and, when that misses, the storage key above, so that
    a re-wrapped view does not repeat the reduction / host read / argsort every step."""

    def __init__(self, limit):
        self.by_id, self.by_key, self.limit = {}, {}, limit

    def get(self, faces, num_vertices):
        hit = self.by_id.get(id(faces))
        if hit is not None and hit[0] is faces and hit[1] == faces._version and hit[2] == num_vertices:
            return hit[3]
        hit = self.by_key.get(_storage_key(faces))
        # (the key holds the CALLER's version counter; the entry is only good while the tensor it pins is unchanged since it was
        # cached -- `q = p.data` after an in-place edit of `p` has a fresh counter at 0, the cached `p`'s address, shape and stride,
        # and stale contents behind the entry: ADVICE r04)
        if hit is not None and hit[2] == num_vertices and hit[0]._version == hit[1]:
            return hit[3]
        return None

    def put(self, faces, num_vertices, value):
        if len(self.by_key) > self.limit:
            self.by_id.clear()
            self.by_key.clear()
        entry = (faces, faces._version, int(num_vertices), value)     # (pins `faces`: neither its id nor its address can be reused)
        self.by_id[id(faces)] = entry
        self.by_key[_storage_key(faces)] = entry

    def clear(self):
        self.by_id.clear()
        self.by_key.clear()

    def __len__(self):
        return len(self.by_key)


_ADJ_CACHE = _TopologyCache(8)      # -> (offsets, entries)


def vertex_face_adjacency(faces, num_vertices):
    """CSR list of the (face, corner) incidences of every vertex: (offsets (V+1) int32, entries (3F) int32 = face*3+k).
    Built with three torch ops the first time a `faces` tensor is seen and cached (mesh topology is static in training)."""
    hit = _ADJ_CACHE.get(faces, num_vertices)
    if hit is not None:
        return hit[0], hit[1], faces
    flat = faces.reshape(-1)
    entries = torch.argsort(flat, stable=True).to(torch.int32)
    counts = torch.bincount(flat, minlength=num_vertices)
    offsets = torch.zeros(num_vertices + 1, dtype=torch.int32, device=faces.device)
    offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
    _ADJ_CACHE.put(faces, num_vertices, (offsets, entries))
    return offsets, entries, faces


_FACES_OK = _TopologyCache(16)      # -> bool


def faces_in_range(faces, num_vertices):
    """True when every index of `faces` addresses a vertex (the reference's index_select would raise otherwise; the fused
    kernels read unchecked).  One reduction and one host read per `faces` storage, cached like the adjacency (mesh topology
    is static); the entry keeps `faces` alive, so that neither its id nor its data_ptr can be handed to another tensor
    while the entry exists (a recycled address would hit a stale answer); an in-place edit bumps the version."""
    hit = _FACES_OK.get(faces, num_vertices)
    if hit is not None:
        return hit
    if faces.numel() == 0:
        ok = True
    else:
        lo, hi = torch.stack(torch.aminmax(faces)).tolist()     # min and max in one pass, one synchronising read
        ok = lo >= 0 and hi < num_vertices
    _FACES_OK.put(faces, num_vertices, ok)
    return ok


def _as(t, dtype):
    """`t` as a contiguous tensor of `dtype` (itself when it already is: the usual case, and a hot host path)."""
    if t is None or (t.dtype == dtype and t.is_contiguous()):
        return t
    return t.to(dtype).contiguous()


def _pv_common(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform):
    fn = 'prepare_vertices'
    batch_size = (camera_transform if camera_transform is not None else camera_rot).size(0)
    if vertices.size(0) not in (1, batch_size):
        raise RuntimeError(f'{fn}: vertices batch {vertices.size(0)} does not match the cameras ({batch_size})')
    # a (1, V, 3) tensor or an expanded view (batch stride 0) is one mesh shared by all views
    vstride = 0 if (vertices.size(0) == 1 or vertices.stride(0) == 0) else vertices.size(1) * 3
    v = vertices[:1].contiguous() if vstride == 0 else vertices.contiguous()
    return fn, batch_size, vstride, v


def prepare_vertices_forward_fused(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform):
    """``kaolin.render.mesh.prepare_vertices`` (utils.py:128-175) in one kernel
    -> (face_vertices_camera (B,F,3,3), face_vertices_image (B,F,3,2), face_normals (B,F,3))."""
    fn, B, vstride, v = _pv_common(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform)
    dtype, device = v.dtype, v.device
    sfx = _lib.dtype_suffix(dtype, fn)
    V, F = v.size(1), faces.size(0)
    lib = _lib.load()
    proj, rot, trans, tf = _as(camera_proj.reshape(-1), dtype), _as(camera_rot, dtype), _as(camera_trans, dtype), _as(camera_transform, dtype)
    if not faces.is_contiguous():
        faces = faces.contiguous()
    with _lib.on_device(device):
        fv_cam = torch.empty((B, F, 3, 3), dtype=dtype, device=device)
        fv_img = torch.empty((B, F, 3, 2), dtype=dtype, device=device)
        nrm = torch.empty((B, F, 3), dtype=dtype, device=device)
        st = getattr(lib, f'kamd_prepare_vertices_forward_{sfx}')(
            _lib.stream_ptr(device), B, V, F, _lib.ptr(v), vstride, _lib.ptr(faces), _lib.ptr(proj), _lib.ptr(rot),
            _lib.ptr(trans), _lib.ptr(tf), _lib.ptr(fv_cam), _lib.ptr(fv_img), _lib.ptr(nrm))
    _lib.check(st, fn)
    return fv_cam, fv_img, nrm


def prepare_vertices_backward_fused(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform,
                                    grad_cam, grad_img, grad_nrm):
    """Gradient of :func:`prepare_vertices_forward_fused` w.r.t. the vertices -> (B, V, 3)."""
    fn, B, vstride, v = _pv_common(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform)
    dtype, device = v.dtype, v.device
    sfx = _lib.dtype_suffix(dtype, fn)
    V, F = v.size(1), faces.size(0)
    lib = _lib.load()
    proj, rot, trans, tf = _as(camera_proj.reshape(-1), dtype), _as(camera_rot, dtype), _as(camera_trans, dtype), _as(camera_transform, dtype)
    if not faces.is_contiguous():
        faces = faces.contiguous()
    offsets, entries, _ = vertex_face_adjacency(faces, V)
    g = [t if (t is None or t.is_contiguous()) else t.contiguous() for t in (grad_cam, grad_img, grad_nrm)]
    with _lib.on_device(device):
        g_vertices = torch.empty((B, V, 3), dtype=dtype, device=device)
        st = getattr(lib, f'kamd_prepare_vertices_backward_{sfx}')(
            _lib.stream_ptr(device), B, V, F, _lib.ptr(v), vstride, _lib.ptr(faces), _lib.ptr(proj), _lib.ptr(rot),
            _lib.ptr(trans), _lib.ptr(tf), _lib.ptr(offsets), _lib.ptr(entries), _lib.ptr(g[0]), _lib.ptr(g[1]),
            _lib.ptr(g[2]), _lib.ptr(g_vertices))
    _lib.check(st, fn)
    return g_vertices


# ---- deftet sparse render --------------------------------------------------------------------
def _deftet_forward_args(fn, face_vertices_z, face_vertices_image, face_bboxes, pixel_coords, pixel_depth_ranges):
    args = [Arg(face_vertices_z, 'face_vertices_z', 1), Arg(face_vertices_image, 'face_vertices_image', 2),
            Arg(face_bboxes, 'face_bboxes', 3), Arg(pixel_coords, 'pixel_coords', 4),
            Arg(pixel_depth_ranges, 'pixel_depth_ranges', 5)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, num_faces, num_points = face_vertices_z.size(0), face_vertices_z.size(1), pixel_coords.size(1)
    check_size(fn, args[0], [batch_size, num_faces, 3])
    check_size(fn, args[1], [batch_size, num_faces, 3, 2])
    check_size(fn, args[2], [batch_size, num_faces, 4])
    check_size(fn, args[3], [batch_size, num_points, 2])
    check_size(fn, args[4], [batch_size, num_points, 2])
    return batch_size, num_faces, num_points


def _deftet_workspace(lib, batch_size, num_faces, num_points, dtype, device):
    nbytes = lib.kamd_deftet_forward_workspace(batch_size, num_faces, num_points, dtype.itemsize)
    return torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=device), nbytes


def deftet_sparse_render_forward_cuda(face_vertices_z, face_vertices_image, face_bboxes, pixel_coords,
                                      pixel_depth_ranges, knum, eps):
    """reference: deftet.cpp:47-108 -> [face_idx (B,P,knum) int64, pixel_depths, w0, w1 (B,P,knum)]: per pixel the
    first knum intersected faces in mesh order (unsorted); unused slots hold -1 / -inf / 0 / 0."""
    fn = 'deftet_sparse_render_forward_cuda'
    batch_size, num_faces, num_points = _deftet_forward_args(
        fn, face_vertices_z, face_vertices_image, face_bboxes, pixel_coords, pixel_depth_ranges)
    dtype, device = face_vertices_z.dtype, face_vertices_z.device
    sfx = _lib.dtype_suffix(dtype, fn)
    lib = _lib.load()
    knum = int(knum)
    with _lib.on_device(device):
        face_idx = torch.empty((batch_size, num_points, knum), dtype=torch.long, device=device)
        depths, w0, w1 = (torch.empty((batch_size, num_points, knum), dtype=dtype, device=device) for _ in range(3))
        ws, nbytes = _deftet_workspace(lib, batch_size, num_faces, num_points, dtype, device)
        st = getattr(lib, f'kamd_deftet_sparse_render_forward_{sfx}')(
            _lib.stream_ptr(device), batch_size, num_faces, num_points, knum, _lib.ptr(face_vertices_z),
            _lib.ptr(face_vertices_image), _lib.ptr(face_bboxes), _lib.ptr(pixel_coords), _lib.ptr(pixel_depth_ranges),
            float(eps), _lib.ptr(face_idx), _lib.ptr(depths), _lib.ptr(w0), _lib.ptr(w1), _lib.ptr(ws), nbytes)
    _lib.check(st, fn)
    return [face_idx, depths, w0, w1]


def deftet_sparse_render_forward_fused(face_vertices_z, face_vertices_image, face_bboxes, pixel_coords,
                                       pixel_depth_ranges, face_features, knum, eps):
    """The whole of DeftetSparseRenderer.forward (kaolin/render/mesh/deftet.py:269-315) in one call: the forward
    operator, the depth sort, w2 and the feature interpolation.
    -> [interpolated_features (B,P,knum,D), sorted_face_idx (B,P,knum) int64, weights (B,P,knum,3)]"""
    fn = 'deftet_sparse_render_forward_fused'
    batch_size, num_faces, num_points = _deftet_forward_args(
        fn, face_vertices_z, face_vertices_image, face_bboxes, pixel_coords, pixel_depth_ranges)
    feat = Arg(face_features, 'face_features', 6)
    check_all_same_gpu(fn, [Arg(face_vertices_z, 'face_vertices_z', 1), feat])
    check_all_contiguous(fn, [feat])
    feat_dim = face_features.size(-1)
    check_size(fn, feat, [batch_size, num_faces, 3, feat_dim])
    dtype, device = face_vertices_z.dtype, face_vertices_z.device
    sfx = _lib.dtype_suffix(dtype, fn)
    lib = _lib.load()
    knum = int(knum)
    shape = (batch_size, num_points, knum)
    with _lib.on_device(device):
        tmp_idx = torch.empty(shape, dtype=torch.long, device=device)
        tmp_d, tmp_w0, tmp_w1 = (torch.empty(shape, dtype=dtype, device=device) for _ in range(3))
        hit_count = torch.empty((batch_size, num_points), dtype=torch.int32, device=device)
        sorted_idx = torch.empty(shape, dtype=torch.long, device=device)
        weights = torch.empty(shape + (3,), dtype=dtype, device=device)
        out = torch.empty(shape + (feat_dim,), dtype=dtype, device=device)
        ws, nbytes = _deftet_workspace(lib, batch_size, num_faces, num_points, dtype, device)
        st = getattr(lib, f'kamd_deftet_sparse_render_forward_fused_{sfx}')(
            _lib.stream_ptr(device), batch_size, num_faces, num_points, knum, feat_dim, _lib.ptr(face_vertices_z),
            _lib.ptr(face_vertices_image), _lib.ptr(face_bboxes), _lib.ptr(pixel_coords), _lib.ptr(pixel_depth_ranges),
            _lib.ptr(face_features), float(eps), _lib.ptr(tmp_idx), _lib.ptr(tmp_d), _lib.ptr(tmp_w0), _lib.ptr(tmp_w1),
            _lib.ptr(hit_count), _lib.ptr(sorted_idx), _lib.ptr(weights), _lib.ptr(out), _lib.ptr(ws), nbytes)
    _lib.check(st, fn)
    return [out, sorted_idx, weights]


def deftet_sparse_render_backward_cuda(grad_interpolated_features, face_idx, weights, face_vertices_image,
                                       face_features, eps):
    """reference: deftet.cpp:110-161 -> [grad_face_vertices_image (B,F,3,2), grad_face_features (B,F,3,D)]"""
    fn = 'deftet_sparse_render_backward_cuda'
    args = [Arg(grad_interpolated_features, 'grad_interpolated_features', 1), Arg(face_idx, 'face_idx', 2),
            Arg(weights, 'weights', 3), Arg(face_vertices_image, 'face_vertices_image', 4),
            Arg(face_features, 'face_features', 5)]
    check_all_same_gpu(fn, args)
    check_all_contiguous(fn, args)
    batch_size, num_pixels, knum, feat_dim = (grad_interpolated_features.size(0), grad_interpolated_features.size(1),
                                              grad_interpolated_features.size(2), grad_interpolated_features.size(3))
    num_faces = face_vertices_image.size(1)
    check_size(fn, args[0], [batch_size, num_pixels, knum, feat_dim])
    check_size(fn, args[1], [batch_size, num_pixels, knum])
    check_size(fn, args[2], [batch_size, num_pixels, knum, 3])
    check_size(fn, args[3], [batch_size, num_faces, 3, 2])
    check_size(fn, args[4], [batch_size, num_faces, 3, feat_dim])
    dtype, device = grad_interpolated_features.dtype, grad_interpolated_features.device
    sfx = _lib.dtype_suffix(dtype, fn)
    lib = _lib.load()
    with _lib.on_device(device):
        g_img = torch.zeros_like(face_vertices_image)
        g_feat = torch.zeros_like(face_features)
        st = getattr(lib, f'kamd_deftet_sparse_render_backward_{sfx}')(
            _lib.stream_ptr(device), batch_size, num_faces, num_pixels, knum, feat_dim,
            _lib.ptr(grad_interpolated_features), _lib.ptr(face_idx), _lib.ptr(weights), _lib.ptr(face_vertices_image),
            _lib.ptr(face_features), float(eps), _lib.ptr(g_img), _lib.ptr(g_feat))
    _lib.check(st, fn)
    return [g_img, g_feat]


# ---- mask_iou / texture_mapping (the steps either side of DIB-R in the training loop) ----------------
def mask_iou_forward_fused(lhs_mask, rhs_mask):
    """``kaolin.metrics.render.mask_iou`` forward in one pass -> (loss (scalar), sums (B, 2) float64 {I_b, U_b})."""
    fn = 'mask_iou'
    sfx = _lib.dtype_suffix(lhs_mask.dtype, fn)
    lib = _lib.load()
    B, P = lhs_mask.shape[0], lhs_mask[0].numel() if lhs_mask.shape[0] else 0
    device = lhs_mask.device
    with _lib.on_device(device):
        loss = torch.empty((), dtype=lhs_mask.dtype, device=device)
        sums = torch.empty((B, 2), dtype=torch.float64, device=device)
        ws = _lib.workspace(lib.kamd_mask_iou_workspace(B), device)
        st = getattr(lib, f'kamd_mask_iou_forward_{sfx}')(_lib.stream_ptr(device), B, P, _lib.ptr(lhs_mask), _lib.ptr(rhs_mask),
                                                          _lib.ptr(ws), _lib.ptr(sums), _lib.ptr(loss))
    _lib.check(st, fn)
    return loss, sums


def mask_iou_backward_fused(grad_loss, other_mask, sums):
    """d loss / d(one mask) of :func:`mask_iou_forward_fused` from the OTHER mask and the forward's sums."""
    fn = 'mask_iou'
    sfx = _lib.dtype_suffix(other_mask.dtype, fn)
    lib = _lib.load()
    B, P = other_mask.shape[0], other_mask[0].numel() if other_mask.shape[0] else 0
    device = other_mask.device
    with _lib.on_device(device):
        grad = torch.empty_like(other_mask)
        g = grad_loss.to(other_mask.dtype).reshape(1).contiguous()
        st = getattr(lib, f'kamd_mask_iou_backward_{sfx}')(_lib.stream_ptr(device), B, P, _lib.ptr(g), _lib.ptr(other_mask),
                                                           _lib.ptr(sums), _lib.ptr(grad))
    _lib.check(st, fn)
    return grad


def weighted_sum2_forward(x1, w1, x2=None, w2=None):
    """sum(x1 * w1) (+ sum(x2 * w2)) -> scalar tensor, one pass over the arrays (``kamd_weighted_sum2_forward_*``)."""
    fn = 'weighted_sum2'
    sfx = _lib.dtype_suffix(x1.dtype, fn)
    lib = _lib.load()
    device = x1.device
    n1 = x1.numel()
    n2 = x2.numel() if x2 is not None else 0
    with _lib.on_device(device):
        out = torch.empty((), dtype=x1.dtype, device=device)
        ws = _lib.workspace(lib.kamd_weighted_sum2_workspace(), device)
        st = getattr(lib, f'kamd_weighted_sum2_forward_{sfx}')(
            _lib.stream_ptr(device), n1, _lib.ptr(x1), _lib.ptr(w1), n2, _lib.ptr(x2) if n2 else None,
            _lib.ptr(w2) if n2 else None, _lib.ptr(ws), _lib.ptr(out))
    _lib.check(st, fn)
    return out


def weighted_sum2_backward(grad_out, w1, w2=None, need1=True, need2=True):
    """(grad_out * w1, grad_out * w2) in one pass; a gradient that is not needed (or has no array) is None."""
    fn = 'weighted_sum2'
    sfx = _lib.dtype_suffix(w1.dtype, fn)
    lib = _lib.load()
    device = w1.device
    with _lib.on_device(device):
        g = grad_out.to(w1.dtype).reshape(1).contiguous()
        g1 = torch.empty_like(w1) if need1 else None
        g2 = torch.empty_like(w2) if (need2 and w2 is not None) else None
        st = getattr(lib, f'kamd_weighted_sum2_backward_{sfx}')(
            _lib.stream_ptr(device), _lib.ptr(g), w1.numel(), _lib.ptr(w1), _lib.ptr(g1) if g1 is not None else None,
            w2.numel() if w2 is not None else 0, _lib.ptr(w2) if w2 is not None else None,
            _lib.ptr(g2) if g2 is not None else None)
    _lib.check(st, fn)
    return g1, g2


def texture_mapping_forward_fused(uv, texture_maps, bilinear):
    """uv (B, N, 2), texture_maps (B, C, h, w) -> (B, N, C): clamp, OpenGL -> grid coordinates, sampling and the output layout
    of kaolin/render/mesh/utils.py:23-76 in one gather kernel."""
    fn = 'texture_mapping'
    sfx = _lib.dtype_suffix(uv.dtype, fn)
    lib = _lib.load()
    B, N = uv.shape[0], uv.shape[1]
    C, TH, TW = texture_maps.shape[1:]
    device = uv.device
    with _lib.on_device(device):
        out = torch.empty((B, N, C), dtype=uv.dtype, device=device)
        st = getattr(lib, f'kamd_texture_mapping_forward_{sfx}')(_lib.stream_ptr(device), B, N, C, TH, TW, int(bool(bilinear)),
                                                                 _lib.ptr(uv), _lib.ptr(texture_maps), _lib.ptr(out))
    _lib.check(st, fn)
    return out


def texture_mapping_backward_fused(uv, texture_maps, grad_out, bilinear, need_tex, need_uv):
    """-> (grad_texture_maps or None, grad_uv or None)."""
    fn = 'texture_mapping'
    sfx = _lib.dtype_suffix(uv.dtype, fn)
    lib = _lib.load()
    B, N = uv.shape[0], uv.shape[1]
    C, TH, TW = texture_maps.shape[1:]
    device = uv.device
    with _lib.on_device(device):
        g_tex = torch.zeros_like(texture_maps) if need_tex else None
        g_uv = torch.empty_like(uv) if need_uv else None
        st = getattr(lib, f'kamd_texture_mapping_backward_{sfx}')(
            _lib.stream_ptr(device), B, N, C, TH, TW, int(bool(bilinear)), _lib.ptr(uv), _lib.ptr(texture_maps),
            _lib.ptr(grad_out), _lib.ptr(g_tex), _lib.ptr(g_uv))
    _lib.check(st, fn)
    return g_tex, g_uv
