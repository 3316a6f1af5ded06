"""``deftet_sparse_render``: every intersection of a mesh with a set of free pixel coordinates, depth sorted
(the volumetric renderer of DefTet, Gao et al. NeurIPS 2020).

Mirror of kaolin/render/mesh/deftet.py:269-417 over the HIP operators of ``kaolin_amd._C.render.mesh``.  The forward
is ONE library call (search + depth sort + interpolation, ``deftet_sparse_render_forward_fused``) where the reference
runs its CUDA operator followed by argsort / three gathers / pad / stack / sum in torch.
"""
import torch

from ... import _C

__all__ = ['deftet_sparse_render']


class DeftetSparseRenderer(torch.autograd.Function):
    """Same inputs, outputs and saved state as the reference's Function of that name (deftet.py:269-331)."""

    @staticmethod
    def forward(ctx, pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features, knum, eps):
        pixel_coords, render_ranges = pixel_coords.contiguous(), render_ranges.contiguous()
        face_vertices_z = face_vertices_z.contiguous()
        face_vertices_image, face_features = face_vertices_image.contiguous(), face_features.contiguous()
        boxes = torch.cat((face_vertices_image.min(dim=2)[0], face_vertices_image.max(dim=2)[0]), dim=2)
        features, face_idx, weights = _C.render.mesh.deftet_sparse_render_forward_fused(
            face_vertices_z, face_vertices_image, boxes, pixel_coords, render_ranges, face_features, knum, eps)
        ctx.save_for_backward(face_idx, weights, face_vertices_image, face_features)
        ctx.mark_non_differentiable(face_idx)
        ctx.set_materialize_grads(False)   # no (B, P, knum) int64 zeros for the index output's gradient
        ctx.eps = eps
        return features, face_idx

    @staticmethod
    def backward(ctx, grad_features, grad_face_idx):
        if grad_features is None:
            return (None,) * 7
        face_idx, weights, face_vertices_image, face_features = ctx.saved_tensors
        g_img, g_feat = _C.render.mesh.deftet_sparse_render_backward_cuda(
            grad_features.contiguous(), face_idx, weights, face_vertices_image, face_features, ctx.eps)
        return None, None, None, g_img, g_feat, None, None


def deftet_sparse_render(pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features, knum=300,
                         eps=1e-8):
    r"""Renders, for every pixel coordinate, ALL the faces it intersects inside its depth range, nearest first.

    Not differentiable w.r.t. ``pixel_coords``, ``render_ranges`` and ``face_vertices_z``.  When a pixel intersects
    more than ``knum`` faces, the ``knum`` first in mesh order are kept (as in the reference's CUDA operator) and then
    sorted; intersections at exactly the same depth keep mesh order.

    Args:
        pixel_coords (torch.Tensor): :math:`(\text{batch_size}, \text{num_pixels}, 2)`, in the image-plane
            coordinates of ``face_vertices_image`` ([-1, 1] with the camera functions of this package).
        render_ranges (torch.Tensor): :math:`(\text{batch_size}, \text{num_pixels}, 2)` = (min, max) depth per pixel;
            an intersection is kept when ``min <= depth < max`` (depths in front of the camera are negative).
        face_vertices_z (torch.Tensor): :math:`(\text{batch_size}, \text{num_faces}, 3)`.
        face_vertices_image (torch.Tensor): :math:`(\text{batch_size}, \text{num_faces}, 3, 2)`.
        face_features (torch.Tensor or list of torch.Tensor):
            :math:`(\text{batch_size}, \text{num_faces}, 3, \text{feature_dim})` (or a list of such).
        knum (int): slots per pixel. Default: 300.
        eps (float): added (with the sign of the denominator) when normalising barycentric weights. Default: 1e-8.

    Returns:
        (torch.Tensor or tuple of torch.Tensor, torch.LongTensor):
            features :math:`(\text{batch_size}, \text{num_pixels}, \text{knum}, \text{feature_dim})` (a tuple split
            like the input list) and face indices :math:`(\text{batch_size}, \text{num_pixels}, \text{knum})`, -1 = void.
    """
    is_list = isinstance(face_features, (list, tuple))
    feats = torch.cat(face_features, dim=-1) if is_list else face_features
    image_features, face_idx = DeftetSparseRenderer.apply(
        pixel_coords, render_ranges, face_vertices_z, face_vertices_image, feats, knum, eps)
    if is_list:
        sizes = [f.shape[-1] for f in face_features]
        image_features = tuple(torch.split(image_features, sizes, dim=-1))
    return image_features, face_idx


def _naive_deftet_sparse_render(pixel_coords, render_ranges, face_vertices_z, face_vertices_image, face_features, knum,
                                valid_faces=None, eps=1e-8):
    r"""Plain-torch rendition of :func:`deftet_sparse_render` on any device -- the oracle the reference's rasterizer and
    DefTet tests are pinned to (kaolin/render/mesh/deftet.py:101-267).  Per pixel: the valid faces whose half-open
    bounding box holds it, whose three normalised edge functions are >= 0 and whose interpolated depth lies strictly
    inside the pixel's render range, nearest first (largest depth; equal depths keep mesh order), at most ``knum``.
    Unlike the operator it keeps the NEAREST ``knum`` faces of a crowded pixel, not the first in mesh order.

    Returns (features (B, P, knum, D) or a tuple of them for a list input, face_idx (B, P, knum) int64, -1 = void);
    differentiable w.r.t. ``face_vertices_image`` and ``face_features``."""
    as_list = isinstance(face_features, (list, tuple))
    feats = torch.cat(face_features, dim=-1) if as_list else face_features
    B, P = pixel_coords.shape[:2]
    F, D = face_vertices_z.shape[1], feats.shape[-1]
    dev = pixel_coords.device
    assert pixel_coords.shape == (B, P, 2) and render_ranges.shape == (B, P, 2)
    assert face_vertices_z.shape == (B, F, 3) and face_vertices_image.shape == (B, F, 3, 2) and feats.shape == (B, F, 3, D)
    if valid_faces is None:
        valid_faces = torch.ones((B, F), dtype=torch.bool, device=dev)
    face_idx = torch.full((B, P, knum), -1, dtype=torch.long, device=dev)
    with torch.no_grad():
        img, z = face_vertices_image.detach(), face_vertices_z.detach()
        lo, hi = img.min(dim=2)[0], img.max(dim=2)[0]                      # (B, F, 2)
        step = max(1, (1 << 21) // max(F, 1))
        for b in range(B):
            for p0 in range(0, P, step):
                px = pixel_coords[b, p0:p0 + step].detach().unsqueeze(1)    # (c, 1, 2)
                rng = render_ranges[b, p0:p0 + step].detach()
                inbox = ((px >= lo[b]) & (px < hi[b])).all(dim=-1) & valid_faces[b]
                e = img[b].unsqueeze(0) - px.unsqueeze(2)                    # (c, F, 3, 2): vertex - pixel
                w0 = e[:, :, 1, 0] * e[:, :, 2, 1] - e[:, :, 1, 1] * e[:, :, 2, 0]
                w1 = e[:, :, 2, 0] * e[:, :, 0, 1] - e[:, :, 2, 1] * e[:, :, 0, 0]
                w2 = e[:, :, 0, 0] * e[:, :, 1, 1] - e[:, :, 0, 1] * e[:, :, 1, 0]
                total = w0 + w1 + w2
                total = total + eps * torch.sign(total)
                w0, w1, w2 = w0 / total, w1 / total, w2 / total
                depth = w0 * z[b, :, 0] + w1 * z[b, :, 1] + w2 * z[b, :, 2]
                hit = inbox & (w0 >= 0.) & (w1 >= 0.) & (w2 >= 0.) & (depth > rng[:, :1]) & (depth < rng[:, 1:])
                key = torch.where(hit, depth, torch.full_like(depth, -float('inf')))
                order = torch.argsort(key, dim=1, descending=True, stable=True)[:, :knum]
                took = torch.gather(hit, 1, order)
                n = order.shape[1]
                face_idx[b, p0:p0 + step, :n] = torch.where(took, order, torch.full_like(order, -1))
    # differentiable part: barycentric weights of every pixel in its selected faces, through the first vertex and the two
    # edge vectors leaving it (void slots read a zero face with unit area)
    safe = face_idx.clamp(min=0)
    void = face_idx < 0
    corners = torch.gather(face_vertices_image, 1, safe.reshape(B, -1, 1, 1).expand(-1, -1, 3, 2)).reshape(B, P, knum, 3, 2)
    corners = torch.where(void[..., None, None], torch.zeros_like(corners), corners)
    org, eb, ec = corners[..., 0, :], corners[..., 1, :] - corners[..., 0, :], corners[..., 2, :] - corners[..., 0, :]
    rel = pixel_coords.unsqueeze(2) - org
    area = eb[..., 0] * ec[..., 1] - ec[..., 0] * eb[..., 1]
    area = torch.where(void, torch.ones_like(area), area)
    denom = area + eps * torch.sign(area)
    wb = (rel[..., 0] * ec[..., 1] - ec[..., 0] * rel[..., 1]) / denom
    wc = (eb[..., 0] * rel[..., 1] - rel[..., 0] * eb[..., 1]) / denom
    weights = torch.stack([1. - wb - wc, wb, wc], dim=-1)                  # (B, P, knum, 3)
    picked = torch.gather(feats, 1, safe.reshape(B, -1, 1, 1).expand(-1, -1, 3, D)).reshape(B, P, knum, 3, D)
    picked = torch.where(void[..., None, None], torch.zeros_like(picked), picked)
    out = (picked * weights.unsqueeze(-1)).sum(dim=-2)
    if as_list:
        sizes = [f.shape[-1] for f in face_features]
        out = tuple(torch.split(out, sizes, dim=-1))
    return out, face_idx
