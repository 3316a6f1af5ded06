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
