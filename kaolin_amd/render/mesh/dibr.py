"""``dibr_soft_mask`` and ``dibr_rasterization`` (API mirror of kaolin/render/mesh/dibr.py:27-209)."""
import torch

from ... import _C
from .rasterization import rasterize

__all__ = ['dibr_soft_mask', 'dibr_rasterization']


class DibrSoftMaskCuda(torch.autograd.Function):
    """Same contract as the reference's DibrSoftMaskCuda (dibr.py:27-73): the vertices are scaled by
    ``multiplier``, the boxes enlarged by ``boxlen * multiplier`` (both inside the bin kernel here, not with torch
    ops), backward returns the gradient w.r.t. the unscaled vertices.  What is kept for backward differs: the reference materialises three
    (B,H,W,knum) K-buffers (390 B/pixel at knum=30); only silhouette-band pixels ever use them, so this Function
    keeps a compact list of the actual hits instead (``_C.render.mesh.dibr_soft_mask_forward_lean``).  The
    K-buffer operators remain available as ``_C.render.mesh.dibr_soft_mask_{forward,backward}_cuda``."""

    @staticmethod
    def forward(ctx, face_vertices_image, selected_face_idx, sigmainv, boxlen, knum, multiplier):
        face_vertices_image = face_vertices_image.contiguous()
        soft_mask, hits = _C.render.mesh.dibr_soft_mask_forward_fused(
            face_vertices_image, selected_face_idx.contiguous(), sigmainv, boxlen, knum, multiplier)
        ctx.multiplier, ctx.sigmainv, ctx.knum = multiplier, sigmainv, knum
        ctx.save_for_backward(soft_mask, face_vertices_image, *hits)
        return soft_mask

    @staticmethod
    def backward(ctx, grad_soft_mask):
        soft_mask, face_vertices_image = ctx.saved_tensors[:2]
        grad = _C.render.mesh.dibr_soft_mask_backward_lean(
            grad_soft_mask.contiguous(), soft_mask, ctx.saved_tensors[2:], face_vertices_image, ctx.sigmainv,
            ctx.knum, ctx.multiplier, img_scale=ctx.multiplier)
        return grad, None, None, None, None, None


def dibr_soft_mask(face_vertices_image, selected_face_idx, sigmainv=7000, boxlen=0.02, knum=30, multiplier=1000.):
    r"""Soft silhouette mask of DIB-R, generally paired with :func:`kaolin_amd.metrics.render.mask_iou`
    (reference: kaolin/render/mesh/dibr.py:75-117).

    Args:
        face_vertices_image (torch.Tensor): 2D vertex positions in [-1, 1], (B, F, 3, 2).
        selected_face_idx (torch.LongTensor): rendered face index (B, H, W) from :func:`rasterize`.
        sigmainv (float): sharpness; recommended [1/3e-4, 1/3e-5]. Default: 7000.
        boxlen (float): bbox margin deciding which pixels a face influences. Default: 0.02.
        knum (int): maximum number of faces influencing one pixel. Default: 30.
        multiplier (float): internal coordinate scale. Default: 1000.

    Returns:
        (torch.FloatTensor): the soft mask, (B, H, W).
    """
    return DibrSoftMaskCuda.apply(face_vertices_image, selected_face_idx, sigmainv, boxlen, knum, multiplier)


class DibrRasterizationCuda(torch.autograd.Function):
    """``rasterize`` (front faces only) + ``dibr_soft_mask`` (all faces) as ONE autograd node: one library call each way
    (``_C.render.mesh.dibr_rasterization_{forward,backward}_fused``), kernels that do not depend on each other run
    concurrently, and both gradient contributions to ``face_vertices_image`` land in one buffer.  Outputs are bit-identical
    to calling the two functions separately (the same kernels)."""

    @staticmethod
    def forward(ctx, height, width, face_vertices_z, face_vertices_image, face_features, valid_faces,
                sigmainv, boxlen, knum, multiplier, eps):
        face_vertices_image = face_vertices_image.contiguous()
        face_features = face_features.contiguous()
        # valid_faces: the bool mask, or the float face-normal z (kept faces: >= 0); z and that scalar may be [..., 2] views
        # of prepare_vertices' outputs -- the library reads them in place (no compare kernel, no contiguous copies)
        if not valid_faces.is_floating_point():
            valid_faces = valid_faces.contiguous()
        # when the image coordinates will be differentiated, the forward's fill launch also clears the gradient buffer the
        # backward kernels accumulate into (held from here to the backward: B*F*6 scalars)
        feats, face_idx, weights, soft_mask, hits, ctx.zeroed_grad = _C.render.mesh.dibr_rasterization_forward_fused(
            height, width, face_vertices_z, face_vertices_image, face_features, valid_faces,
            sigmainv, boxlen, knum, multiplier, eps, prepare_grad=ctx.needs_input_grad[3])
        ctx.save_for_backward(face_idx, weights, soft_mask, face_vertices_image, face_features, *hits)
        ctx.mark_non_differentiable(face_idx)
        # no zero tensors for the gradients of outputs nobody differentiates (the index output alone is B*H*W*8 bytes)
        ctx.set_materialize_grads(False)
        ctx.cfg = (sigmainv, knum, multiplier, eps)
        return feats, soft_mask, face_idx

    @staticmethod
    def backward(ctx, grad_feats, grad_soft_mask, grad_face_idx):
        face_idx, weights, soft_mask, face_vertices_image, face_features = ctx.saved_tensors[:5]
        sigmainv, knum, multiplier, eps = ctx.cfg
        if grad_feats is None and grad_soft_mask is None:
            return (None,) * 11
        if grad_feats is None:
            grad_feats = torch.zeros(soft_mask.shape + (face_features.shape[-1],), dtype=soft_mask.dtype, device=soft_mask.device)
        if grad_soft_mask is None:
            grad_soft_mask = torch.zeros_like(soft_mask)
        zeroed, ctx.zeroed_grad = ctx.zeroed_grad, None     # (a second backward through a retained graph clears its own)
        g_img, g_feat = _C.render.mesh.dibr_rasterization_backward_fused(
            grad_feats.contiguous(), grad_soft_mask.contiguous(), face_idx, weights, soft_mask, ctx.saved_tensors[5:],
            face_vertices_image, face_features, sigmainv, knum, multiplier, eps,
            need_feature_grad=ctx.needs_input_grad[4], zeroed_grad_image=zeroed)
        return None, None, None, g_img, g_feat, None, None, None, None, None, None


def dibr_rasterization(height, width, face_vertices_z, face_vertices_image, face_features, face_normals_z,
                       sigmainv=7000, boxlen=0.02, knum=30, multiplier=None, eps=None, rast_backend='cuda'):
    r"""DIB-R rasterization: :func:`rasterize` restricted to front faces (``face_normals_z >= 0``) followed
    by :func:`dibr_soft_mask` over ALL faces (reference: kaolin/render/mesh/dibr.py:119-209).

    Returns:
        (torch.Tensor or tuple, torch.Tensor, torch.LongTensor): features (B, H, W, D), soft mask (B, H, W),
        face index (B, H, W).
    """
    _multiplier = 1000. if multiplier is None else multiplier
    if rast_backend != 'cuda' or not face_vertices_image.is_cuda:
        interpolated_features, face_idx = rasterize(height, width, face_vertices_z, face_vertices_image,
                                                    face_features, face_normals_z >= 0., multiplier, eps, rast_backend)
        soft_mask = dibr_soft_mask(face_vertices_image, face_idx, sigmainv, boxlen, knum, _multiplier)
        return interpolated_features, soft_mask, face_idx
    is_list = isinstance(face_features, (list, tuple))
    _features = torch.cat(face_features, dim=-1) if is_list else face_features
    # rasterize()'s defaults (multiplier 1000, eps 1e-8) and dibr_soft_mask's multiplier (1000.) coincide numerically
    front = face_normals_z if face_normals_z.dtype == face_vertices_z.dtype else face_normals_z >= 0.
    image_features, soft_mask, face_idx = DibrRasterizationCuda.apply(
        height, width, face_vertices_z, face_vertices_image, _features, front.detach(), sigmainv, boxlen, knum,
        _multiplier, 1e-8 if eps is None else eps)
    if is_list:
        out, cur = [], 0
        for f in face_features:
            out.append(image_features[..., cur:cur + f.shape[-1]])
            cur += f.shape[-1]
        image_features = tuple(out)
    return image_features, soft_mask, face_idx
