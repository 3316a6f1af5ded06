"""``rasterize`` -- DIB-R differentiable rasterization (API mirror of
kaolin/render/mesh/rasterization.py:226-492).  Coverage / depth / barycentric interpolation and the
backward scatter run in hand-written HIP kernels (kaolin_amd/csrc/rasterize.hip)."""
import torch

from ... import _C

__all__ = ['rasterize']


class RasterizeCuda(torch.autograd.Function):
    """autograd shim with the contract of the reference's RasterizeCuda (rasterization.py:226-371): only valid
    faces are rasterized, coordinates are scaled by ``multiplier``, the returned face index is mesh-relative with -1
    for empty pixels; backward returns gradients for face_vertices_image and face_features only (none for
    face_vertices_z).  The reference does the packing / scaling / bounding boxes with ~15 torch kernels and a host
    sync (torch.where); here they are folded into the bin kernel (``_C.render.mesh.rasterize_forward_fused``).  The
    reference-contract operator ``_C.render.mesh.packed_rasterize_forward_cuda`` is kept and gives the same result
    (``_packed_forward`` below spells the reference's glue around it; tests compare the two)."""

    @staticmethod
    def forward(ctx, height, width, face_vertices_z, face_vertices_image, face_features, valid_faces,
                multiplier, eps):
        face_features = face_features.contiguous()
        face_vertices_image = face_vertices_image.contiguous()
        valid = None if valid_faces is None else valid_faces.contiguous()
        interpolated_features, face_idx, output_weights = _C.render.mesh.rasterize_forward_fused(
            height, width, face_vertices_z.contiguous(), face_vertices_image, face_features, valid, multiplier, eps)
        ctx.save_for_backward(interpolated_features, face_idx, output_weights, face_vertices_image, face_features)
        ctx.mark_non_differentiable(face_idx)
        # no zero tensors for the gradients of outputs nobody differentiates (the index output alone is B*H*W*8 bytes)
        ctx.set_materialize_grads(False)
        ctx.eps = eps
        return interpolated_features, face_idx

    @staticmethod
    def backward(ctx, grad_interpolated_features, grad_face_idx):
        if grad_interpolated_features is None:
            return (None,) * 8
        interpolated_features, face_idx, output_weights, face_vertices_image, face_features = ctx.saved_tensors
        grad_img, grad_feat = _C.render.mesh.rasterize_backward_cuda(
            grad_interpolated_features.contiguous(), interpolated_features, face_idx, output_weights,
            face_vertices_image, face_features, ctx.eps, need_feature_grad=ctx.needs_input_grad[4])
        return None, None, None, grad_img, grad_feat, None, None, None


def _packed_forward(height, width, face_vertices_z, face_vertices_image, face_features, valid_faces, multiplier, eps):
    """The reference's own glue (rasterization.py:282-346) around the reference-contract operator
    ``packed_rasterize_forward_cuda``: pack valid faces, scale, per-face boxes, map packed indices back.
    Returns (features, face_idx, weights).  Not used by ``rasterize`` (see RasterizeCuda); kept as the
    contract-level path and exercised by the parity tests."""
    batch_size, num_faces = face_vertices_z.shape[0], face_vertices_z.shape[1]
    feat_dim = face_features.shape[-1]
    device = face_vertices_z.device
    face_features = face_features.contiguous()
    face_vertices_image = face_vertices_image.contiguous()
    if valid_faces is None:
        faces_of = None
        packed_img = face_vertices_image.reshape(batch_size * num_faces, 3, 2)
        packed_z = face_vertices_z.reshape(batch_size * num_faces, 3)
        packed_feat = face_features.reshape(batch_size * num_faces, 3, feat_dim)
        first_idx = torch.arange(batch_size + 1, dtype=torch.long, device=device) * num_faces
    else:
        mesh_of, faces_of = torch.where(valid_faces)
        packed_img = face_vertices_image[mesh_of, faces_of]
        packed_z = face_vertices_z[mesh_of, faces_of]
        packed_feat = face_features[mesh_of, faces_of]
        first_idx = torch.zeros(batch_size + 1, dtype=torch.long, device=device)
        torch.cumsum(valid_faces.reshape(batch_size, -1).sum(dim=1), dim=0, out=first_idx[1:])
    packed_img = packed_img * multiplier
    bboxes = torch.cat((packed_img.min(dim=1)[0], packed_img.max(dim=1)[0]), dim=1)
    interpolated_features, selected, output_weights = _C.render.mesh.packed_rasterize_forward_cuda(
        height, width, packed_z.contiguous(), packed_img.contiguous(), bboxes.contiguous(),
        packed_feat.contiguous(), first_idx.contiguous(), multiplier, eps)
    if faces_of is None:
        face_idx = selected
    else:
        lookup = (selected + first_idx[:-1].reshape(-1, 1, 1)).reshape(-1)
        if faces_of.numel() > 0:
            face_idx = faces_of[lookup.clamp_(min=0, max=faces_of.numel() - 1)].reshape(selected.shape).contiguous()
        else:
            face_idx = torch.full_like(selected, -1)
        face_idx[selected == -1] = -1
    return interpolated_features, face_idx, output_weights


def rasterize(height, width, face_vertices_z, face_vertices_image, face_features, valid_faces=None,
              multiplier=None, eps=None, backend='cuda'):
    r"""Fully differentiable rasterization of triangle meshes with per-vertex per-face features into
    feature "images" (reference: kaolin/render/mesh/rasterization.py:373-492).

    Args:
        height, width (int): size of the rendered images.
        face_vertices_z (torch.FloatTensor): depth of the face vertices in camera space, (B, F, 3).
        face_vertices_image (torch.FloatTensor): 2D vertex positions in [-1, 1] NDC, (B, F, 3, 2).
        face_features (torch.FloatTensor or list): (B, F, 3, D) or a list of such tensors.
        valid_faces (torch.BoolTensor): (B, F) mask of faces to rasterize. Default: all.
        multiplier (int): coordinates are enlarged by this factor internally. Default: 1000.
        eps (float): epsilon added to the barycentric normaliser. Default: 1e-8.
        backend (str): only ``'cuda'`` (the HIP kernels of this package); the nvdiffrast backends of the
            reference are not available on this platform.

    Returns:
        (torch.FloatTensor or tuple, torch.LongTensor): features (B, H, W, D) (split like the input list)
        and the rendered face index (B, H, W), -1 where no face covers the pixel.
    """
    if multiplier is None:
        multiplier = 1000
    if eps is None:
        eps = 1e-8
    _features = torch.cat(face_features, dim=-1) if isinstance(face_features, (list, tuple)) else face_features
    if backend == 'cuda':
        image_features, face_idx = RasterizeCuda.apply(height, width, face_vertices_z, face_vertices_image,
                                                       _features, valid_faces, multiplier, eps)
    elif backend in ('nvdiffrast', 'nvdiffrast_fwd'):
        raise ValueError(f'backend "{backend}" is not available: nvdiffrast is a CUDA-only package')
    else:
        raise ValueError(f'"{backend}" is not a valid backend, ',
                         'valid choices are ["cuda", "nvdiffrast", "nvdiffrast_fwd"]')
    if isinstance(face_features, (list, tuple)):
        out, cur = [], 0
        for f in face_features:
            out.append(image_features[..., cur:cur + f.shape[-1]])
            cur += f.shape[-1]
        image_features = tuple(out)
    return image_features, face_idx
