"""Steps either side of DIB-R in the training loop (pure torch): ``prepare_vertices`` and
``texture_mapping`` (behaviour of kaolin/render/mesh/utils.py:23-76,128-175)."""
import torch

from .. import camera
from ... import _C
from ...ops import mesh as _mesh

__all__ = ['prepare_vertices', 'texture_mapping']


class _PrepareVerticesCuda(torch.autograd.Function):
    """One kernel each way for the torch op chain of ``prepare_vertices`` (fused path, gradients w.r.t. vertices only)."""

    @staticmethod
    def forward(ctx, vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform):
        out = _C.render.mesh.prepare_vertices_forward_fused(vertices, faces, camera_proj, camera_rot, camera_trans,
                                                            camera_transform)
        ctx.save_for_backward(vertices, faces, camera_proj,
                              *(t for t in (camera_rot, camera_trans, camera_transform) if t is not None))
        ctx.has_transform = camera_transform is not None
        # an output nobody differentiates (typically the camera-space vertices and the normals: only their z feeds the
        # rasterizer, without gradient) arrives as None, not as a zero tensor the backward kernel would have to read
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, grad_cam, grad_img, grad_nrm):
        vertices, faces, camera_proj = ctx.saved_tensors[:3]
        rest = ctx.saved_tensors[3:]
        rot, trans, tf = (None, None, rest[0]) if ctx.has_transform else (rest[0], rest[1], None)
        if grad_cam is None and grad_img is None and grad_nrm is None:
            return (None,) * 6
        g = _C.render.mesh.prepare_vertices_backward_fused(vertices, faces, camera_proj, rot, trans, tf,
                                                           grad_cam, grad_img, grad_nrm)
        if g.shape[0] != vertices.shape[0]:      # one shared (1, V, 3) mesh: sum the per-view gradients
            g = g.sum(dim=0, keepdim=True)
        return g, None, None, None, None, None


def _fusable(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform):
    """The fused kernels read one camera per view at fixed strides and ONE projection: anything else (a batched
    projection, a broadcast translation, a single camera for batched vertices, cameras that require grad, CPU tensors,
    other dtypes) takes the torch op chain, which broadcasts like the reference's."""
    cams = [t for t in (camera_proj, camera_rot, camera_trans, camera_transform) if t is not None]
    if not (vertices.is_cuda and vertices.dtype in (torch.float32, torch.float64) and vertices.dim() == 3 and
            vertices.shape[-1] == 3 and faces.is_cuda and faces.dtype == torch.long and faces.dim() == 2 and
            faces.shape[1] == 3 and all(t.is_cuda for t in cams) and not any(t.requires_grad for t in cams)):
        return False
    if camera_proj.numel() != 3:
        return False
    if camera_transform is not None:
        batch = camera_transform.shape[0]
        ok = tuple(camera_transform.shape) == (batch, 4, 3)
    else:
        batch = camera_rot.shape[0]
        ok = tuple(camera_rot.shape) == (batch, 3, 3) and camera_trans.numel() == batch * 3 and camera_trans.shape[0] == batch
    return ok and batch > 0 and vertices.shape[0] in (1, batch) and _C.render.mesh.faces_in_range(faces, vertices.shape[1])


def prepare_vertices(vertices, faces, camera_proj, camera_rot=None, camera_trans=None, camera_transform=None):
    """World-space vertices (B, V, 3) -> (face_vertices_camera (B,F,3,3), face_vertices_image (B,F,3,2),
    unit face_normals (B,F,3)).  Either (camera_rot, camera_trans) or a (B, 4, 3) camera_transform.
    On the GPU (float/double, no gradient required for the camera tensors) this is one fused HIP kernel each way;
    otherwise the torch op chain below, which is also the definition the fused path is tested against."""
    if camera_transform is None:
        if camera_rot is None or camera_trans is None:
            raise AssertionError('camera_transform or camera_trans and camera_rot must be defined')
    elif camera_rot is not None or camera_trans is not None:
        raise AssertionError('camera_trans and camera_rot must be None when camera_transform is defined')
    if (faces.is_cuda and faces.dtype == torch.long and vertices.dim() == 3 and
            not _C.render.mesh.faces_in_range(faces, vertices.shape[1])):
        # the reference's index_select trips a device-side assert here (on ROCm that aborts the process): raise instead
        raise IndexError('prepare_vertices: faces hold an index outside [0, num_vertices)')
    if _fusable(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform):
        return _PrepareVerticesCuda.apply(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform)
    return _prepare_vertices_torch(vertices, faces, camera_proj, camera_rot, camera_trans, camera_transform)


def _prepare_vertices_torch(vertices, faces, camera_proj, camera_rot=None, camera_trans=None, camera_transform=None):
    """The reference's op chain (kaolin/render/mesh/utils.py:156-175)."""
    if camera_transform is None:
        if camera_rot is None or camera_trans is None:
            raise AssertionError('camera_transform or camera_trans and camera_rot must be defined')
        v_cam = camera.rotate_translate_points(vertices, camera_rot, camera_trans)
    else:
        if camera_rot is not None or camera_trans is not None:
            raise AssertionError('camera_trans and camera_rot must be None when camera_transform is defined')
        v_cam = torch.nn.functional.pad(vertices, (0, 1), mode='constant', value=1.) @ camera_transform
    v_img = camera.perspective_camera(v_cam, camera_proj)
    fv_cam = _mesh.index_vertices_by_faces(v_cam, faces)
    fv_img = _mesh.index_vertices_by_faces(v_img, faces)
    return fv_cam, fv_img, _mesh.face_normals(fv_cam, unit=True)


class _TextureMappingCuda(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uv, texture_maps, bilinear):
        uv, texture_maps = uv.contiguous(), texture_maps.contiguous()
        out = _C.render.mesh.texture_mapping_forward_fused(uv, texture_maps, bilinear)
        ctx.save_for_backward(uv, texture_maps)
        ctx.bilinear = bilinear
        return out

    @staticmethod
    def backward(ctx, grad_out):
        uv, texture_maps = ctx.saved_tensors
        g_tex, g_uv = _C.render.mesh.texture_mapping_backward_fused(
            uv, texture_maps, grad_out.contiguous(), ctx.bilinear, ctx.needs_input_grad[1], ctx.needs_input_grad[0])
        return g_uv, g_tex, None


def _texture_mapping_torch(texture_coordinates, texture_maps, mode='nearest'):
    """The reference's op chain (kaolin/render/mesh/utils.py:58-76)."""
    B, C = texture_coordinates.shape[0], texture_maps.shape[1]
    uv = torch.clamp(texture_coordinates.reshape(B, -1, 1, 2), 0., 1.) * 2 - 1
    grid = torch.stack([uv[..., 0], -uv[..., 1]], dim=-1)
    out = torch.nn.functional.grid_sample(texture_maps, grid, mode=mode, align_corners=False, padding_mode='border')
    return out.permute(0, 2, 3, 1).reshape(B, *texture_coordinates.shape[1:-1], C)


def texture_mapping(texture_coordinates, texture_maps, mode='nearest'):
    """Samples texture_maps (B, C, h', w') at OpenGL-style coordinates in [0, 1] (y up), given densely
    (B, h, w, 2) or sparsely (B, N, 2); returns (B, h, w, C) or (B, N, C).  On the GPU (float / double, ``nearest`` or
    ``bilinear``) this is one fused HIP gather each way; otherwise the torch chain, which also defines it."""
    if (mode in ('nearest', 'bilinear') and texture_coordinates.is_cuda and texture_maps.is_cuda and
            texture_coordinates.dtype == texture_maps.dtype and texture_coordinates.dtype in (torch.float32, torch.float64) and
            texture_maps.dim() == 4 and texture_coordinates.shape[-1] == 2 and texture_coordinates.numel() > 0 and
            texture_maps.shape[0] == texture_coordinates.shape[0] and texture_maps[0].numel() > 0 and
            texture_coordinates.device == texture_maps.device and
            texture_maps.shape[0] <= 65535):      # (the kernels put the batch on a grid dimension)
        B, C = texture_coordinates.shape[0], texture_maps.shape[1]
        out = _TextureMappingCuda.apply(texture_coordinates.reshape(B, -1, 2), texture_maps, mode == 'bilinear')
        return out.reshape(B, *texture_coordinates.shape[1:-1], C)
    return _texture_mapping_torch(texture_coordinates, texture_maps, mode)
