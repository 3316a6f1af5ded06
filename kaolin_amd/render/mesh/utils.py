"""Steps either side of DIB-R in the training loop (pure torch): ``prepare_vertices`` and
``texture_mapping`` (behaviour of kaolin/render/mesh/utils.py:23-76,128-175)."""
import torch

from .. import camera
from ...ops import mesh as _mesh

__all__ = ['prepare_vertices', 'texture_mapping']


def prepare_vertices(vertices, faces, camera_proj, camera_rot=None, camera_trans=None, camera_transform=None):
    """World-space vertices (B, V, 3) -> (face_vertices_camera (B,F,3,3), face_vertices_image (B,F,3,2),
    unit face_normals (B,F,3)).  Either (camera_rot, camera_trans) or a (B, 4, 3) camera_transform."""
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


def texture_mapping(texture_coordinates, texture_maps, mode='nearest'):
    """Samples texture_maps (B, C, h', w') at OpenGL-style coordinates in [0, 1] (y up), given densely
    (B, h, w, 2) or sparsely (B, N, 2); returns (B, h, w, C) or (B, N, C)."""
    B, C = texture_coordinates.shape[0], texture_maps.shape[1]
    uv = torch.clamp(texture_coordinates.reshape(B, -1, 1, 2), 0., 1.) * 2 - 1
    grid = torch.stack([uv[..., 0], -uv[..., 1]], dim=-1)
    out = torch.nn.functional.grid_sample(texture_maps, grid, mode=mode, align_corners=False, padding_mode='border')
    return out.permute(0, 2, 3, 1).reshape(B, *texture_coordinates.shape[1:-1], C)
