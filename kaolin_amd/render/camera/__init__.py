"""Legacy camera helpers (pure torch) used to build DIB-R inputs and test fixtures.

Behaviour follows kaolin/render/camera/legacy.py:22-159 (the tutorial's `prepare_vertices` path):
    P_cam = R (P_world - t),  image = (P_cam * proj)[..., :2] / (P_cam * proj)[..., 2:3].
The `Camera` class hierarchy of the reference is out of scope.
"""
from math import tan

import torch

__all__ = ['generate_perspective_projection', 'generate_rotate_translate_matrices', 'rotate_translate_points',
           'perspective_camera']

_TINY = 1e-10


def generate_perspective_projection(fovyangle, ratio=1.0, dtype=torch.float):
    """(3, 1) projection vector [1/(ratio*tan(fovy/2)), 1/tan(fovy/2), -1] (legacy.py:142-159)."""
    t = tan(fovyangle / 2.0)
    return torch.tensor([[1.0 / (ratio * t)], [1.0 / t], [-1]], dtype=dtype)


def _unit(v):
    return v / (v.norm(dim=1, keepdim=True) + _TINY)


def generate_rotate_translate_matrices(camera_position, look_at, camera_up_direction):
    """Rotation (B, 3, 3) with rows (x, y, -z) of the camera frame and translation = camera position
    (legacy.py:40-83)."""
    fwd = _unit(look_at - camera_position)
    up = camera_up_direction
    if up.shape[0] < fwd.shape[0]:
        up = up.repeat(fwd.shape[0], 1)
    elif up.shape[0] > fwd.shape[0]:
        fwd = fwd.repeat(up.shape[0], 1)
    right = _unit(torch.cross(fwd, up, dim=1))
    true_up = _unit(torch.cross(right, fwd, dim=1))
    return torch.stack([right, true_up, -fwd], dim=1), camera_position


def rotate_translate_points(points, camera_rot, camera_trans):
    """R (P - t) for points (B, N, 3) (legacy.py:22-38)."""
    return torch.matmul(points - camera_trans.view(-1, 1, 3), camera_rot.permute(0, 2, 1))


def perspective_camera(points, camera_proj):
    """Perspective divide of camera-space points (B, N, 3) -> (B, N, 2) (legacy.py:123-140)."""
    proj = points * camera_proj.view(-1, 1, 3)
    return proj[:, :, :2] / proj[:, :, 2:3]
