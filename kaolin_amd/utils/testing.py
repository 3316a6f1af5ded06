"""Test/bench support: dtype lists (the three constants the reference's hot-path tests use,
kaolin/utils/testing.py:35-49), seeding, and synthetic DIB-R scenes (geodesic spheres + cameras as
specified in SURVEY.md section 8(d))."""
import functools
import math
import random

import numpy as np
import torch

from ..ops.mesh import index_vertices_by_faces, face_normals
from ..render import camera as _cam

FLOAT_DTYPES = [torch.half, torch.float, torch.double]
CUDA_FLOAT_TYPES = [('cuda', d) for d in FLOAT_DTYPES]
FLOAT_TYPES = CUDA_FLOAT_TYPES + [('cpu', torch.float), ('cpu', torch.double)]


def with_seed(torch_seed=0, numpy_seed=0, random_seed=0):
    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            torch.manual_seed(torch_seed)
            np.random.seed(numpy_seed)
            random.seed(random_seed)
            return fn(*a, **k)
        return wrapped
    return deco


def check_allclose(a, b, rtol=1e-5, atol=1e-8):
    if not torch.allclose(a, b, rtol=rtol, atol=atol):
        raise AssertionError(f'max abs diff {(a - b).abs().max().item()}')


def check_tensor(tensor, shape=None, dtype=None, device=None, throw=True):
    """True when `tensor` has the given shape (None entries match anything), dtype and device; otherwise raises
    (ValueError for the shape, TypeError for dtype / device) or, with throw=False, returns False
    (behaviour of kaolin/utils/testing.py:73-111)."""
    problem = None
    if shape is not None:
        if len(shape) != tensor.ndim:
            problem = ValueError(f"tensor have {tensor.ndim} ndim, should have {len(shape)}")
        elif any(want is not None and have != want for have, want in zip(tensor.shape, shape)):
            problem = ValueError(f"tensor shape is {tensor.shape}, should be {shape}")
    if problem is None and dtype is not None and tensor.dtype != dtype:
        problem = TypeError(f"tensor dtype is {tensor.dtype}, should be {dtype}")
    if problem is None and device is not None:
        want = torch.device(device)
        if want.type != tensor.device.type or (want.index is not None and want.index != tensor.device.index):
            problem = TypeError(f"tensor device is {tensor.device}, should be {want}")
    if problem is not None and throw:
        raise problem
    return problem is None


def geodesic_sphere(frequency, radius=0.5):
    """Class-I geodesic icosphere: every icosahedron face split into frequency^2 triangles and pushed
    to the sphere: 20*f^2 faces, 10*f^2+2 vertices (f=16 -> 5120 faces, f=50 -> 50000 faces).
    Returns (vertices (V,3) float64, faces (F,3) int64) with outward counter-clockwise winding."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    base_v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                       [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    base_f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
              (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
              (8, 6, 7), (9, 8, 1)]
    n = int(frequency)
    key_to_id, verts, faces = {}, [], []

    def vid(pt):
        k = tuple(np.round(pt * 1e9).astype(np.int64))
        if k not in key_to_id:
            key_to_id[k] = len(verts)
            verts.append(pt)
        return key_to_id[k]

    for (a, b, c) in base_f:
        A, Bv, C = base_v[a], base_v[b], base_v[c]
        grid = {}
        for i in range(n + 1):
            for j in range(n + 1 - i):
                grid[(i, j)] = vid((A * (n - i - j) + Bv * i + C * j) / n)
        for i in range(n):
            for j in range(n - i):
                faces.append((grid[(i, j)], grid[(i + 1, j)], grid[(i, j + 1)]))
                if i + j < n - 1:
                    faces.append((grid[(i + 1, j)], grid[(i + 1, j + 1)], grid[(i, j + 1)]))
    v = np.stack(verts)
    v = v / np.linalg.norm(v, axis=1, keepdims=True) * radius
    return torch.from_numpy(v), torch.tensor(faces, dtype=torch.long)


def fibonacci_cameras(num_views, distance=2.5, dtype=torch.float):
    """Camera positions on a Fibonacci sphere of the given radius."""
    i = torch.arange(num_views, dtype=torch.float64) + 0.5
    phi = torch.acos(1 - 2 * i / num_views)
    theta = math.pi * (1 + 5 ** 0.5) * i
    pos = torch.stack([torch.cos(theta) * torch.sin(phi), torch.cos(phi), torch.sin(theta) * torch.sin(phi)], dim=1)
    return (pos * distance).to(dtype)


def project_mesh(vertices, faces, camera_position, fov=math.pi / 4, up=(0., 1., 0.)):
    """The tutorial's prepare_vertices in legacy-camera form (kaolin/render/mesh/utils.py:128-175):
    returns face_vertices_z (B,F,3), face_vertices_image (B,F,3,2), face_normals_z (B,F)."""
    B = camera_position.shape[0]
    dtype, device = vertices.dtype, vertices.device
    look_at = torch.zeros((B, 3), dtype=dtype, device=device)
    upv = torch.tensor([up], dtype=dtype, device=device).repeat(B, 1)
    # avoid a degenerate frame when a camera sits on the up axis
    bad = (torch.cross(_cam._unit(look_at - camera_position), upv, dim=1).norm(dim=1) < 1e-3)
    upv[bad] = torch.tensor([1., 0., 0.], dtype=dtype, device=device)
    rot, trans = _cam.generate_rotate_translate_matrices(camera_position, look_at, upv)
    v_cam = _cam.rotate_translate_points(vertices.unsqueeze(0).expand(B, -1, -1), rot, trans)
    proj = _cam.generate_perspective_projection(fov, dtype=dtype).to(device)
    v_img = _cam.perspective_camera(v_cam, proj)
    fv_cam = index_vertices_by_faces(v_cam, faces)
    fv_img = index_vertices_by_faces(v_img, faces)
    normals_z = face_normals(fv_cam, unit=True)[..., 2]
    return fv_cam[..., 2].contiguous(), fv_img.contiguous(), normals_z.contiguous()


def sphere_scene(level=16, num_views=1, device='cpu', dtype=torch.float, seed=0, distance=2.5):
    """Synthetic DIB-R inputs: geodesic sphere (20*level^2 faces, radius 0.5), cameras at `distance`
    looking at the origin (view 0 = (0,0,distance) as config C2, more views on a Fibonacci sphere),
    features = [uv-like rand (B,F,3,2), ones (B,F,3,1)] (D = 3).
    Returns (face_vertices_z, face_vertices_image, [feat_uv, feat_ones], face_normals_z)."""
    v, f = geodesic_sphere(level)
    v = v.to(dtype)
    if num_views == 1:
        cams = torch.tensor([[0., 0., distance]], dtype=dtype)
    else:
        cams = fibonacci_cameras(num_views, distance, dtype)
    fz, fimg, nz = project_mesh(v, f, cams)
    g = torch.Generator().manual_seed(seed)
    uv = torch.rand((1, f.shape[0], 3, 2), generator=g, dtype=torch.float).to(dtype).repeat(num_views, 1, 1, 1)
    ones = torch.ones((num_views, f.shape[0], 3, 1), dtype=dtype)
    return fz.to(device), fimg.to(device), [uv.to(device), ones.to(device)], nz.to(device)
