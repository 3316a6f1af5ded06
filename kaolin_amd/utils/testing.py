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


def elementwise_mismatch(a, b, tol=1e-5, term_abs_sum=None, sum_ulps=64.0):
    """Element-wise comparison of a float result `a` with its reference `b` (the north star's "within 1e-5 relative"):
        |a - b| <= tol * |b| + tol * median(|b| over the elements where b != 0)   [+ sum_ulps * eps(a.dtype) * term_abs_sum]
    -- relative to EACH element, with a floor of `tol` times the typical magnitude so that elements which are tiny because
    their terms cancel are not held to a relative bound their own reference does not meet (a tolerance scaled by the LARGEST
    element, which these tests used through round 3, lets a small entry be off by orders of magnitude).
    `term_abs_sum` (optional, same shape): for an element that is a SUM of many terms accumulated in `a`'s precision in an
    unspecified order (the reference adds them with float atomicAdd; so do the kernels; the oracle sums in double), the sum of
    the terms' magnitudes -- the rounding error of any such accumulation scales with it, not with the (possibly cancelling)
    result: a face of the knot scene that spans 200 pixels collects tens of thousands of terms of both signs.
    Returns None when every element passes, else a message naming the worst element."""
    eps = float(torch.finfo(a.dtype).eps) if a.is_floating_point() else 0.0
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if a.shape != b.shape:
        return f'shape {tuple(a.shape)} vs {tuple(b.shape)}'
    nz = b[b != 0].abs()
    floor = float(nz.median()) if nz.numel() else 0.0
    bound = tol * b.abs() + tol * floor
    if term_abs_sum is not None:
        plain = bound
        bound = bound + sum_ulps * eps * term_abs_sum.detach().double().cpu()
        # how many elements pass only thanks to the accumulation slack (it must not grow silently): kept in
        # `last_slack_use` and printed by the full-size tests (pytest -s / the captured stdout of a failure)
        needed = ((a - b).abs() > plain) & ((a - b).abs() <= bound)
        elementwise_mismatch.last_slack_use = (int(needed.sum()), int(a.numel()))
    bad = ~((a - b).abs() <= bound)                      # (NaN anywhere fails)
    both_nan = torch.isnan(a) & torch.isnan(b)
    bad &= ~both_nan
    if not bool(bad.any()):
        return None
    excess = torch.where(bad, (a - b).abs() / bound.clamp(min=1e-300), torch.zeros_like(a))
    i = int(excess.reshape(-1).argmax())
    extra = '' if term_abs_sum is None else f', sum of its terms\' magnitudes {float(term_abs_sum.reshape(-1)[i]):.4e}'
    return (f'{int(bad.sum())} of {a.numel()} elements outside |a-b| <= {tol:g}|b| + {tol:g}*{floor:.3e}'
            f'{"" if term_abs_sum is None else f" + {sum_ulps:g} eps sum|terms|"}; worst at flat index {i}: '
            f'{float(a.reshape(-1)[i])!r} vs {float(b.reshape(-1)[i])!r} ({float(excess.reshape(-1)[i]):.2f}x the bound{extra})')


elementwise_mismatch.last_slack_use = (0, 0)


def elementwise_close(a, b, tol=1e-5):
    return elementwise_mismatch(a, b, tol) is None


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


def knot_mesh(nu=440, nv=48):
    """A NON-CONVEX test mesh of ~50 000 triangles (the second scene of the parity tests and of bench.py --scene knot;
    the convex sphere of config C4 has depth complexity 2, uniform tiny triangles and nothing off screen):
      * a trefoil-knot tube (2 nu nv triangles): strands cross in front of each other -- 4 surface layers at a crossing;
      * two nested spheres in the knot's central hole (2 880 + 3 920 triangles): with a strand in front or behind, 6-8 layers;
      * a coarse bowl below everything (an inward-facing cap of a frequency-6 icosphere of radius 1.4, ~170 triangles
        100-200 pixels across at 1024^2, partly beyond the image border): image-sized faces among tiny ones --
        the tile lists' big-face path, boxes clipped by the image border, a silhouette against geometry instead of background.
    Returns (vertices (V, 3) float64, faces (F, 3) int64); outward winding for the tube and the spheres, inward for the bowl."""
    t = np.linspace(0.0, 2.0 * np.pi, nu, endpoint=False)
    c = 0.17 * np.stack([np.sin(t) + 2.0 * np.sin(2.0 * t), np.cos(t) - 2.0 * np.cos(2.0 * t), -np.sin(3.0 * t)], axis=1)
    d = 0.17 * np.stack([np.cos(t) + 4.0 * np.cos(2.0 * t), -np.sin(t) + 4.0 * np.sin(2.0 * t), -3.0 * np.cos(3.0 * t)], axis=1)
    tangent = d / np.linalg.norm(d, axis=1, keepdims=True)
    # a frame without twist jumps: project a fixed axis, fall back where it is nearly parallel to the tangent
    ref = np.tile(np.array([[0.0, 0.0, 1.0]]), (nu, 1))
    nearly = np.abs((tangent * ref).sum(1)) > 0.95
    ref[nearly] = np.array([1.0, 0.0, 0.0])
    n1 = np.cross(tangent, ref)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tangent, n1)
    a = np.linspace(0.0, 2.0 * np.pi, nv, endpoint=False)
    ring = np.cos(a)[None, :, None] * n1[:, None, :] + np.sin(a)[None, :, None] * n2[:, None, :]
    tube_v = (c[:, None, :] + 0.06 * ring).reshape(-1, 3)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    v00, v10 = i * nv + j, ((i + 1) % nu) * nv + j
    v01, v11 = i * nv + (j + 1) % nv, ((i + 1) % nu) * nv + (j + 1) % nv
    tube_f = np.concatenate([np.stack([v00, v10, v11], -1).reshape(-1, 3), np.stack([v00, v11, v01], -1).reshape(-1, 3)])
    # orient the tube outwards (the frame's handedness decides; check one face against its ring direction)
    f0 = tube_f[0]
    nrm = np.cross(tube_v[f0[1]] - tube_v[f0[0]], tube_v[f0[2]] - tube_v[f0[0]])
    if (nrm * ring[0, 0]).sum() < 0:
        tube_f = tube_f[:, [0, 2, 1]]
    parts_v, parts_f, base = [tube_v], [tube_f], tube_v.shape[0]
    for freq, radius in ((12, 0.10), (14, 0.13)):
        sv, sf = geodesic_sphere(freq, radius)
        parts_v.append(sv.numpy())
        parts_f.append(sf.numpy() + base)
        base += sv.shape[0]
    bv, bf = geodesic_sphere(6, 1.4)
    bv, bf = bv.numpy(), bf.numpy()
    keep = bv[bf].mean(axis=1)[:, 1] < -0.75
    parts_v.append(bv)
    parts_f.append(bf[keep][:, [0, 2, 1]] + base)          # (inward: the far side of the bowl faces the camera)
    return torch.from_numpy(np.concatenate(parts_v)), torch.from_numpy(np.concatenate(parts_f)).long()


def mesh_scene(vertices, faces, num_views=1, device='cpu', dtype=torch.float, seed=0, distance=2.5):
    """DIB-R inputs for any mesh: cameras as in :func:`sphere_scene`, features = [uv-like rand (B,F,3,2), ones (B,F,3,1)].
    Returns (face_vertices_z, face_vertices_image, [feat_uv, feat_ones], face_normals_z)."""
    v = vertices.to(dtype)
    cams = torch.tensor([[0., 0., distance]], dtype=dtype) if num_views == 1 else fibonacci_cameras(num_views, distance, dtype)
    fz, fimg, nz = project_mesh(v, faces, cams)
    g = torch.Generator().manual_seed(seed)
    uv = torch.rand((1, faces.shape[0], 3, 2), generator=g, dtype=torch.float).to(dtype).repeat(num_views, 1, 1, 1)
    ones = torch.ones((num_views, faces.shape[0], 3, 1), dtype=dtype)
    return fz.to(device), fimg.to(device), [uv.to(device), ones.to(device)], nz.to(device)


def knot_scene(num_views=8, device='cpu', dtype=torch.float, seed=0, distance=2.5):
    """:func:`knot_mesh` seen from the cameras of config C4 (a Fibonacci sphere of radius `distance` looking at the origin)."""
    v, f = knot_mesh()
    return mesh_scene(v, f, num_views, device, dtype, seed, distance)


def scene_mesh(name, frequency=50):
    """The meshes bench.py and the full-size parity tests render: 'sphere' (config C4: geodesic sphere, 20 f^2 triangles),
    'knot' (:func:`knot_mesh`) or 'knot_shuffled' (the knot's faces in a random order: neighbours in the list are not neighbours
    on the screen -- the reference's loops over all faces do not care, tile lists do).  -> (vertices float64, faces int64)"""
    if name == 'sphere':
        return geodesic_sphere(frequency)
    if name == 'knot':
        return knot_mesh()
    if name == 'knot_shuffled':
        v, f = knot_mesh()
        return v, f[torch.randperm(f.shape[0], generator=torch.Generator().manual_seed(1))].contiguous()
    raise ValueError(f'unknown scene {name!r} (sphere | knot | knot_shuffled)')


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
