"""Mesh helpers on the DIB-R path (pure torch): per-face gathers and normals, plus the vertex-set
subdivision the voxelizer's definition rests on."""
import torch

__all__ = ['index_vertices_by_faces', 'face_normals', 'check_sign']


def index_vertices_by_faces(vertices_features, faces):
    """(B, V, C) per-vertex features + (F, k) faces -> (B, F, k, C)
    (behaviour of kaolin/ops/mesh/mesh.py:54-75)."""
    if vertices_features.ndim != 3:
        raise AssertionError('vertices_features must have 3 dimensions of shape (batch_size, num_points, knum)')
    if faces.ndim != 2:
        raise AssertionError('faces must have 2 dimensions of shape (num_faces, num_vertices)')
    B, _, C = vertices_features.shape
    F, k = faces.shape
    flat = faces.reshape(-1)
    return vertices_features.index_select(1, flat).reshape(B, F, k, C)


def face_normals(face_vertices, unit=False):
    """(B, F, 3, 3) -> (B, F, 3) normals, (v1 - v0) x (v2 - v0); optional normalisation with a 1e-10
    guard (behaviour of kaolin/ops/mesh/trianglemesh.py:314-337)."""
    if face_vertices.shape[-2] != 3:
        raise NotImplementedError('face_normals is only implemented for triangle meshes')
    e0 = face_vertices[:, :, 1] - face_vertices[:, :, 0]
    e1 = face_vertices[:, :, 2] - face_vertices[:, :, 0]
    n = torch.cross(e0, e1, dim=2)
    if unit:
        n = n / (n.norm(dim=2, keepdim=True) + 1e-10)
    return n


def _unbatched_check_sign_cuda(verts, faces, points):
    """kaolin/ops/mesh/check_sign.py:45-54."""
    from ... import _C
    corners = [verts[faces[:, k]].contiguous() for k in range(3)]
    crossings = _C.ops.unbatched_mesh_intersection_cuda(points.contiguous(), *corners)
    return crossings % 2 == 1.


def check_sign(verts, faces, points, hash_resolution=512):
    r"""Checks if a set of points is contained inside a watertight triangle mesh: shoots a ray from each point along +x
    and uses the parity of the number of crossed faces (reference: kaolin/ops/mesh/check_sign.py:56-155).

    Args:
        verts (torch.Tensor): (B, V, 3).  faces (torch.LongTensor): (F, 3).  points (torch.Tensor): (B, N, 3).
        hash_resolution (int): only used by the reference's CPU path; kept for signature compatibility.

    Returns:
        (torch.BoolTensor): (B, N), True for points inside the mesh.
    """
    assert verts.device == points.device
    assert faces.device == points.device
    if not faces.dtype == torch.int64:
        raise TypeError(f"Expected faces entries to be torch.int64 "
                        f"but got {faces.dtype}.")
    if not isinstance(hash_resolution, int):
        raise TypeError(f"Expected hash_resolution to be int "
                        f"but got {type(hash_resolution)}.")
    for name, t, what, n in (('verts', verts, 'dimensions', 3), ('faces', faces, 'dimensions', 2),
                             ('points', points, 'dimensions', 3)):
        if t.ndim != n:
            raise ValueError(f"Expected {name} to have {n} dimensions "
                             f"but got {t.ndim} dimensions.")
    if verts.shape[2] != 3:
        raise ValueError(f"Expected verts to have 3 coordinates "
                         f"but got {verts.shape[2]} coordinates.")
    if faces.shape[1] != 3:
        raise ValueError(f"Expected faces to have 3 vertices "
                         f"but got {faces.shape[1]} vertices.")
    if points.shape[2] != 3:
        raise ValueError(f"Expected points to have 3 coordinates "
                         f"but got {points.shape[2]} coordinates.")
    if points.device.type != 'cuda':
        raise RuntimeError('check_sign: only the GPU path is implemented (the reference CPU path is a C++ TriangleHash, '
                           'out of scope: SURVEY.md section 2)')
    # normalise by the largest extent of each mesh (check_sign.py:139-145): the ray's far end is then surely outside
    extent = verts.max(dim=1)[0] - verts.min(dim=1)[0]                      # (B, 3)
    scale = extent.max(dim=1)[0].view(-1, 1, 1)
    verts, points = verts / scale, points / scale
    return torch.stack([_unbatched_check_sign_cuda(verts[i], faces, points[i]) for i in range(verts.shape[0])])
