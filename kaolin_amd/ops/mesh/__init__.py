"""Mesh helpers on the DIB-R path (pure torch): per-face gathers and normals, plus the vertex-set
subdivision the voxelizer's definition rests on."""
import torch

__all__ = ['index_vertices_by_faces', 'face_normals']


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
