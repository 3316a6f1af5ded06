"""Triangle-mesh metrics: ``point_to_mesh_distance`` (+ ``average_edge_length``).

API mirror of kaolin/metrics/trianglemesh.py:20-149,279-315.  The point -> triangle search and its gradient run
in hand-written HIP kernels (kaolin_amd/csrc/triangle_distance.hip) reached through ``kaolin_amd._C.metrics``.
"""
import torch

from .. import _C

__all__ = ['point_to_mesh_distance', 'average_edge_length']


class _UnbatchedTriangleDistanceCuda(torch.autograd.Function):
    """Same contract as the reference's shim (metrics/trianglemesh.py:125-149): the three outputs are allocated
    here (zeros) and filled by the operator; face_idx / dist_type are non-differentiable; backward accumulates
    into zero-initialised grad_points / grad_face_vertices."""

    @staticmethod
    def forward(ctx, points, face_vertices):
        num_points = points.shape[0]
        points, face_vertices = points.contiguous(), face_vertices.contiguous()
        min_dist = torch.zeros((num_points), device=points.device, dtype=points.dtype)
        min_dist_idx = torch.zeros((num_points), device=points.device, dtype=torch.long)
        dist_type = torch.zeros((num_points), device=points.device, dtype=torch.int32)
        _C.metrics.unbatched_triangle_distance_forward_cuda(points, face_vertices, min_dist, min_dist_idx, dist_type)
        ctx.save_for_backward(points, face_vertices, min_dist_idx, dist_type)
        ctx.mark_non_differentiable(min_dist_idx, dist_type)
        return min_dist, min_dist_idx, dist_type

    @staticmethod
    def backward(ctx, grad_dist, grad_face_idx, grad_dist_type):
        points, face_vertices, face_idx, dist_type = ctx.saved_tensors
        grad_points = torch.zeros_like(points)
        grad_face_vertices = torch.zeros_like(face_vertices)
        _C.metrics.unbatched_triangle_distance_backward_cuda(
            grad_dist.contiguous(), points, face_vertices, face_idx, dist_type, grad_points, grad_face_vertices)
        return grad_points, grad_face_vertices


def point_to_mesh_distance(pointclouds, face_vertices):
    r"""Squared euclidean distance from each point to the closest point of a triangle mesh
    (reference: kaolin/metrics/trianglemesh.py:20-99).  The distance is not signed.

    Args:
        pointclouds (torch.Tensor): (B, N, 3).
        face_vertices (torch.Tensor): (B, F, 3, 3) vertices indexed by faces.

    Returns:
        (torch.Tensor, torch.LongTensor, torch.IntTensor): squared distances (B, N); index of the closest face
        (B, N); region code (B, N): 0 the face interior, 1-3 vertex v1/v2/v3, 4-6 edge v1v2 / v2v3 / v3v1.
    """
    dists, idxs, types = [], [], []
    for i in range(pointclouds.shape[0]):
        d, f, t = _UnbatchedTriangleDistanceCuda.apply(pointclouds[i], face_vertices[i])
        dists.append(d)
        idxs.append(f)
        types.append(t)
    return torch.stack(dists, dim=0), torch.stack(idxs, dim=0), torch.stack(types, dim=0)


def average_edge_length(vertices, faces):
    r"""Mean length of the three edges of every face, (B, F) (reference: metrics/trianglemesh.py:279-315)."""
    p = [torch.index_select(vertices, 1, faces[:, k]) for k in range(3)]
    lens = [torch.sqrt(torch.sum((a - b) ** 2, dim=2)) for a, b in ((p[1], p[0]), (p[2], p[0]), (p[1], p[2]))]
    return (lens[0] + lens[1] + lens[2]) / 3.
