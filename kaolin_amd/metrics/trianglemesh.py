"""Triangle-mesh metrics: ``point_to_mesh_distance`` (+ ``average_edge_length``).

API mirror of kaolin/metrics/trianglemesh.py:20-149,279-315.  The point -> triangle search and its gradient run
in hand-written HIP kernels (kaolin_amd/csrc/triangle_distance.hip) reached through ``kaolin_amd._C.metrics``.
"""
import torch

from .. import _C

__all__ = ['point_to_mesh_distance', 'average_edge_length']


class _UnbatchedTriangleDistanceCuda(torch.autograd.Function):
    """Same contract as the reference's shim (metrics/trianglemesh.py:125-149): the operator writes into three
    caller-allocated outputs; the face index and region code are non-differentiable; backward accumulates into
    zero-initialised gradients."""

    @staticmethod
    def forward(ctx, points, face_vertices):
        pts, tris = points.contiguous(), face_vertices.contiguous()
        n, dev = pts.shape[0], pts.device
        out = (torch.zeros(n, device=dev, dtype=pts.dtype), torch.zeros(n, device=dev, dtype=torch.long),
               torch.zeros(n, device=dev, dtype=torch.int32))
        _C.metrics.unbatched_triangle_distance_forward_cuda(pts, tris, *out)
        ctx.mark_non_differentiable(out[1], out[2])
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(pts, tris, out[1], out[2])
        return out

    @staticmethod
    def backward(ctx, grad_dist, _grad_idx, _grad_type):
        if grad_dist is None:
            return None, None
        pts, tris, nearest_face, region = ctx.saved_tensors
        g_pts, g_tris = torch.zeros_like(pts), torch.zeros_like(tris)
        _C.metrics.unbatched_triangle_distance_backward_cuda(grad_dist.contiguous(), pts, tris, nearest_face, region,
                                                             g_pts, g_tris)
        return g_pts, g_tris


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
    per_item = [_UnbatchedTriangleDistanceCuda.apply(pointclouds[b], face_vertices[b]) for b in range(pointclouds.shape[0])]
    return tuple(torch.stack(column, dim=0) for column in zip(*per_item))


def average_edge_length(vertices, faces):
    r"""Mean length of the three edges of every face, (B, F) (reference: metrics/trianglemesh.py:279-315)."""
    p = [torch.index_select(vertices, 1, faces[:, k]) for k in range(3)]
    lens = [torch.sqrt(torch.sum((a - b) ** 2, dim=2)) for a, b in ((p[1], p[0]), (p[2], p[0]), (p[1], p[2]))]
    return (lens[0] + lens[1] + lens[2]) / 3.
