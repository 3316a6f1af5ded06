"""Triangle-mesh metrics: ``point_to_mesh_distance`` (+ ``average_edge_length``).

API mirror of kaolin/metrics/trianglemesh.py:20-149,279-315.  The point -> triangle search and its gradient run
in hand-written HIP kernels (kaolin_amd/csrc/triangle_distance.hip) reached through ``kaolin_amd._C.metrics``.
"""
import torch

from .. import _C
from ..ops.mesh import uniform_laplacian

__all__ = ['point_to_mesh_distance', 'average_edge_length', 'uniform_laplacian_smoothing']


class _UnbatchedTriangleDistanceCuda(torch.autograd.Function):
    """Same contract as the reference's shim (metrics/trianglemesh.py:125-149): the operator writes into three
    caller-allocated outputs; the face index and region code are non-differentiable; backward accumulates into
    zero-initialised gradients."""

    @staticmethod
    def forward(ctx, points, face_vertices):
        pts, tris = points.contiguous(), face_vertices.contiguous()
        n, dev = pts.shape[0], pts.device
        # the operator writes every element of its three outputs when there is a face; with none they keep the zeros the
        # reference allocates (unbatched_triangle_distance_cuda.cu:247: the loop never runs)
        alloc = torch.empty if tris.shape[0] > 0 else torch.zeros
        out = (alloc(n, device=dev, dtype=pts.dtype), alloc(n, device=dev, dtype=torch.long),
               alloc(n, device=dev, dtype=torch.int32))
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
    # GPU tensors take the HIP operator, CPU tensors the torch formulation below (the reference's split, :88-93)
    fn = _UnbatchedTriangleDistanceCuda.apply if pointclouds.is_cuda else _unbatched_naive_point_to_mesh_distance
    per_item = [fn(pointclouds[b], face_vertices[b]) for b in range(pointclouds.shape[0])]
    if len(per_item) == 1:  # (B = 1: the item's outputs with a leading axis, no copy -- torch.stack would move 16 B per point)
        return tuple(column.unsqueeze(0) for column in per_item[0])
    return tuple(torch.stack(column, dim=0) for column in zip(*per_item))


def _closest_point_regions(points, tris):
    """(n, 3) points against (n or m, 3, 3) triangles, broadcast on the leading dimension(s):
    -> (region code, closest point) with the reference's decision cascade: vertex regions 1-3 (both neighbouring edge
    parameters outside), edge regions 4-6 (parameter inside [0, 1] and the point not above the edge in the triangle's
    plane), else the plane (0).  The first matching region in that order wins (unbatched_triangle_distance_cuda.cu:272-300)."""
    a, b, c = tris[..., 0, :], tris[..., 1, :], tris[..., 2, :]
    ab, bc, ca = b - a, c - b, a - c
    normal = -torch.cross(ab, ca, dim=-1)

    def dot(u, v):                     # x, y, z products added left to right (the order the reference's oracle and kernels use)
        return u[..., 0] * v[..., 0] + u[..., 1] * v[..., 1] + u[..., 2] * v[..., 2]

    def along(origin, edge):           # parameter of the projection of the point on the edge's line
        return dot(points - origin, edge) / dot(edge, edge)

    def outside(origin, edge):         # on the outer side of the edge, seen in the triangle's plane (or on the line)
        return dot(torch.cross(normal, edge, dim=-1), points - origin) <= 0

    t_ab, t_bc, t_ca = along(a, ab), along(b, bc), along(c, ca)
    conds = [(t_ca > 1.) & (t_ab < 0.), (t_ab > 1.) & (t_bc < 0.), (t_bc > 1.) & (t_ca < 0.),
             (t_ab >= 0.) & (t_ab <= 1.) & outside(a, ab), (t_bc >= 0.) & (t_bc <= 1.) & outside(b, bc),
             (t_ca >= 0.) & (t_ca <= 1.) & outside(c, ca)]
    unit = normal / normal.norm(dim=-1, keepdim=True)
    on_plane = points - unit * dot(points - a, unit).unsqueeze(-1)
    cands = [a.expand_as(on_plane), b.expand_as(on_plane), c.expand_as(on_plane), a + ab * t_ab.unsqueeze(-1),
             b + bc * t_bc.unsqueeze(-1), c + ca * t_ca.unsqueeze(-1)]
    region = torch.zeros(on_plane.shape[:-1], dtype=torch.int32, device=points.device)
    closest = on_plane
    for code in range(6, 0, -1):       # lower codes are assigned last: they win
        m = conds[code - 1]
        region = torch.where(m, torch.full_like(region, code), region)
        closest = torch.where(m.unsqueeze(-1), cands[code - 1], closest)
    return region, closest


def _unbatched_naive_point_to_mesh_distance(points, face_vertices):
    """All-pairs torch formulation of the point -> triangle-soup distance, any device (the reference's CPU path and test
    oracle of the same name, kaolin/metrics/trianglemesh.py:151-276): (N, 3) points, (F, 3, 3) triangles ->
    (squared distance (N), face index (N) int64, region code (N) int32).  The search runs without autograd in chunks of
    points; the distance is then recomputed against the selected faces only, so the graph is O(N), not O(N * F)."""
    n, f = points.shape[0], face_vertices.shape[0]
    nearest = torch.zeros(n, dtype=torch.long, device=points.device)
    with torch.no_grad():
        step = max(1, (1 << 22) // max(f, 1))
        for s0 in range(0, n, step):
            p = points[s0:s0 + step].detach().unsqueeze(1)                       # (c, 1, 3) against (1, F, 3, 3)
            _, closest = _closest_point_regions(p, face_vertices.detach().unsqueeze(0))
            delta = closest - p
            nearest[s0:s0 + step] = (delta[..., 0] * delta[..., 0] + delta[..., 1] * delta[..., 1] +
                                     delta[..., 2] * delta[..., 2]).argmin(dim=1)
    region, closest = _closest_point_regions(points, face_vertices[nearest])
    return ((closest - points) ** 2).sum(-1), nearest, region


def average_edge_length(vertices, faces):
    r"""Mean length of the three edges of every face, (B, F) (reference: metrics/trianglemesh.py:279-315)."""
    p = [torch.index_select(vertices, 1, faces[:, k]) for k in range(3)]
    lens = [torch.sqrt(torch.sum((a - b) ** 2, dim=2)) for a, b in ((p[1], p[0]), (p[2], p[0]), (p[1], p[2]))]
    return (lens[0] + lens[1] + lens[2]) / 3.


def uniform_laplacian_smoothing(vertices, faces):
    r"""One step of uniform Laplacian smoothing: every vertex moves to the mean of its neighbours
    (reference: kaolin/metrics/trianglemesh.py:318-350).

    Args:
        vertices (torch.Tensor): (B, V, 3).  faces (torch.LongTensor): (F, face_size).

    Returns:
        (torch.Tensor): smoothed vertices, (B, V, 3).
    """
    lap = uniform_laplacian(vertices.shape[1], faces).to(vertices.dtype)
    return torch.matmul(lap, vertices) + vertices
