"""Mesh helpers on the DIB-R path (pure torch): per-face gathers and normals, plus the vertex-set
subdivision the voxelizer's definition rests on."""
import torch

__all__ = ['index_vertices_by_faces', 'face_normals', 'check_sign', 'adjacency_matrix', 'uniform_laplacian']


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


def adjacency_matrix(num_vertices, faces, sparse=True):
    """Vertex adjacency of a polygon mesh, (V, V), entries 1 between vertices that share a face edge (consecutive corners of
    a face, cyclically).  Behaviour of kaolin/ops/mesh/mesh.py:77-114: COO tensor by default, dense with ``sparse=False``."""
    k = faces.shape[1]
    src = faces.reshape(-1)
    dst = torch.roll(faces, -1, dims=1).reshape(-1)
    pairs = torch.unique(torch.stack([torch.cat([src, dst]), torch.cat([dst, src])], dim=1), dim=0)
    if k < 2:
        pairs = pairs[:0]
    if sparse:
        return torch.sparse_coo_tensor(pairs.t(), torch.ones(pairs.shape[0], dtype=torch.float, device=faces.device),
                                       (num_vertices, num_vertices))
    dense = torch.zeros((num_vertices, num_vertices), dtype=torch.float, device=faces.device)
    dense[pairs[:, 0], pairs[:, 1]] = 1.
    return dense


def uniform_laplacian(num_vertices, faces):
    """Uniform Laplacian of a mesh, (V, V): 1 / #neighbours(i) between neighbours, -1 on the diagonal, rows of isolated
    vertices are zero off the diagonal (behaviour of kaolin/ops/mesh/mesh.py:116-164)."""
    adj = adjacency_matrix(num_vertices, faces, sparse=False)
    degree = adj.sum(dim=1, keepdim=True)
    lap = torch.where(degree > 0, adj / degree.clamp(min=1.), torch.zeros_like(adj))
    lap.fill_diagonal_(-1.)
    return lap


def _unbatched_check_sign_cuda(verts, faces, points):
    """kaolin/ops/mesh/check_sign.py:45-54."""
    from ... import _C
    corners = [verts[faces[:, k]].contiguous() for k in range(3)]
    crossings = _C.ops.unbatched_mesh_intersection_cuda(points.contiguous(), *corners)
    return crossings % 2 == 1.


def _unbatched_check_sign_torch(verts, faces, points):
    """The ray-parity count of the GPU operator in plain torch, for CPU tensors (the reference's CPU path is a C++
    ``TriangleHash``, kaolin/ops/mesh/check_sign.py:56-58; this follows the rules of its CUDA kernel,
    mesh_intersection_cuda.cu:101-218, so that both devices answer alike): a +x ray per point; a face counts when the point's
    (y, z) projection lies in the face's projected box and triangle and the ray's two ends lie on different sides of
    its plane; a projection exactly on an edge / vertex is credited to one of the faces sharing it."""
    p1, p2, p3 = (verts[faces[:, k]] for k in range(3))                          # (F, 3)

    def signed_volume(q):                                                         # (c, 1, 3) -> (c, F)
        n = torch.cross(p2 - p1, p3 - p1, dim=-1)
        return (n.unsqueeze(0) * (p1.unsqueeze(0) - q)).sum(-1) * -1.

    def signed_area(q, b, c):            # q (c, 1, 2); b, c (F, 2): side of q w.r.t. the edge, direction-normalised
        swap = (c[:, 0] > b[:, 0]) | ((b[:, 0] == c[:, 0]) & (c[:, 1] < b[:, 1]))
        lo = torch.where(swap.unsqueeze(-1), c, b)
        hi = torch.where(swap.unsqueeze(-1), b, c)
        val = (hi[:, 1] - lo[:, 1]) * (q[..., 0] - lo[:, 0]) + (lo[:, 0] - hi[:, 0]) * (q[..., 1] - lo[:, 1])
        return torch.where(swap, -val, val)

    def above(v, l, r):                  # v left of the directed line l -> r
        return ((r[..., 0] - l[..., 0]) * (v[..., 1] - l[..., 1]) - (r[..., 1] - l[..., 1]) * (v[..., 0] - l[..., 0])) > 0.

    a, b, c = p1[:, 1:], p2[:, 1:], p3[:, 1:]                                    # (y, z) projections
    lo = torch.minimum(torch.minimum(a, b), c).float()
    hi = torch.maximum(torch.maximum(a, b), c).float()
    counts = torch.zeros(points.shape[0], dtype=points.dtype, device=points.device)
    step = max(1, (1 << 21) // max(faces.shape[0], 1))
    for s0 in range(0, points.shape[0], step):
        pts = points[s0:s0 + step]
        q = pts[:, None, 1:]                                                     # (c, 1, 2)
        inbox = ((q.float() >= lo) & (q.float() <= hi)).all(-1) if pts.dtype != torch.double else \
            ((q >= lo.double()) & (q <= hi.double())).all(-1)
        far = pts + torch.tensor([10., 0., 0.], dtype=pts.dtype, device=pts.device)
        crosses = (signed_volume(pts[:, None, :]) > 0.) != (signed_volume(far[:, None, :]) > 0.)
        d1, d2, d3 = signed_area(q, a, b), signed_area(q, b, c), signed_area(q, c, a)
        inside = (d1 * d2 >= 0) & (d3 * d1 >= 0) & (d2 * d3 >= 0)
        hit = inbox & crosses & inside
        at_a, at_b, at_c = ((q == x).all(-1) for x in (a, b, c))
        on_vertex = at_a | at_b | at_c
        on_e1 = ~on_vertex & (d1 == 0.)
        on_e2 = ~on_vertex & ~on_e1 & (d2 == 0.)
        on_e3 = ~on_vertex & ~on_e1 & ~on_e2 & (d3 == 0.)

        def pick(m_a, m_b, m_c, xa, xb, xc):   # per-(point, face) 2-vectors chosen by exclusive masks
            z = torch.zeros(hit.shape + (2,), dtype=pts.dtype, device=pts.device)
            return torch.where(m_a.unsqueeze(-1), xa, torch.where(m_b.unsqueeze(-1), xb, torch.where(m_c.unsqueeze(-1), xc, z)))

        A, Bv, C = (x.unsqueeze(0).expand(hit.shape + (2,)) for x in (a, b, c))
        # the two ends of the edge the point lies on (or the two other vertices when it lies on a vertex), and the third vertex
        e1 = torch.where(on_vertex.unsqueeze(-1), pick(at_a, at_b, at_c, Bv, A, A), pick(on_e1, on_e2, on_e3, A, Bv, C))
        e2 = torch.where(on_vertex.unsqueeze(-1), pick(at_a, at_b, at_c, C, C, Bv), pick(on_e1, on_e2, on_e3, Bv, C, A))
        other = pick(on_e1, on_e2, on_e3, C, A, Bv)
        flip = (e1[..., 0] > e2[..., 0]) | ((e1[..., 0] == e2[..., 0]) & (e1[..., 1] > e2[..., 1]))
        e1, e2 = torch.where(flip.unsqueeze(-1), e2, e1), torch.where(flip.unsqueeze(-1), e1, e2)
        on_edge = on_e1 | on_e2 | on_e3
        qq = q.expand(hit.shape + (2,))
        drop = (on_edge & above(other, e1, e2)) | \
               (on_vertex & ~(above(qq, e1, e2) & (e1[..., 0] < qq[..., 0]) & (e2[..., 0] >= qq[..., 0])))
        counts[s0:s0 + step] = (hit & ~drop).sum(dim=1).to(points.dtype)
    return counts % 2 == 1.


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
    # normalise by the largest extent of each mesh (check_sign.py:139-145): the ray's far end is then surely outside
    extent = verts.max(dim=1)[0] - verts.min(dim=1)[0]                      # (B, 3)
    scale = extent.max(dim=1)[0].view(-1, 1, 1)
    verts, points = verts / scale, points / scale
    one = _unbatched_check_sign_cuda if points.is_cuda else _unbatched_check_sign_torch
    return torch.stack([one(verts[i], faces, points[i]) for i in range(verts.shape[0])])
