"""``trianglemeshes_to_voxelgrids`` (API mirror of kaolin/ops/conversions/trianglemesh.py:29-110)."""
import torch

from ... import _C

__all__ = ['trianglemeshes_to_voxelgrids', 'unbatched_mesh_to_spc']


def _torch_dense(points_per_item, faces, resolution):
    """Device-agnostic torch path for non-GPU tensors (the reference op is itself plain torch and accepts CPU
    tensors): level-synchronous subdivision collecting the midpoint SET, then round + scatter."""
    thr = ((resolution - 1) / (resolution ** 2)) ** 2
    grids = []
    for verts in points_per_item:
        pts = [verts]
        v1, v2, v3 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
        while v1.shape[0] > 0:
            e = torch.stack([torch.sum((v1 - v2) ** 2, dim=1), torch.sum((v2 - v3) ** 2, dim=1),
                             torch.sum((v3 - v1) ** 2, dim=1)], dim=1)
            keep = e.max(dim=1)[0] > thr
            if not bool(keep.any()):
                break
            v1, v2, v3 = v1[keep], v2[keep], v3[keep]
            v4, v5, v6 = (v1 + v3) / 2, (v1 + v2) / 2, (v2 + v3) / 2
            pts += [v4, v5, v6]
            v1, v2, v3 = torch.cat((v1, v2, v4, v3)), torch.cat((v4, v5, v5, v4)), torch.cat((v5, v6, v6, v6))
        idx = torch.round(torch.cat(pts) * (resolution - 1)).long()
        idx = idx[((idx >= 0) & (idx <= resolution - 1)).all(dim=1)]
        g = torch.zeros((resolution,) * 3, dtype=verts.dtype, device=verts.device)
        g[idx[:, 0], idx[:, 1], idx[:, 2]] = 1
        grids.append(g)
    return torch.stack(grids)


def trianglemeshes_to_voxelgrids(vertices, faces, resolution, origin=None, scale=None, return_sparse=False):
    r"""Converts meshes to surface voxelgrids of a given resolution: vertices are normalised with
    ``(vertices - origin) / scale``, every triangle is subdivided until its longest edge is below the voxel
    size, and each resulting point marks the voxel ``round(p * (resolution - 1))``
    (reference: kaolin/ops/conversions/trianglemesh.py:29-110).

    Args:
        vertices (torch.Tensor): (B, V, 3).
        faces (torch.LongTensor): (F, 3), shared by the batch.
        resolution (int): grid size along each axis.
        origin (torch.Tensor): (B, 3) origin of the grid. Default: per-mesh minimum.
        scale (torch.Tensor): (B) scale of the grid. Default: largest per-mesh extent.
        return_sparse (bool): return a sparse COO tensor instead of a dense one.

    Returns:
        (torch.Tensor): binary voxelgrids (B, R, R, R) in the dtype of ``vertices``.

    Note:
        The GPU path subdivides a triangle at most ``L0 + 20`` times (``L0 <= 10`` levels are spread over threads), i.e.
        it is exact for normalised edges up to about ``2**20 / resolution`` -- a mesh thousands of times larger than the
        grid, which only a caller-supplied ``scale`` can produce.  Faces must index existing vertices (``IndexError``).
    """
    if not isinstance(resolution, int):
        raise TypeError(f"Expected resolution to be int "
                        f"but got {type(resolution)}.")
    assert resolution > 1
    if vertices.is_cuda and vertices.dtype in (torch.float32, torch.float64) and vertices.shape[1] > 0:
        # the normalisation (and its default origin / scale) is part of the device pass: no torch glue kernels
        if return_sparse:
            # the COO tensor straight from the marked voxels (a bit grid, compacted): no R^3 scalars on the way
            return _C.ops.trianglemeshes_to_voxelgrids_cuda(vertices, faces, resolution, origin, scale, return_sparse=True)
        dense = _C.ops.trianglemeshes_to_voxelgrids_cuda(vertices, faces, resolution, origin, scale)
    else:
        if origin is None:
            origin = torch.min(vertices, dim=1)[0]
        if scale is None:
            scale = torch.max(torch.max(vertices, dim=1)[0] - origin, dim=1)[0]
        normalized = (vertices - origin.unsqueeze(1)) / scale.view(-1, 1, 1)
        dense = _torch_dense(normalized, faces, resolution)
    return dense.to_sparse() if return_sparse else dense


def unbatched_mesh_to_spc(face_vertices, level):
    r"""Conservatively voxelizes a triangle soup into a structured point cloud octree: every cell of the
    :math:`2^\text{level}` grid over :math:`[-1, 1]^3` that a triangle touches (13-axis separating-axis test) becomes a
    point of the SPC (reference: kaolin/ops/conversions/trianglemesh.py:112-140).

    Args:
        face_vertices (torch.FloatTensor): vertices gathered per face, of shape :math:`(\text{num_faces}, 3, 3)`.
        level (int): depth of the octree.

    Returns:
        (torch.ByteTensor, torch.LongTensor, torch.FloatTensor):
            the octree (one byte per node, levels root first), of shape :math:`(\text{num_nodes})`; for every occupied
            voxel of the last level, in Morton order, the smallest index of a face touching it,
            :math:`(\text{num_voxels})`; and the first two barycentric weights, w.r.t. that face, of its point closest to
            the voxel centre, :math:`(\text{num_voxels}, 2)`.
    """
    if face_vertices.shape[-1] != 3:
        raise NotImplementedError('unbatched_mesh_to_spc is only implemented for triangle meshes')
    return tuple(_C.ops.conversions.mesh_to_spc_cuda(face_vertices.contiguous(), level))
