"""Host shim for the fused voxelizer.  The reference has no ``kaolin._C`` operator here (the op is pure
PyTorch, kaolin/ops/conversions/trianglemesh.py:29-110); this entry point exists only on our side."""
import torch

from .. import _lib
from .._checks import torch_check


def trianglemeshes_to_voxelgrids_cuda(normalized_vertices, faces, resolution):
    """normalized_vertices (B, V, 3) already mapped by (v - origin) / scale; faces (F, 3) int64 shared by the
    batch -> dense (B, R, R, R) grid of 0/1 in the vertices' dtype."""
    fn = 'trianglemeshes_to_voxelgrids_cuda'
    torch_check(normalized_vertices.is_cuda and faces.is_cuda, f'{fn}: vertices and faces must be CUDA tensors')
    torch_check(normalized_vertices.dim() == 3 and normalized_vertices.size(2) == 3, 'vertices must of size {batch_size, num_vertices, 3}')
    torch_check(faces.dim() == 2 and faces.size(1) == 3, 'faces must of size {num_faces, 3}')
    torch_check(faces.dtype == torch.long, 'faces must be long')
    v = normalized_vertices.contiguous()
    f = faces.contiguous()
    sfx = _lib.dtype_suffix(v.dtype, fn)
    B, V, F, R = v.size(0), v.size(1), f.size(0), int(resolution)
    lib = _lib.load()
    with torch.cuda.device(v.device):
        grid = torch.empty((B, R, R, R), dtype=v.dtype, device=v.device)
        st = getattr(lib, f'kamd_trianglemeshes_to_voxelgrids_{sfx}')(
            _lib.stream_ptr(v.device), B, V, F, R, _lib.ptr(v), _lib.ptr(f), _lib.ptr(grid))
    _lib.check(st, fn)
    return grid


def unbatched_mesh_intersection_cuda(points, verts_1, verts_2, verts_3):
    """reference: kaolin/csrc/ops/mesh/mesh_intersection.cpp (bindings.cpp, ``_C.ops.mesh.unbatched_mesh_intersection_cuda``):
    points (N,3), verts_k (F,3) -> (N) tensor, the number of faces the +x ray from every point crosses."""
    fn = 'unbatched_mesh_intersection_cuda'
    for name, t in (('points', points), ('verts_1', verts_1), ('verts_2', verts_2), ('verts_3', verts_3)):
        torch_check(t.is_cuda, f'{name} must be a CUDA tensor')
    for name, t in (('points', points), ('verts_1', verts_1), ('verts_2', verts_2), ('verts_3', verts_3)):
        torch_check(t.is_contiguous(), f'{name} must be contiguous')
    n, m = points.size(0), verts_1.size(0)
    torch_check(list(points.shape) == [n, 3], 'points must of size {num_points, 3}')
    for name, t in (('verts_1', verts_1), ('verts_2', verts_2), ('verts_3', verts_3)):
        torch_check(list(t.shape) == [m, 3], f'{name} must of size {{num_faces, 3}}')
        torch_check(t.dtype == points.dtype, 'expected points and vertices to have the same scalar type')
    sfx = _lib.dtype_suffix(points.dtype, fn)
    lib = _lib.load()
    with torch.cuda.device(points.device):
        result = torch.empty(n, dtype=points.dtype, device=points.device)
        st = getattr(lib, f'kamd_mesh_intersection_{sfx}')(
            _lib.stream_ptr(points.device), n, m, _lib.ptr(points), _lib.ptr(verts_1), _lib.ptr(verts_2),
            _lib.ptr(verts_3), _lib.ptr(result))
    _lib.check(st, fn)
    return result
