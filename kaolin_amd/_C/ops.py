"""Host shims of ``kaolin._C.ops`` operators, plus the fused voxelizer (the reference has no ``kaolin._C`` operator for
it: the op is pure PyTorch, kaolin/ops/conversions/trianglemesh.py:29-110; that entry point exists only on our side)."""
import torch

from .. import _lib
from .._checks import torch_check


def trianglemeshes_to_voxelgrids_cuda(vertices, faces, resolution, origin=None, scale=None, return_sparse=False):
    """vertices (B, V, 3) raw; faces (F, 3) int64 shared by the batch; origin (B, 3) / scale (B) of the normalisation
    ``(vertices - origin) / scale`` or None (per-mesh minimum / largest extent, computed on the device)
    -> dense (B, R, R, R) grid of 0/1 in the vertices' dtype; ``return_sparse``: the same as a coalesced sparse COO tensor built
    from a BIT grid (R^3 / 8 bytes per mesh) -- R^3 scalars are never allocated (reference: the COO of the unique voxel indices,
    kaolin/ops/conversions/pointcloud.py:66-73)."""
    fn = 'trianglemeshes_to_voxelgrids_cuda'
    torch_check(vertices.is_cuda and faces.is_cuda, f'{fn}: vertices and faces must be CUDA tensors')
    torch_check(vertices.dim() == 3 and vertices.size(2) == 3, 'vertices must of size {batch_size, num_vertices, 3}')
    torch_check(faces.dim() == 2 and faces.size(1) == 3, 'faces must of size {num_faces, 3}')
    torch_check(faces.dtype == torch.long, 'faces must be long')
    v = vertices.contiguous()
    f = faces.contiguous()
    sfx = _lib.dtype_suffix(v.dtype, fn)
    B, V, F, R = v.size(0), v.size(1), f.size(0), int(resolution)
    if origin is not None:
        torch_check(origin.is_cuda and tuple(origin.shape) == (B, 3), 'origin must be a CUDA tensor of size {batch_size, 3}')
        origin = origin.to(v.dtype).contiguous()
    if scale is not None:
        torch_check(scale.is_cuda and scale.numel() == B, 'scale must be a CUDA tensor of size {batch_size}')
        scale = scale.to(v.dtype).reshape(B).contiguous()
    from .render.mesh import faces_in_range
    if not faces_in_range(f, V):
        # the kernels gather vertices unchecked; the reference's indexing raises (a device assert) on such a mesh
        raise IndexError(f'{fn}: faces hold an index outside [0, num_vertices)')
    lib = _lib.load()
    if return_sparse:
        with _lib.on_device(v.device):
            words = int(lib.kamd_trianglemeshes_to_voxelbits_words(R))
            bits = torch.empty((B, words), dtype=torch.int32, device=v.device)
            norm = _lib.workspace(lib.kamd_trianglemeshes_to_voxelgrids_workspace(B, V, v.element_size()), v.device)
            st = getattr(lib, f'kamd_trianglemeshes_to_voxelbits_{sfx}')(
                _lib.stream_ptr(v.device), B, V, F, R, _lib.ptr(v), _lib.ptr(f), _lib.ptr(origin), _lib.ptr(scale), _lib.ptr(norm),
                _lib.ptr(bits))
        _lib.check(st, fn)
        # compaction, sized by the occupied words: (mesh, word) of every non-zero word, its 32 bits, the set ones -> linear
        # voxel indices in ascending order per mesh (= the order of a coalesced COO tensor)
        nzw = torch.nonzero(bits)                                           # (n_words, 2), lexicographic
        w = bits[nzw[:, 0], nzw[:, 1]]
        on = ((w.unsqueeze(1) >> torch.arange(32, device=v.device, dtype=torch.int32)) & 1).bool()   # (n_words, 32)
        sel = torch.nonzero(on)                                             # (nnz, 2): (word slot, bit), lexicographic
        lin = nzw[sel[:, 0], 1] * 32 + sel[:, 1]
        idx = torch.stack([nzw[sel[:, 0], 0], lin // (R * R), (lin // R) % R, lin % R])
        return torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1], dtype=v.dtype, device=v.device), (B, R, R, R),
                                       is_coalesced=True)
    with _lib.on_device(v.device):
        grid = torch.empty((B, R, R, R), dtype=v.dtype, device=v.device)
        norm = _lib.workspace(lib.kamd_trianglemeshes_to_voxelgrids_workspace(B, V, v.element_size()), v.device)
        st = getattr(lib, f'kamd_trianglemeshes_to_voxelgrids_{sfx}')(
            _lib.stream_ptr(v.device), B, V, F, R, _lib.ptr(v), _lib.ptr(f), _lib.ptr(origin), _lib.ptr(scale), _lib.ptr(norm),
            _lib.ptr(grid))
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
    with _lib.on_device(points.device):
        result = torch.empty(n, dtype=points.dtype, device=points.device)
        st = getattr(lib, f'kamd_mesh_intersection_{sfx}')(
            _lib.stream_ptr(points.device), n, m, _lib.ptr(points), _lib.ptr(verts_1), _lib.ptr(verts_2),
            _lib.ptr(verts_3), _lib.ptr(result))
    _lib.check(st, fn)
    return result


def mesh_to_spc_cuda(face_vertices, level):
    """reference: kaolin/csrc/ops/conversions/mesh_to_spc/mesh_to_spc.cpp:27-42 (``_C.ops.conversions.mesh_to_spc_cuda``):
    face_vertices (F,3,3) float32 in [-1,1]^3 -> [octree uint8 (num_nodes), face_ids int64 (num_voxels),
    barycoords float (num_voxels, 2)]; nothing occupied -> sizes (0,), (0,), (0, 3) as in the reference.

    Result sizes depend on the data, so the host reads one count per stage of three octree levels and the level
    sizes once (the reference reads one count per level, twice); see include/kaolin_amd.h for the sequence."""
    fn = 'mesh_to_spc_cuda'
    torch_check(face_vertices.is_cuda, 'face_vertices must be a CUDA tensor')
    torch_check(face_vertices.is_contiguous(), 'face_vertices must be contiguous')
    torch_check(face_vertices.dim() == 3, f'face_vertices must have 3 dimensions, but got {face_vertices.dim()}')
    torch_check(face_vertices.size(1) == 3, f'face_vertices must have size 3 on dimension 1, but got {face_vertices.size(1)}')
    torch_check(face_vertices.size(2) == 3, f'face_vertices must have size 3 on dimension 2, but got {face_vertices.size(2)}')
    torch_check(face_vertices.dtype == torch.float32, 'expected scalar type Float but found ' + _lib.pretty_dtype(face_vertices.dtype))
    level = int(level)
    torch_check(0 <= level <= 15, 'level must be in [0, 15]')
    dev = face_vertices.device
    lib = _lib.load()
    sp = _lib.stream_ptr(dev)
    depth = lib.kamd_mesh_to_spc_stage_levels()

    def empty_result():
        return [torch.empty(0, dtype=torch.uint8, device=dev), torch.empty(0, dtype=torch.long, device=dev),
                torch.zeros((0, 3), dtype=torch.float32, device=dev)]

    with _lib.on_device(dev):
        n = face_vertices.size(0)
        if n == 0:
            return empty_result()
        morton = torch.zeros(n, dtype=torch.long, device=dev)
        tri = torch.arange(n, dtype=torch.long, device=dev)
        level_from, tested = 0, 0
        while True:
            level_to = min(level_from + depth, level)
            counts = torch.empty(n, dtype=torch.int32, device=dev)
            offsets = torch.empty(n + 1, dtype=torch.long, device=dev)
            scan_ws = torch.empty(lib.kamd_mesh_to_spc_scan_workspace(n), dtype=torch.uint8, device=dev)
            _lib.check(lib.kamd_mesh_to_spc_stage_count(sp, n, _lib.ptr(face_vertices), _lib.ptr(morton), _lib.ptr(tri),
                                                        level_from, level_to, tested, _lib.ptr(counts), _lib.ptr(offsets),
                                                        _lib.ptr(scan_ws)), fn)
            total = int(offsets[n].item())                     # data-dependent size: one host read per stage
            if total == 0:
                return empty_result()
            morton_out = torch.empty(total, dtype=torch.long, device=dev)
            tri_out = torch.empty(total, dtype=torch.long, device=dev)
            _lib.check(lib.kamd_mesh_to_spc_stage_emit(sp, n, _lib.ptr(face_vertices), _lib.ptr(morton), _lib.ptr(tri),
                                                       level_from, level_to, tested, _lib.ptr(offsets), _lib.ptr(morton_out),
                                                       _lib.ptr(tri_out)), fn)
            morton, tri, n = morton_out, tri_out, total
            if level_to == level:
                break
            level_from, tested = level_to, 1
        nbytes = lib.kamd_mesh_to_spc_build_workspace(n, level)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        sizes = torch.empty(1 + level, dtype=torch.long, device=dev)
        _lib.check(lib.kamd_mesh_to_spc_build(sp, n, level, _lib.ptr(morton), _lib.ptr(tri), _lib.ptr(ws), nbytes,
                                              _lib.ptr(sizes)), fn)
        host_sizes = sizes.tolist()                            # voxels + nodes per octree level: one host read
        num_voxels, octree_bytes = host_sizes[0], sum(host_sizes[1:])
        octree = torch.empty(octree_bytes, dtype=torch.uint8, device=dev)
        face_ids = torch.empty(num_voxels, dtype=torch.long, device=dev)
        bary = torch.empty((num_voxels, 2), dtype=torch.float32, device=dev)
        _lib.check(lib.kamd_mesh_to_spc_results(sp, n, level, _lib.ptr(face_vertices), _lib.ptr(ws), num_voxels, octree_bytes,
                                                _lib.ptr(octree), _lib.ptr(face_ids), _lib.ptr(bary)), fn)
    return [octree, face_ids, bary]


# the reference groups these operators in sub-modules: kaolin._C.ops.mesh / kaolin._C.ops.conversions (bindings.cpp)
import types as _types  # noqa: E402
mesh = _types.SimpleNamespace(unbatched_mesh_intersection_cuda=unbatched_mesh_intersection_cuda)
conversions = _types.SimpleNamespace(mesh_to_spc_cuda=mesh_to_spc_cuda)
