"""unbatched_mesh_to_spc: conservative voxelization of a triangle soup into an SPC octree.

CPU: pins oracle/mesh_to_spc_oracle.inc against the reference's only known answer for this operator -- 4 triangles at
level 3: the 28 octree bytes, the 65 face indices (`torch.equal`) and the barycentric weights at the reference's own
tolerance (tests/python/kaolin/ops/conversions/test_trianglemesh.py:244-369) -- plus structural invariants of the octree.
The reference has no CPU implementation of this operator (CUDA only), hence no generated goldens.
GPU: the HIP path through the C ABI vs that table and, bit-exact (octree bytes, face indices, Morton order; barycentric
weights equal), vs the oracle on spheres and random soups at several levels, incl. stage boundaries (levels 0..7)."""
import pytest
import torch

import oracle

FACES = [[0, 1, 2], [2, 1, 3], [4, 5, 6], [7, 8, 9]]
VERTS = [[-0.4272, 0.0795, 0.3548], [-0.9217, 0.3106, 0.1516], [-0.2636, 0.3794, -0.7979], [0.1259, 0.9089, 0.7439],
         [0.0710, -0.6947, -0.0480], [0.6215, 0.2809, -0.0480], [0.4972, 0.3347, 0.4422], [-0.4374, 0.4967, -0.6047],
         [0.0397, 0.1230, -0.7417], [-0.3534, 0.9970, -0.4558]]
OCTREE = [252, 242, 213, 10, 5, 35, 29, 232, 172, 79, 170, 55, 245, 48, 7, 179, 81, 8, 162, 4, 209, 2, 32, 10, 176, 11, 4, 15]
FACE_IDX = [0, 0, 0, 0, 0, 0, 3, 1, 0, 0, 0, 0, 1, 3, 3, 3, 3, 1, 1, 3, 1, 1, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1, 1,
            1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 2, 2, 2, 2]
BARY = [[4.5012e-08, 7.7766e-01], [2.8764e-01, 4.0506e-01], [3.5860e-08, 4.7760e-01], [1.0753e-01, 5.5666e-01],
        [2.5024e-08, 3.0500e-04], [5.3537e-03, 1.7265e-01], [2.4690e-01, 7.5310e-01], [8.8672e-01, 4.2031e-09],
        [4.0263e-01, 1.9203e-05], [6.0202e-01, 3.6483e-08], [2.2252e-01, 1.5161e-01], [4.3968e-01, 1.3058e-01],
        [7.3768e-01, 2.0286e-02], [6.2631e-01, 8.5154e-02], [2.8269e-01, 2.0198e-02], [7.7711e-08, 4.6322e-01],
        [9.7272e-08, 2.4475e-01], [6.3429e-01, 1.7624e-01], [4.4455e-01, 2.6633e-01], [1.8181e-01, 2.8813e-08],
        [7.0239e-01, 3.2221e-08], [5.5041e-01, 2.5328e-02], [1.7266e-01, 8.1010e-01], [1.7232e-08, 9.5489e-01],
        [5.0481e-01, 3.8403e-01], [6.9277e-01, 3.0723e-01], [3.2469e-01, 5.3563e-01], [2.7070e-08, 7.3333e-01],
        [1.4894e-01, 5.9743e-01], [5.6993e-09, 6.5052e-01], [8.0139e-01, 4.1631e-08], [1.0000e+00, 0.0000e+00],
        [2.5233e-01, 4.4148e-01], [2.5480e-01, 3.5643e-01], [6.5063e-02, 4.4652e-01], [3.6067e-01, 1.1542e-01],
        [1.7093e-01, 2.0551e-01], [1.7340e-01, 1.2046e-01], [4.2470e-08, 4.2354e-01], [2.3278e-08, 2.7855e-01],
        [4.6319e-08, 1.9574e-01], [9.2212e-01, 7.7879e-02], [7.2775e-01, 2.7225e-01], [6.1808e-01, 3.8192e-01],
        [4.2371e-01, 5.7629e-01], [8.7880e-01, 2.5213e-08], [7.0510e-01, 8.7381e-09], [6.1462e-01, 1.1334e-01],
        [4.1944e-01, 2.4449e-01], [3.7678e-01, 3.8143e-08], [3.8031e-08, 9.9842e-01], [2.2935e-01, 7.7065e-01],
        [1.1967e-01, 8.8033e-01], [0.0000e+00, 1.0000e+00], [2.2426e-01, 3.7564e-01], [2.0308e-01, 1.5850e-08],
        [2.3061e-02, 3.8613e-02], [3.9331e-01, 1.1329e-08], [2.5610e-01, 1.6592e-08], [2.0898e-01, 1.9427e-08],
        [7.1771e-02, 1.6657e-08], [1.1603e-01, 5.9735e-01], [1.1001e-01, 1.2918e-01], [2.4004e-08, 6.5422e-01],
        [2.3279e-08, 1.8040e-01]]


def reference_case(device='cpu'):
    v = torch.tensor(VERTS, device=device)
    return v[torch.tensor(FACES, device=device)].contiguous()


def check_reference_table(octree, face_idx, bary):
    assert torch.equal(octree.cpu(), torch.tensor(OCTREE, dtype=torch.uint8))
    assert torch.equal(face_idx.cpu(), torch.tensor(FACE_IDX))
    assert torch.allclose(bary.cpu(), torch.tensor(BARY), atol=1e-3, rtol=1e-3)


def test_oracle_reference_table():
    check_reference_table(*oracle.mesh_to_spc(reference_case(), 3))


def octree_levels(octree, level):
    """(node count per level, number of set bits of the last level) by walking the bytes root first."""
    counts, pos, nodes = [], 0, 1
    for _ in range(level):
        counts.append(nodes)
        nodes = int(sum(bin(int(b)).count('1') for b in octree[pos:pos + nodes]))
        pos += counts[-1]
    return counts, nodes, pos


def sphere_soup(level=8, radius=0.6):
    from kaolin_amd.utils.testing import geodesic_sphere
    v, f = geodesic_sphere(level)
    return (v.float() * (radius / 0.5))[f].contiguous()


def random_soup(n, seed, size=0.3):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 1, 3, generator=g) * 1.6 - 0.8
    fv = c + (torch.rand(n, 3, 3, generator=g) - 0.5) * size
    fv[::17] *= 1.6                                     # some triangles stick out of [-1,1]^3
    return fv.contiguous()


@pytest.mark.parametrize('level', [1, 4, 6])
def test_oracle_octree_is_consistent(level):
    fv = sphere_soup(6)
    octree, face_idx, bary, mortons = oracle.mesh_to_spc(fv, level, omp=True, return_mortons=True)
    counts, leaves, used = octree_levels(octree, level)
    assert used == octree.numel() and leaves == face_idx.numel() == bary.shape[0]
    assert bool((mortons[1:] > mortons[:-1]).all())                        # strictly ascending = unique + sorted
    assert int(face_idx.min()) >= 0 and int(face_idx.max()) < fv.shape[0]
    assert bool((bary >= 0).all()) and bool((bary.sum(-1) <= 1 + 1e-5).all())
    # every vertex of the mesh lies in an occupied voxel (conservative)
    g = ((fv.reshape(-1, 3) + 1.) * 0.5 * (1 << level)).floor().clamp(0, (1 << level) - 1).long()
    code = torch.zeros(g.shape[0], dtype=torch.long)
    for i in range(level):
        code |= ((g[:, 2] >> i) & 1) << (3 * i)
        code |= ((g[:, 1] >> i) & 1) << (3 * i + 1)
        code |= ((g[:, 0] >> i) & 1) << (3 * i + 2)
    assert bool(torch.isin(code, mortons).all())


def test_oracle_nothing_inside():
    fv = reference_case() + 5.
    octree, face_idx, bary = oracle.mesh_to_spc(fv, 3)
    assert octree.shape == (0,) and face_idx.shape == (0,) and bary.shape == (0, 3)


# ====================================================================================================== GPU
@pytest.mark.gpu
def test_gpu_reference_table():
    import kaolin_amd as kal
    check_reference_table(*kal.ops.conversions.unbatched_mesh_to_spc(reference_case('cuda'), 3))
    octree, face_idx, bary = kal._C.ops.conversions.mesh_to_spc_cuda(reference_case('cuda'), 3)
    assert octree.dtype == torch.uint8 and face_idx.dtype == torch.long and bary.dtype == torch.float32


def assert_matches_oracle(fv, level):
    import kaolin_amd as kal
    got = kal.ops.conversions.unbatched_mesh_to_spc(fv.cuda(), level)
    ref = oracle.mesh_to_spc(fv, level, omp=True)
    assert got[0].shape == ref[0].shape and got[1].shape == ref[1].shape and got[2].shape == ref[2].shape
    assert torch.equal(got[0].cpu(), ref[0])
    assert torch.equal(got[1].cpu(), ref[1])
    assert torch.equal(got[2].cpu(), ref[2])                               # same float expressions, no atomics
    return got


@pytest.mark.gpu
@pytest.mark.parametrize('level', [0, 1, 2, 3, 4, 5, 6, 7])
def test_gpu_random_soup_bit_exact(level):
    assert_matches_oracle(random_soup(300, level), level)


@pytest.mark.gpu
@pytest.mark.parametrize('level', [5, 8])
def test_gpu_sphere_bit_exact(level):
    octree, face_idx, bary = assert_matches_oracle(sphere_soup(16), level)
    counts, leaves, used = octree_levels(octree.cpu(), level)
    assert used == octree.numel() and leaves == face_idx.numel()


@pytest.mark.gpu
def test_gpu_large_triangles_and_empty():
    import kaolin_amd as kal
    big = torch.tensor([[[-0.9, -0.9, -0.2], [0.9, -0.8, 0.1], [0.0, 0.9, 0.3]],
                        [[-0.9, 0.1, -0.9], [0.9, 0.2, -0.8], [0.1, 0.3, 0.9]]])
    assert_matches_oracle(big, 7)
    octree, face_idx, bary = kal.ops.conversions.unbatched_mesh_to_spc((big + 5.).cuda(), 4)
    assert octree.shape == (0,) and face_idx.shape == (0,) and bary.shape == (0, 3)
    octree, face_idx, bary = kal.ops.conversions.unbatched_mesh_to_spc(big[:0].cuda(), 4)
    assert octree.shape == (0,) and face_idx.shape == (0,) and bary.shape == (0, 3)
    with pytest.raises(NotImplementedError):
        kal.ops.conversions.unbatched_mesh_to_spc(torch.zeros(2, 4, 4, device='cuda'), 3)
    with pytest.raises(RuntimeError, match='size 3 on dimension 1'):
        kal._C.ops.conversions.mesh_to_spc_cuda(torch.zeros(2, 4, 3, device='cuda'), 3)
    with pytest.raises(RuntimeError, match='Float'):
        kal._C.ops.conversions.mesh_to_spc_cuda(torch.zeros(2, 3, 3, device='cuda', dtype=torch.double), 3)


@pytest.mark.gpu
def test_gpu_full_size_sphere_level_9():
    """The 50k-face sphere of config C4/C5 at level 9 (512^3): structural invariants + an oracle comparison."""
    import kaolin_amd as kal
    fv = sphere_soup(50)
    octree, face_idx, bary = kal.ops.conversions.unbatched_mesh_to_spc(fv.cuda(), 9)
    counts, leaves, used = octree_levels(octree.cpu().numpy(), 9)
    assert used == octree.numel() and leaves == face_idx.numel() == bary.shape[0]
    assert int(face_idx.min()) >= 0 and int(face_idx.max()) < fv.shape[0]
    ref = oracle.mesh_to_spc(fv, 9, omp=True)
    assert torch.equal(octree.cpu(), ref[0]) and torch.equal(face_idx.cpu(), ref[1]) and torch.equal(bary.cpu(), ref[2])
