"""check_sign: ray-parity inside test.

CPU: pins oracle/meshint_oracle.inc against the reference's known answers -- the 33-point table with points
projecting exactly onto vertices / edges (tests/python/kaolin/ops/mesh/test_check_sign.py:27-112,167-185, incl. zero-area
faces) and the docstring example (kaolin/ops/mesh/check_sign.py:75-92).
GPU: the HIP kernel through the C ABI vs the same KATs and, bit-exact (integer crossing counts), vs the oracle."""
import pytest
import torch

import oracle

VERTS = [[1., 0., 0.], [1., 0., 1.], [1., -1., -1.], [1., 1., -1.], [-1., 0., 0.], [-1., 0., -4.], [-1., -4., 4.], [-1., 4., 4.]]
FACES = [[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 2, 6], [2, 3, 5], [3, 1, 7], [5, 6, 2], [6, 7, 1], [7, 5, 3], [4, 6, 5], [4, 7, 6],
         [4, 5, 7]]
POINTS = [[0.9, 0., 0.], [0.9, 0., -1.], [0.9, 0., -0.9], [0.9, 0.1, -1.0], [0.9, 0., 1.], [0.9, -1., -1.], [0.9, 1., -1.],
          [-0.99, 0., -3.9], [-0.99, -3.9, 3.9], [-0.99, 3.9, 3.9],
          [0.9, 0., -4.], [0.9, -4., 4.], [0.9, 4., 4.], [0.9, 0., 4.], [-0.9, 0., -3.9], [-0.9, -3.9, 3.9], [-0.9, 3.9, 3.9],
          [0.5, 0., 5.], [0.5, -5., 4.], [1.1, 0., 0.], [1.1, 0., -1.], [1.1, 0., -0.9], [1.1, 0.1, -1.0], [1.1, 0., 1.],
          [1.1, -1., -1.], [1.1, 1., -1.], [-1.1, 0., 0.], [-1.1, 0., -1.], [-1.1, 0., -0.9], [-1.1, 0.1, -1.0], [-1.1, 0., 1.],
          [-1.1, -1., -1.], [-1.1, 1., -1.]]
EXPECTED = [True] * 10 + [False] * 23


def kat(dtype, device='cpu'):
    v = torch.tensor([VERTS], dtype=dtype, device=device)
    verts = torch.cat([v, -v], dim=0)
    faces = torch.tensor(FACES, dtype=torch.long, device=device)
    p = torch.tensor([POINTS], dtype=dtype, device=device)
    points = torch.cat([p, torch.flip(-p, dims=(1,))], dim=0)
    e = torch.tensor([EXPECTED], device=device)
    return verts, faces, points, torch.cat([e, torch.flip(e, dims=(1,))], dim=0)


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
def test_oracle_kat_table(dtype):
    verts, faces, points, expected = kat(dtype)
    assert torch.equal(oracle.check_sign(verts, faces, points), expected)
    assert torch.equal(oracle.check_sign(verts[:1], faces, points[:1]), expected[:1])
    degenerate = torch.cat([faces, torch.tensor([[1, 1, 1], [0, 0, 0], [2, 2, 2], [3, 3, 3]])])
    assert torch.equal(oracle.check_sign(verts, degenerate, points), expected)


def docstring_case(device='cpu'):
    verts = torch.tensor([[[0., 0., 0.], [1., 0.5, 1.], [0.5, 1., 1.], [1., 1., 0.5]]], device=device)
    faces = torch.tensor([[0, 3, 1], [0, 1, 2], [0, 2, 3], [3, 2, 1]], device=device)
    axis = torch.linspace(0.1, 0.9, 3, device=device)
    p_x, p_y, p_z = torch.meshgrid(axis + 0.01, axis + 0.02, axis + 0.03, indexing='ij')
    points = torch.cat((p_x.unsqueeze(-1), p_y.unsqueeze(-1), p_z.unsqueeze(-1)), dim=3).view(1, -1, 3)
    expected = torch.zeros(27, dtype=torch.bool, device=device)
    expected[[0, 13, 17, 23, 25]] = True
    return verts, faces, points, expected[None]


def test_oracle_docstring_example():
    verts, faces, points, expected = docstring_case()
    assert torch.equal(oracle.check_sign(verts, faces, points), expected)


def test_oracle_sphere_inside_outside():
    from kaolin_amd.utils.testing import geodesic_sphere
    v, f = geodesic_sphere(6)
    torch.manual_seed(0)
    pts = torch.rand(1, 4000, 3, dtype=torch.double) * 1.2 - 0.6
    got = oracle.check_sign(v[None], f, pts, omp=True)
    r = pts.norm(dim=-1)
    sure = (r < 0.49) | (r > 0.51)          # the faceted sphere lies between radius ~0.493 and 0.5
    assert torch.equal(got[sure], (r < 0.5)[sure])


# ------------------------------------------------------------------ GPU
def _mesh():
    from kaolin_amd.ops import mesh
    return mesh


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float, torch.double])
def test_gpu_kat_table(dtype):
    verts, faces, points, expected = kat(dtype, 'cuda')
    m = _mesh()
    assert torch.equal(m.check_sign(verts, faces, points), expected)
    assert torch.equal(m.check_sign(verts[:1], faces, points[:1]), expected[:1])
    degenerate = torch.cat([faces, torch.tensor([[1, 1, 1], [0, 0, 0], [2, 2, 2], [3, 3, 3]], device='cuda')])
    assert torch.equal(m.check_sign(verts, degenerate, points), expected)
    v, f, p, e = docstring_case('cuda')
    assert torch.equal(m.check_sign(v, f, p), e)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('level,N', [(4, 3000), (16, 20000), (32, 1000)])
def test_gpu_counts_bit_exact_vs_oracle(dtype, level, N):
    """The crossing COUNT (not just its parity) must equal the oracle's: integer-exact semantics incl. points whose
    projection hits mesh vertices / edges exactly (queries placed on the vertices' (y,z))."""
    from kaolin_amd.utils.testing import geodesic_sphere
    from kaolin_amd import _C
    v, f = geodesic_sphere(level)
    v = v.to(dtype)
    torch.manual_seed(level)
    pts = torch.rand(N, 3, dtype=dtype) * 1.2 - 0.6
    pts[:200, 1:] = v[torch.randint(0, v.shape[0], (200,))][:, 1:]          # exactly over vertices
    mid = (v[f[:, 0]] + v[f[:, 1]]) / 2
    pts[200:400, 1:] = mid[torch.randint(0, mid.shape[0], (200,))][:, 1:]  # over edge midpoints
    a, b, c = v[f[:, 0]].contiguous(), v[f[:, 1]].contiguous(), v[f[:, 2]].contiguous()
    ref = oracle.mesh_intersection(pts, a, b, c, omp=True)
    got = _C.ops.unbatched_mesh_intersection_cuda(pts.cuda(), a.cuda(), b.cuda(), c.cuda())
    assert torch.equal(got.cpu(), ref)
    inside = _mesh().check_sign(v[None].cuda(), f.cuda(), pts[None].cuda())
    assert torch.equal(inside.cpu(), oracle.check_sign(v[None], f, pts[None], omp=True))


@pytest.mark.gpu
def test_gpu_argument_errors():
    verts, faces, points, _ = kat(torch.float, 'cuda')
    m = _mesh()
    with pytest.raises(TypeError, match=r'Expected faces entries to be torch.int64 but got torch.int32.'):
        m.check_sign(verts, faces.int(), points)
    with pytest.raises(TypeError, match=r"Expected hash_resolution to be int but got <class 'float'>."):
        m.check_sign(verts, faces, points, 512.0)
    with pytest.raises(ValueError, match=r'Expected verts to have 3 dimensions but got 4 dimensions.'):
        m.check_sign(verts.unsqueeze(-1), faces, points)
    with pytest.raises(ValueError, match=r'Expected points to have 3 coordinates but got 2 coordinates.'):
        m.check_sign(verts, faces, points[..., :2])
