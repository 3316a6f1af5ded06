"""The torch-only pieces of the API that the reference's tests reach into (private oracles, CPU branches, small mesh
metrics), pinned -- without a GPU -- to goldens produced by the REFERENCE's own functions of the same names
(tests/golden/make_golden.py) and to the reference's known answers."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN_DIR


def test_naive_deftet_renderer_vs_reference_goldens():
    """_naive_deftet_sparse_render(knum=1) on the rasterizer fixtures (test_rasterization.py:137-233): face_idx equal,
    features and autograd gradients at the reference's tolerances."""
    from kaolin_amd.render.mesh.deftet import _naive_deftet_sparse_render
    g = np.load(os.path.join(GOLDEN_DIR, 'rasterize.npz'))
    gb = np.load(os.path.join(GOLDEN_DIR, 'rasterize_backward.npz'))
    H = W = 32
    for dn, dtype in (('f32', torch.float), ('f64', torch.double)):
        for flip in (0, 1):
            tag = f'{dn}_flip{flip}'
            t = lambda k: torch.from_numpy(g[f'{tag}_{k}'])  # noqa: E731
            fz, fimg, fuv, valid = t('z'), t('img'), t('uv'), t('valid')
            x = (2 * torch.arange(W, dtype=dtype) + 1 - W) / W
            y = (H - 2 * torch.arange(H, dtype=dtype) - 1.) / H
            pix = torch.stack([x.reshape(1, 1, -1).repeat(3, H, 1), y.reshape(1, -1, 1).repeat(3, 1, W)], dim=-1).reshape(3, -1, 2)
            zmin, zmax = fz.reshape(3, -1).min(1)[0], fz.reshape(3, -1).max(1)[0]
            ranges = torch.stack([zmin - 1e-2, zmax + 1e-2], dim=-1).unsqueeze(1).repeat(1, H * W, 1)
            for wv in (0, 1):
                kw = {'valid_faces': valid.bool()} if wv else {}
                feats, idx = _naive_deftet_sparse_render(pix, ranges, fz, fimg, fuv, 1, **kw)
                assert torch.equal(idx.reshape(3, H, W), t(f'valid{wv}_face_idx').long())
                assert torch.allclose(feats.reshape(3, H, W, 2), t(f'valid{wv}_feat'), rtol=1e-5, atol=1e-6)
            if dn == 'f64':
                a, u = fimg.clone().requires_grad_(), fuv.clone().requires_grad_()
                feats, _ = _naive_deftet_sparse_render(pix, ranges, fz, a, u, 1)
                feats.reshape(3, H, W, 2).backward(torch.from_numpy(gb[f'flip{flip}_grad_out']))
                assert torch.allclose(a.grad, torch.from_numpy(gb[f'flip{flip}_g_img']), rtol=1e-6, atol=1e-9)
                assert torch.allclose(u.grad, torch.from_numpy(gb[f'flip{flip}_g_uv']), rtol=1e-6, atol=1e-9)


def test_naive_deftet_renderer_multi_hit_order_and_lists():
    """knum > 1: nearest first, void slots -1 / zero features, list features split like the input."""
    from kaolin_amd.render.mesh.deftet import _naive_deftet_sparse_render
    img = torch.tensor([[[[-1., -1.], [1., -1.], [0., 1.]], [[-1., -1.], [1., -1.], [0., 1.]]]])
    z = torch.tensor([[[-3., -3., -3.], [-2., -2., -2.]]])
    feat = [torch.ones(1, 2, 3, 2), torch.arange(6.).reshape(1, 2, 3, 1)]
    pix = torch.tensor([[[0., 0.], [5., 5.]]])
    rng = torch.tensor([[[-10., 0.], [-10., 0.]]])
    (a, b), idx = _naive_deftet_sparse_render(pix, rng, z, img, feat, 3)
    assert idx.tolist() == [[[1, 0, -1], [-1, -1, -1]]]
    assert a.shape == (1, 2, 3, 2) and b.shape == (1, 2, 3, 1)
    assert torch.allclose(a[0, 0, :2], torch.ones(2, 2)) and float(a[0, 0, 2].abs().max()) == 0. and float(a[0, 1].abs().max()) == 0.


@pytest.mark.parametrize('tag', ['rand', 'sphere'])
@pytest.mark.parametrize('dn', ['f32', 'f64'])
def test_naive_point_to_mesh_distance_vs_reference_goldens(tag, dn):
    """_unbatched_naive_point_to_mesh_distance vs outputs and autograd gradients of the reference's function of that name."""
    from kaolin_amd.metrics import trianglemesh as tm
    g = np.load(os.path.join(GOLDEN_DIR, 'triangle_distance.npz'))
    k = f'{tag}_{dn}'
    t = lambda s: torch.from_numpy(g[f'{k}_{s}'])  # noqa: E731
    a, b = t('points').clone().requires_grad_(), t('faces').clone().requires_grad_()
    dist, idx, typ = tm._unbatched_naive_point_to_mesh_distance(a, b)
    tol = dict(rtol=1e-4, atol=1e-5) if dn == 'f32' else dict(rtol=1e-9, atol=1e-12)
    same = idx == t('idx')
    assert float(same.float().mean()) > 0.999                    # (ties between faces may resolve differently in fp32)
    assert torch.equal(typ[same], t('type')[same]) and typ.dtype == torch.int32 and idx.dtype == torch.long
    assert torch.allclose(dist, t('dist'), **tol)
    dist.backward(t('grad_out'))
    if bool(same.all()):
        assert torch.allclose(a.grad, t('g_points'), **tol) and torch.allclose(b.grad, t('g_faces'), **tol)
    # the CPU branch of the public function is this formulation (kaolin/metrics/trianglemesh.py:88-93)
    d2, i2, t2 = tm.point_to_mesh_distance(t('points')[None], t('faces')[None])
    assert torch.equal(i2[0], idx) and torch.equal(t2[0], typ) and torch.equal(d2[0], dist.detach())
    # and it agrees with the C oracle of the HIP kernel
    # (on a mesh the nearest point often lies on a shared edge / vertex: the face index is a tie, the distance is not)
    d3, i3, t3 = oracle.triangle_distance_forward(t('points'), t('faces'))
    assert torch.allclose(d3, dist.detach(), rtol=1e-4, atol=1e-5)   # (the kernel keeps `float dist` even for doubles)


def test_point_to_mesh_docstring_example_cpu():
    from kaolin_amd.metrics.trianglemesh import point_to_mesh_distance
    pts = torch.tensor([[[0.5, 0.5, 0.5], [3., 4., 5.]]])
    fv = torch.tensor([[[[0., 0., 0.], [0., 1., 0.], [0., 0., 1.]]]])
    d, i, t = point_to_mesh_distance(pts, fv)
    assert d.tolist() == [[0.25, 41.0]] and i.tolist() == [[0, 0]] and t.tolist() == [[5, 5]]


def test_sided_distance_torch_formulation():
    """_sided_distance (kaolin/metrics/pointcloud.py:186-197): values of the operator's first output."""
    from kaolin_amd.metrics.pointcloud import _sided_distance
    torch.manual_seed(0)
    p1, p2 = torch.rand(2, 300, 3), torch.rand(2, 257, 3)
    d_ref, _ = oracle.sided_distance_forward(p1, p2)
    assert torch.allclose(_sided_distance(p1, p2), d_ref, rtol=1e-5, atol=1e-7)


def test_mesh_metrics_known_answers():
    """average_edge_length / uniform_laplacian_smoothing: the reference's tables (test_trianglemesh.py:160-201)."""
    from kaolin_amd.metrics import trianglemesh as tm
    from kaolin_amd.ops.mesh import uniform_laplacian, adjacency_matrix
    for dtype in (torch.float, torch.double):
        v = torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]], [[3, 0, 0], [0, 4, 0], [0, 0, 5]]], dtype=dtype)
        f = torch.tensor([[0, 1, 2]])
        assert torch.allclose(tm.average_edge_length(v, f), torch.tensor([[1.4142], [5.7447]], dtype=dtype), atol=1e-4)
        v = torch.tensor([[[0, 0, 1], [2, 1, 2], [3, 1, 2]], [[3, 1, 2], [0, 0, 3], [0, 3, 3]]], dtype=dtype)
        want = torch.tensor([[[2.5, 1., 2.], [1.5, 0.5, 1.5], [1., 0.5, 1.5]], [[0., 1.5, 3.], [1.5, 2., 2.5], [1.5, 0.5, 2.5]]], dtype=dtype)
        assert torch.equal(tm.uniform_laplacian_smoothing(v, f), want)
    quad = torch.tensor([[0, 1, 2, 3]])
    assert adjacency_matrix(5, quad, sparse=False).tolist() == [[0, 1, 0, 1, 0], [1, 0, 1, 0, 0], [0, 1, 0, 1, 0], [1, 0, 1, 0, 0], [0] * 5]
    lap = uniform_laplacian(5, quad)
    assert torch.allclose(lap[0], torch.tensor([-1., .5, 0., .5, 0.])) and lap[4].tolist() == [0., 0., 0., 0., -1.]


def test_check_tensor_helper():
    from kaolin_amd.utils.testing import check_tensor
    t = torch.zeros(2, 3)
    assert check_tensor(t, shape=(2, None), dtype=torch.float, device='cpu')
    assert not check_tensor(t, shape=(3, 3), throw=False) and not check_tensor(t, dtype=torch.double, throw=False)
    with pytest.raises(ValueError):
        check_tensor(t, shape=(2, 3, 1))
    with pytest.raises(TypeError):
        check_tensor(t, dtype=torch.long)


def test_check_sign_cpu_branch_matches_the_kernel_oracle():
    """check_sign on CPU tensors (plain torch) answers like the restated CUDA kernel: a sphere, seeded points, plus points
    whose projection falls exactly on vertices / edges of the mesh (the double-count rules)."""
    from kaolin_amd.ops.mesh import check_sign
    from kaolin_amd.utils.testing import geodesic_sphere
    v, f = geodesic_sphere(3)
    for dtype in (torch.float, torch.double):
        verts = v.to(dtype)[None]
        g = torch.Generator().manual_seed(3)
        pts = (torch.rand(1, 4000, 3, generator=g, dtype=dtype) - 0.5) * 1.4
        on_vertices = verts[0, :200].clone()
        on_vertices[:, 0] -= 1.                                   # same (y, z) as a vertex, to its left
        mid = 0.5 * (verts[0, f[:100, 0]] + verts[0, f[:100, 1]])
        mid[:, 0] -= 1.                                           # same (y, z) as an edge midpoint
        pts = torch.cat([pts, on_vertices[None], mid[None]], dim=1)
        assert torch.equal(check_sign(verts, f, pts), oracle.check_sign(verts, f, pts))


def test_weighted_sum_cpu_fallback_is_the_torch_expression():
    from kaolin_amd.metrics.render import weighted_sum
    g = torch.Generator().manual_seed(5)
    x1, w1 = torch.rand(4, 5, generator=g, requires_grad=True), torch.rand(4, 5, generator=g)
    x2, w2 = torch.rand(7, generator=g), torch.rand(7, generator=g)
    out = weighted_sum(x1, w1, x2, w2)
    assert torch.allclose(out, (x1 * w1).sum() + (x2 * w2).sum())
    out.backward()
    assert torch.equal(x1.grad, w1)
    with pytest.raises(ValueError):
        weighted_sum(x1, w1, x2)


def test_faces_in_range_cache_pins_its_tensor():
    """the index-range check is cached per `faces` tensor by (data_ptr, shape, version): the entry must keep
    the tensor alive, otherwise a new mesh of the same shape allocated at the recycled address would inherit the answer."""
    import gc
    import weakref
    from kaolin_amd._C.render import mesh as m
    m._FACES_OK.clear()
    good = torch.tensor([[0, 1, 2], [2, 3, 1]])
    ref = weakref.ref(good)
    assert m.faces_in_range(good, 4) is True
    del good
    gc.collect()
    assert ref() is not None                       # pinned by the cache entry: its address cannot be reused
    bad = torch.tensor([[0, 1, 2], [2, 7, 1]])     # same shape, out of range
    assert bad.data_ptr() != ref().data_ptr()
    assert m.faces_in_range(bad, 4) is False and m.faces_in_range(bad, 8) is True
    neg = torch.tensor([[0, -1, 2]])
    assert m.faces_in_range(neg, 4) is False
    bad[1, 1] = 3                                   # in-place edit bumps the version: re-checked
    assert m.faces_in_range(bad, 4) is True


def test_topology_caches_hit_for_a_rewrapped_view():
    """the caches are probed by id(faces) first, but a caller that hands over a FRESH tensor object over the same
    storage every step (`faces[:]`, `.view(...)`) must still hit -- otherwise the range check (a reduction + a synchronising
    host read) and the adjacency (argsort, bincount, cumsum) are repeated each step."""
    from kaolin_amd._C.render import mesh as m
    m._FACES_OK.clear()
    m._ADJ_CACHE.clear()
    faces = torch.tensor([[0, 1, 2], [2, 3, 1], [0, 2, 3]])
    assert m.faces_in_range(faces, 4) is True
    off, ent, _ = m.vertex_face_adjacency(faces, 4)
    calls = []
    real = torch.aminmax
    torch.aminmax = lambda *a, **k: calls.append(1) or real(*a, **k)
    try:
        for view in (faces[:], faces.view(3, 3), faces.reshape(-1).view(3, 3)):
            assert view is not faces
            assert m.faces_in_range(view, 4) is True
            off2, ent2, _ = m.vertex_face_adjacency(view, 4)
            assert off2 is off and ent2 is ent              # the cached tensors, not a rebuild
        assert calls == []
        assert m.faces_in_range(faces[:2], 4) is True        # another shape over the same storage: its own entry
        assert calls == [1]
        faces[0, 0] = 3                                      # an edit through ANY view bumps the shared version
        assert m.faces_in_range(faces[:], 4) is True
        assert calls == [1, 1]
    finally:
        torch.aminmax = real
