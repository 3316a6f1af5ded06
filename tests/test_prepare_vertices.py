"""prepare_vertices: the fused HIP path against the torch op chain that defines it
(kaolin/render/mesh/utils.py:128-175), forward and all three gradient inputs, both camera parametrisations."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dtype, B, shared, transform):
    import kaolin_amd as kal
    from kaolin_amd.utils.testing import geodesic_sphere, fibonacci_cameras
    v, f = geodesic_sphere(6)
    v = v.to(dtype).cuda()
    f = f.cuda()
    cams = fibonacci_cameras(B, 2.5, dtype).cuda()
    up = torch.tensor([[0., 1., 0.]], dtype=dtype, device='cuda').repeat(B, 1)
    rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams, torch.zeros_like(cams), up)
    proj = kal.render.camera.generate_perspective_projection(math.pi / 4, dtype=dtype).cuda()
    if shared:
        verts = v.unsqueeze(0)
    else:
        verts = v.unsqueeze(0).repeat(B, 1, 1) + torch.randn(B, v.shape[0], 3, dtype=dtype, device='cuda') * 0.01
    kw = {'camera_rot': rot, 'camera_trans': trans}
    if transform:
        M = torch.cat([rot.transpose(1, 2), -(rot @ trans.unsqueeze(-1)).transpose(1, 2)], dim=1)  # (B,4,3): [v,1] M = R(v-t)
        kw = {'camera_transform': M}
    return kal, verts, f, proj, kw


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('shared', [True, False])
@pytest.mark.parametrize('transform', [False, True])
def test_fused_matches_torch_chain(dtype, shared, transform):
    kal, verts, f, proj, kw = _setup(dtype, 3, shared, transform)
    from kaolin_amd.render.mesh.utils import _prepare_vertices_torch
    inp = verts.expand(3, -1, -1) if shared else verts
    a = inp.clone().requires_grad_() if not shared else None
    base = verts.clone().requires_grad_()
    x = (base.expand(3, -1, -1) if shared else base)
    out = kal.render.mesh.prepare_vertices(x, f, proj, **kw)
    ref_base = verts.clone().requires_grad_()
    y = (ref_base.expand(3, -1, -1) if shared else ref_base)
    ref = _prepare_vertices_torch(y, f, proj, **kw)
    tol = 2e-5 if dtype == torch.float else 1e-11
    for o, r in zip(out, ref):
        assert o.shape == r.shape and float((o - r).abs().max()) <= tol * max(float(r.abs().max()), 1.0)
    torch.manual_seed(0)
    gs = [torch.rand_like(r) for r in ref]
    sum((o * g).sum() for o, g in zip(out, gs)).backward()
    sum((r * g).sum() for r, g in zip(ref, gs)).backward()
    scale = float(ref_base.grad.abs().max())
    assert float((base.grad - ref_base.grad).abs().max()) <= (5e-4 if dtype == torch.float else 1e-10) * scale
    # only the image-plane gradient (what DIB-R sends back)
    base.grad = None
    ref_base.grad = None
    out = kal.render.mesh.prepare_vertices((base.expand(3, -1, -1) if shared else base), f, proj, **kw)
    (out[1] * gs[1]).sum().backward()
    ref = _prepare_vertices_torch((ref_base.expand(3, -1, -1) if shared else ref_base), f, proj, **kw)
    (ref[1] * gs[1]).sum().backward()
    assert float((base.grad - ref_base.grad).abs().max()) <= (5e-4 if dtype == torch.float else 1e-10) * float(ref_base.grad.abs().max())


def test_gradcheck_double():
    kal, verts, f, proj, kw = _setup(torch.double, 2, False, False)
    verts = verts[:, :40].clone().requires_grad_()
    faces = torch.randint(0, 40, (30, 3), device='cuda')
    fn = lambda v: kal.render.mesh.prepare_vertices(v, faces, proj, **kw)  # noqa: E731
    assert torch.autograd.gradcheck(fn, (verts,), eps=1e-6, atol=1e-6)


def test_falls_back_when_camera_needs_grad():
    kal, verts, f, proj, kw = _setup(torch.float, 2, True, False)
    kw['camera_trans'] = kw['camera_trans'].clone().requires_grad_()
    out = kal.render.mesh.prepare_vertices(verts.expand(2, -1, -1), f, proj, **kw)
    out[1].sum().backward()
    assert kw['camera_trans'].grad is not None


def test_fused_path_only_for_shapes_it_reads():
    """a batched projection (B, 3, 1), a broadcast (1, 3) translation, a single camera for batched vertices and
    out-of-range face indices must not reach the fused kernels (fixed strides, unchecked gathers): they take the torch
    chain, which broadcasts / raises like the reference."""
    kal, verts, f, proj, kw = _setup(torch.float, 3, True, False)
    from kaolin_amd.render.mesh.utils import _fusable, _prepare_vertices_torch, prepare_vertices
    rot, trans = kw['camera_rot'], kw['camera_trans']
    assert _fusable(verts, f, proj, rot, trans, None)
    # batched projection: every view has its own focal terms
    projs = torch.stack([proj.reshape(-1) * s for s in (1.0, 1.5, 2.0)]).reshape(3, 3, 1)
    assert not _fusable(verts.expand(3, -1, -1), f, projs, rot, trans, None)
    got = prepare_vertices(verts.expand(3, -1, -1), f, projs, camera_rot=rot, camera_trans=trans)
    want = _prepare_vertices_torch(verts.expand(3, -1, -1), f, projs, camera_rot=rot, camera_trans=trans)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    assert not torch.allclose(got[1][0], got[1][1])
    # broadcast translation
    assert not _fusable(verts.expand(3, -1, -1), f, proj, rot, trans[:1], None)
    got = prepare_vertices(verts.expand(3, -1, -1), f, proj, camera_rot=rot, camera_trans=trans[:1])
    want = _prepare_vertices_torch(verts.expand(3, -1, -1), f, proj, camera_rot=rot, camera_trans=trans[:1])
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    # one camera, batched vertices
    assert not _fusable(verts.repeat(2, 1, 1), f, proj, rot[:1], trans[:1], None)
    # a face that points past the vertex list
    bad = f.clone()
    bad[0, 0] = verts.shape[1]
    assert not _fusable(verts, bad, proj, rot, trans, None)
    with pytest.raises(IndexError):
        prepare_vertices(verts.expand(3, -1, -1), bad, proj, camera_rot=rot, camera_trans=trans)
