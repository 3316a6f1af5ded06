"""mask_iou and texture_mapping: the fused HIP paths against the torch op chains that define them
(kaolin/metrics/render.py:18-40, kaolin/render/mesh/utils.py:23-76), values and gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('shape', [(1, 7, 5), (3, 64, 48), (8, 1024, 1024)])
def test_mask_iou_fused_matches_torch(dtype, shape):
    from kaolin_amd.metrics.render import mask_iou, _mask_iou_torch
    g = torch.Generator().manual_seed(0)
    a = torch.rand(shape, generator=g, dtype=dtype).cuda()
    b = (torch.rand(shape, generator=g, dtype=dtype) > 0.5).to(dtype).cuda()
    x, y = a.clone().requires_grad_(), b.clone().requires_grad_()
    loss = mask_iou(x, y)
    (loss * 3.).backward()
    xr, yr = a.double().clone().requires_grad_(), b.double().clone().requires_grad_()
    ref = _mask_iou_torch(xr, yr)
    (ref * 3.).backward()
    tol = 1e-5 if dtype == torch.float else 1e-12
    assert loss.dtype == dtype and loss.shape == ()
    assert abs(float(loss) - float(ref)) <= tol * max(abs(float(ref)), 1e-3)
    assert float((x.grad.double() - xr.grad).abs().max()) <= tol * float(xr.grad.abs().max())
    assert float((y.grad.double() - yr.grad).abs().max()) <= tol * float(yr.grad.abs().max())
    # the reference's known answer (tests/python/kaolin/metrics/test_render.py): identical masks -> 0
    assert abs(float(mask_iou(b, b))) < 1e-6


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('mode', ['nearest', 'bilinear'])
@pytest.mark.parametrize('dense', [True, False])
def test_texture_mapping_fused_matches_torch(dtype, mode, dense):
    from kaolin_amd.render.mesh.utils import texture_mapping, _texture_mapping_torch
    g = torch.Generator().manual_seed(1)
    B, C, th, tw = 2, 3, 17, 23
    tex = torch.rand((B, C, th, tw), generator=g, dtype=dtype).cuda()
    shape = (B, 31, 29, 2) if dense else (B, 500, 2)
    uv = (torch.rand(shape, generator=g, dtype=dtype) * 1.4 - 0.2).cuda()      # some coordinates outside [0, 1]
    uv.view(-1, 2)[:4] = torch.tensor([[0., 0.], [1., 1.], [0.5, 0.5], [1., 0.]], dtype=dtype)
    t1, u1 = tex.clone().requires_grad_(), uv.clone().requires_grad_()
    t2, u2 = tex.clone().requires_grad_(), uv.clone().requires_grad_()
    out = texture_mapping(u1, t1, mode)
    ref = _texture_mapping_torch(u2, t2, mode)
    assert out.shape == ref.shape == shape[:-1] + (C,)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float else dict(rtol=1e-12, atol=1e-13)
    assert torch.allclose(out, ref, **tol)
    w = torch.rand(out.shape, generator=g, dtype=dtype).cuda()
    (out * w).sum().backward()
    (ref * w).sum().backward()
    assert torch.allclose(t1.grad, t2.grad, **(dict(rtol=1e-4, atol=1e-5) if dtype == torch.float else tol))
    if mode == 'bilinear':
        assert torch.allclose(u1.grad, u2.grad, **(dict(rtol=1e-3, atol=1e-4) if dtype == torch.float else dict(rtol=1e-9, atol=1e-10)))
    else:
        assert float(u1.grad.abs().max()) == 0.


def test_cpu_inputs_take_the_torch_chain():
    from kaolin_amd.metrics.render import mask_iou
    from kaolin_amd.render.mesh.utils import texture_mapping
    a = torch.rand(2, 5, 5)
    assert abs(float(mask_iou(a, a)) - (1. - float(((a * a).flatten(1).sum(1) / ((2 * a - a * a).flatten(1).sum(1) + 1e-10)).mean()))) < 1e-6
    out = texture_mapping(torch.rand(1, 4, 2), torch.rand(1, 3, 8, 8))
    assert out.shape == (1, 4, 3)
