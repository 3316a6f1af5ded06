"""Pins the CPU oracle's DIB-R restatement (oracle/dibr_oracle.inc + the glue in oracle/__init__.py)
against the reference: (i) the reference's pure-torch rasterize oracle on its own fixtures
(tests/golden/rasterize.npz), (ii) the CUDA-kernel goldens shipped with the reference
(tests/samples/dibr/*.pt, repacked in tests/golden/dibr_soft_mask.npz) with the tolerances of
tests/python/kaolin/render/mesh/test_dibr.py.  CPU only."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN_DIR

DT = {'f32': torch.float, 'f64': torch.double}


@pytest.fixture(scope='module')
def g_rast():
    return np.load(os.path.join(GOLDEN_DIR, 'rasterize.npz'))


@pytest.fixture(scope='module')
def g_dibr():
    return np.load(os.path.join(GOLDEN_DIR, 'dibr_soft_mask.npz'))


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('flip', [0, 1])
@pytest.mark.parametrize('with_valid', [0, 1])
@pytest.mark.parametrize('batch_size', [1, 3])
def test_rasterize_vs_reference_torch_oracle(g_rast, dn, flip, with_valid, batch_size):
    """test_rasterization.py:137-158: face_idx torch.equal, features rtol=atol=1e-5."""
    tag = f'{dn}_flip{flip}'
    t = lambda k: torch.from_numpy(g_rast[f'{tag}_{k}'])[:batch_size]  # noqa: E731
    valid = t('valid') if with_valid else None
    feats, face_idx, _ = oracle.rasterize(32, 32, t('z'), t('img'), t('uv'), valid)
    assert torch.equal(face_idx, t(f'valid{with_valid}_face_idx').long())
    assert torch.allclose(feats, t(f'valid{with_valid}_feat'), rtol=1e-5, atol=1e-5)


SIMPLE_IMG = [[[[-0.7, 0.], [0., -0.7], [0., 0.7]], [[-0.7, 0.], [0., 0.7], [0., -0.7]], [[0., -0.7], [0., 0.7], [0.7, 0.]]],
              [[[-0.7, -0.7], [0.7, -0.7], [-0.7, 0.7]], [[-0.7, -0.7], [0.7, -0.7], [-0.7, 0.7]],
               [[-0.7, -0.7], [0.7, -0.7], [-0.7, 0.7]]]]
SIMPLE_Z = [[[-2., -1., -1.], [-2.5, -3., -3.], [-2., -2., -2.]], [[-2., -1., -3.], [-2., -2., -2.], [-2., -3., -1.]]]


def simple_inputs(dtype):
    img, z = torch.tensor(SIMPLE_IMG, dtype=dtype), torch.tensor(SIMPLE_Z, dtype=dtype)
    _, face_idx, _ = oracle.rasterize(35, 31, z, img, torch.zeros(z.shape + (1,), dtype=dtype))
    return img, face_idx


def test_simple_face_idx_golden(g_dibr):
    """tests/samples/dibr/simple/new_face_idx_35_31.pt (int64 face_idx of the 2x3 hand-written triangles)."""
    for dtype in (torch.float, torch.double):
        _, face_idx = simple_inputs(dtype)
        assert torch.equal(face_idx, torch.from_numpy(g_dibr['simple_new_face_idx']).long())


def _mask_iou(a, b):
    inter = (a * b).flatten(1)
    union = (a + b).flatten(1) - inter
    return 1. - (inter.sum(1) / (union.sum(1) + 1e-10)).mean()


def _check_forward_backward(g_dibr, tag, img, face_idx, sigmainv, boxlen, knum, multiplier, dtype, sphere):
    bs = img.shape[0]
    soft, prob, idx, typ, simg = oracle.dibr_soft_mask(img, face_idx, sigmainv, boxlen, knum, float(multiplier))
    gt = lambda k: torch.from_numpy(g_dibr[f'{tag}_{k}'])[:bs]  # noqa: E731
    assert torch.allclose(soft, gt('soft_mask').to(dtype), atol=1e-5, rtol=1e-5)
    kk = min(knum, gt('idx').shape[-1])
    assert torch.equal(idx[..., :kk], gt('idx')[..., :kk].long())
    assert torch.allclose(prob[..., :kk], gt('prob')[..., :kk].to(dtype), atol=1e-5, rtol=1e-5)
    if sphere:   # test_dibr.py:339-341: <= 1 % of dist_type may differ
        assert float((typ[..., :kk] != gt('type')[..., :kk]).float().mean()) <= 0.01
    else:
        assert torch.equal(typ[..., :kk], gt('type')[..., :kk])
    # backward through mask_iou against the shifted silhouette (test_dibr.py:172-191, :375-394)
    s = soft.clone().requires_grad_()
    mask = (face_idx != -1)
    shifted = torch.nn.functional.pad(mask, (0, 5))[..., 5:]
    _mask_iou(s, shifted.to(dtype)).backward()
    g_img = oracle.dibr_soft_mask_backward(s.grad, soft, face_idx, prob, idx, typ, simg, sigmainv, float(multiplier))
    return g_img, gt('grad').to(dtype)


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('sigmainv', [7000, 70])
@pytest.mark.parametrize('boxlen', [0.02, 0.2])
@pytest.mark.parametrize('multiplier', [1000, 100, 1])
@pytest.mark.parametrize('knum', [30, 20])
def test_simple_vs_cuda_goldens(g_dibr, dtype, sigmainv, boxlen, multiplier, knum):
    img, face_idx = simple_inputs(dtype)
    g, gt = _check_forward_backward(g_dibr, f'simple_{sigmainv}_{boxlen}', img, face_idx, sigmainv, boxlen, knum,
                                    multiplier, dtype, False)
    assert torch.allclose(g, gt, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('flip', [0, 1])
@pytest.mark.parametrize('sigmainv', [7000, 70])
@pytest.mark.parametrize('boxlen', [0.02, 0.01])
@pytest.mark.parametrize('multiplier,knum,batch_size', [(1000, 30, 3), (100, 40, 1)])
def test_sphere_vs_cuda_goldens(g_dibr, dn, flip, sigmainv, boxlen, multiplier, knum, batch_size):
    dtype = DT[dn]
    img = torch.from_numpy(g_dibr[f'sphere_in_{dn}_flip{flip}_img'])[:batch_size]
    z = torch.from_numpy(g_dibr[f'sphere_in_{dn}_flip{flip}_z'])[:batch_size]
    _, face_idx, _ = oracle.rasterize(35, 31, z, img, torch.zeros(z.shape + (1,), dtype=dtype))
    g, gt = _check_forward_backward(g_dibr, f'sphere_{sigmainv}_{boxlen}', img, face_idx, sigmainv, boxlen, knum,
                                    multiplier, dtype, True)
    if flip:   # the stored gradient is w.r.t. the un-flipped vertex order
        g = torch.flip(g, dims=(2,))
    # the goldens were made with batch 3 (mask_iou averages over the batch)
    gt = gt * (3. / batch_size)
    # test_dibr.py:392-394 allows 1e-1 here; the oracle is far tighter
    assert torch.allclose(g, gt, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize('flip', [0, 1])
def test_rasterize_backward_vs_reference_autograd(g_rast, flip):
    """K2 restatement vs autograd through the reference's torch oracle (golden rasterize_backward.npz);
    reference tolerances are rtol 1e-3 / atol 1e-2 (vertices) and 1e-3 (features),
    test_rasterization.py:228-233 -- in float64 the analytic gradient agrees far better."""
    g = np.load(os.path.join(GOLDEN_DIR, 'rasterize_backward.npz'))
    z = torch.from_numpy(g_rast[f'f64_flip{flip}_z'])
    img = torch.from_numpy(g_rast[f'f64_flip{flip}_img'])
    uv = torch.from_numpy(g_rast[f'f64_flip{flip}_uv'])
    feats, face_idx, wts = oracle.rasterize(32, 32, z, img, uv)
    g_img, g_uv = oracle.rasterize_backward(torch.from_numpy(g[f'flip{flip}_grad_out']), face_idx, wts, img, uv, 1e-8)
    # feature grads = grad*w: the torch oracle normalises w in UNSCALED coordinates where eps=1e-8 is ~1e-5 of
    # the edge-function sum, the kernel in x1000 coordinates where it is negligible -> 1e-4 absolute
    assert torch.allclose(g_uv, torch.from_numpy(g[f'flip{flip}_g_uv']), rtol=1e-4, atol=1e-4)
    assert torch.allclose(g_img, torch.from_numpy(g[f'flip{flip}_g_img']), rtol=1e-5, atol=1e-7)
