"""DIB-R on the GPU: the HIP path through the C ABI (kaolin_amd._C.render.mesh.*) against
(i) the reference's own golden vectors (tests/golden/*.npz, see make_golden.py) and
(ii) the CPU oracle on seeded synthetic scenes.  face_idx / close_face_idx / dist_type bit-exact,
floats within 1e-5 relative (the tolerance BASELINE.json's north_star states)."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN_DIR
from test_dibr_oracle import SIMPLE_IMG, SIMPLE_Z, _mask_iou

pytestmark = pytest.mark.gpu
DT = {'f32': torch.float, 'f64': torch.double}


def kal():
    import kaolin_amd
    return kaolin_amd


@pytest.fixture(scope='module')
def g_rast():
    return np.load(os.path.join(GOLDEN_DIR, 'rasterize.npz'))


@pytest.fixture(scope='module')
def g_dibr():
    return np.load(os.path.join(GOLDEN_DIR, 'dibr_soft_mask.npz'))


def rel_close(a, b, tol=1e-5, term_abs_sum=None):
    """ELEMENT-WISE: |a - b| <= tol |b| + tol median|b != 0| (kaolin_amd.utils.testing.elementwise_mismatch; through round 3
    this scaled the tolerance by the largest element of `b`, which let small entries be off by orders of magnitude).
    `term_abs_sum`: for sums of many float terms, + 64 eps * the sum of the terms' magnitudes (see there)."""
    from kaolin_amd.utils.testing import elementwise_mismatch
    msg = elementwise_mismatch(a, b, tol, term_abs_sum=term_abs_sum)
    if term_abs_sum is not None:
        used, of = elementwise_mismatch.last_slack_use
        print(f'[term_abs_sum slack] {used} of {of} elements pass only with the 64 eps sum|terms| accumulation slack')
    assert msg is None, msg
    return True


def same_sum_other_order(a, b, tol):
    """Two GPU paths that add the SAME float terms in a different (atomic) order: compared against the largest element -- an
    element whose terms cancel carries the rounding noise of their magnitude in either path (oracle comparisons use the
    element-wise rel_close above)."""
    a, b = a.double().cpu(), b.double().cpu()
    scale = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) <= tol * scale


# ------------------------------------------------------------------ rasterize vs the reference's goldens
@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('flip', [0, 1])
@pytest.mark.parametrize('with_valid', [0, 1])
@pytest.mark.parametrize('batch_size', [1, 3])
def test_rasterize_vs_reference_golden(g_rast, dn, flip, with_valid, batch_size):
    tag = f'{dn}_flip{flip}'
    t = lambda k: torch.from_numpy(g_rast[f'{tag}_{k}'])[:batch_size].cuda()  # noqa: E731
    kw = {'valid_faces': t('valid')} if with_valid else {}
    feats, face_idx = kal().render.mesh.rasterize(32, 32, t('z'), t('img'), t('uv'), **kw)
    assert torch.equal(face_idx, t(f'valid{with_valid}_face_idx').long())
    assert torch.allclose(feats, t(f'valid{with_valid}_feat'), rtol=1e-5, atol=1e-5)
    # list features (test_rasterization.py:160-187)
    (uv, ones), face_idx2 = kal().render.mesh.rasterize(32, 32, t('z'), t('img'), [t('uv'), torch.ones_like(t('uv')[..., 1:])], **kw)
    assert torch.equal(face_idx2, face_idx) and torch.equal(uv, feats)
    assert torch.allclose(ones, (face_idx >= 0).to(ones.dtype).unsqueeze(-1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('flip', [0, 1])
def test_rasterize_backward_vs_reference_autograd(g_rast, flip):
    g = np.load(os.path.join(GOLDEN_DIR, 'rasterize_backward.npz'))
    z = torch.from_numpy(g_rast[f'f64_flip{flip}_z']).cuda()
    img = torch.from_numpy(g_rast[f'f64_flip{flip}_img']).cuda().requires_grad_()
    uv = torch.from_numpy(g_rast[f'f64_flip{flip}_uv']).cuda().requires_grad_()
    zz = z.clone().requires_grad_()
    feats, _ = kal().render.mesh.rasterize(32, 32, zz, img, uv)
    feats.backward(torch.from_numpy(g[f'flip{flip}_grad_out']).cuda())
    assert zz.grad is None or bool((zz.grad == 0).all())
    assert torch.allclose(uv.grad.cpu(), torch.from_numpy(g[f'flip{flip}_g_uv']), rtol=1e-4, atol=1e-4)
    assert torch.allclose(img.grad.cpu(), torch.from_numpy(g[f'flip{flip}_g_img']), rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ rasterize vs the oracle, synthetic scenes
def _scene(level, views, dtype, seed=0):
    from kaolin_amd.utils import testing as T
    return T.sphere_scene(level=level, num_views=views, dtype=dtype, seed=seed)


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('level,views,H,W', [(16, 1, 256, 256), (6, 3, 35, 31), (4, 2, 7, 5), (10, 2, 100, 180)])
def test_rasterize_forward_backward_vs_oracle(dtype, level, views, H, W):
    """C2 (5120 faces, 256^2) and ragged sizes: sel/face_idx bit-exact, weights & features bit-exact
    (same IEEE operations in the same order), gradients 1e-5 relative (atomic summation order differs)."""
    fz, fimg, feats, nz = _scene(level, views, dtype)
    feat = torch.cat(feats, -1)
    valid = nz >= 0
    r_feat, r_idx, r_w = oracle.rasterize(H, W, fz, fimg, feat, valid, omp=True)
    a = fimg.cuda().requires_grad_()
    f = feat.cuda().requires_grad_()
    out, face_idx = kal().render.mesh.rasterize(H, W, fz.cuda(), a, f, valid.cuda())
    assert torch.equal(face_idx.cpu(), r_idx)
    assert torch.equal(out.detach().cpu(), r_feat)
    torch.manual_seed(1)
    g = torch.rand(out.shape, dtype=dtype)
    out.backward(g.cuda())
    g_img, g_feat = oracle.rasterize_backward(g, r_idx, r_w, fimg, feat, 1e-8)
    assert rel_close(a.grad, g_img) and rel_close(f.grad, g_feat)


def test_rasterize_edge_cases():
    """no valid face at all; tiny images (B*H*W < 512, where the reference's backward launches 0 blocks);
    a face list that exceeds one LDS round (> 512 faces in one tile); coincident faces (ties -> lowest index)."""
    m = kal().render.mesh
    fz, fimg, feats, nz = _scene(4, 2, torch.float)
    feat = torch.cat(feats, -1).cuda()
    none = torch.zeros(nz.shape, dtype=torch.bool).cuda()
    out, idx = m.rasterize(16, 16, fz.cuda(), fimg.cuda(), feat, none)
    assert int(idx.max()) == -1 and float(out.abs().max()) == 0.
    a = fimg.cuda().requires_grad_()
    out, idx = m.rasterize(3, 5, fz.cuda(), a, feat)
    out.sum().backward()
    r_feat, r_idx, r_w = oracle.rasterize(3, 5, fz, fimg, feat)
    assert torch.equal(idx.cpu(), r_idx)
    assert rel_close(a.grad, oracle.rasterize_backward(torch.ones_like(r_feat), r_idx, r_w, fimg, feat.cpu(), 1e-8)[0])
    # 20480 faces on a 32x32 image: one tile holds thousands of faces
    fz, fimg, feats, nz = _scene(32, 1, torch.float)
    feat = torch.cat(feats, -1)
    out, idx = m.rasterize(32, 32, fz.cuda(), fimg.cuda(), feat.cuda())
    r_feat, r_idx, _ = oracle.rasterize(32, 32, fz, fimg, feat, omp=True)
    assert torch.equal(idx.cpu(), r_idx) and torch.equal(out.cpu(), r_feat)
    # duplicated mesh: every pixel has an exact depth tie between face f and f + F -> lowest index wins
    fz, fimg, feats, nz = _scene(6, 1, torch.float)
    F = fz.shape[1]
    out, idx = m.rasterize(64, 64, fz.repeat(1, 2, 1).cuda(), fimg.repeat(1, 2, 1, 1).cuda(),
                           torch.cat(feats, -1).repeat(1, 2, 1, 1).cuda())
    assert int(idx.max()) < F and int(idx.max()) >= 0


def _adversarial_faces(seed, n, dtype):
    """Faces built to sit where the conservative edge test of the tile kernel decides: slivers, near-zero and zero areas,
    vertices on pixel centres (edge functions exactly 0), shared edges through pixel centres, both windings, faces far
    larger than the image, duplicated vertices."""
    g = torch.Generator().manual_seed(seed)
    H = W = 48
    px = (2 * torch.arange(W, dtype=torch.float64) + 1 - W) / W          # pixel centres in NDC (x); y is the mirror image
    pick = lambda k: px[torch.randint(0, W, (k,), generator=g)]
    faces = []
    for i in range(n):
        kind = i % 8
        if kind == 0:     # all three vertices on pixel centres: edges run exactly through centres
            v = torch.stack([torch.stack([pick(1)[0], -pick(1)[0]]) for _ in range(3)])
        elif kind == 1:   # sliver: third vertex almost on the line through the first two
            a, b = torch.rand(2, generator=g, dtype=torch.float64) * 2 - 1, torch.rand(2, generator=g, dtype=torch.float64) * 2 - 1
            t = torch.rand(1, generator=g, dtype=torch.float64)
            v = torch.stack([a, b, a + t * (b - a) + (torch.rand(2, generator=g, dtype=torch.float64) - 0.5) * 1e-6])
        elif kind == 2:   # exactly degenerate: two equal vertices
            a, b = torch.rand(2, generator=g, dtype=torch.float64) * 2 - 1, torch.rand(2, generator=g, dtype=torch.float64) * 2 - 1
            v = torch.stack([a, b, b.clone()])
        elif kind == 3:   # far larger than the image
            v = (torch.rand((3, 2), generator=g, dtype=torch.float64) - 0.5) * 40
        elif kind == 4:   # about one pixel, anywhere (sub-pixel offsets)
            c = torch.rand(2, generator=g, dtype=torch.float64) * 2 - 1
            v = c + (torch.rand((3, 2), generator=g, dtype=torch.float64) - 0.5) * (3.0 / W)
        elif kind == 5:   # axis-aligned right triangle whose legs lie on pixel-centre lines
            x0, x1, y0, y1 = pick(1)[0], pick(1)[0], -pick(1)[0], -pick(1)[0]
            v = torch.stack([torch.stack([x0, y0]), torch.stack([x1, y0]), torch.stack([x0, y1])])
        elif kind == 6:   # ordinary
            v = torch.rand((3, 2), generator=g, dtype=torch.float64) * 2 - 1
        else:             # the previous face with the opposite winding (shared edges, equal depths elsewhere)
            v = faces[-1].flip(0)
        faces.append(v)
    img = torch.stack(faces).unsqueeze(0).to(dtype)                       # (1, n, 3, 2)
    z = -(torch.rand((1, n, 3), generator=g, dtype=torch.float64) + 1)
    z[0, 3::8] -= 1.5                                                     # the huge faces lie behind the others
    z = z.to(dtype)
    feat = torch.rand((1, n, 3, 2), generator=g, dtype=torch.float64).to(dtype)
    return H, W, z, img, feat


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('multiplier,eps', [(1000, 1e-8), (1, 1e-8), (1000, 0.5), (1000, 2.0), (1e5, 1e-8), (1000, 1e-30)])
@pytest.mark.parametrize('seed', [0, 1])
def test_rasterize_adversarial_faces_bit_exact(dtype, multiplier, eps, seed):
    """The fp32 tile kernel drops (pixel, face) pairs early with conservative affine edge functions (tile_lists.h,
    edge_coefficients) and repeats the reference's arithmetic for the rest: face_idx, weights and features must stay
    bit-identical to the oracle where that early test is closest to wrong -- and for an eps its bounds do not cover."""
    H, W, z, img, feat = _adversarial_faces(seed, 400, dtype)
    r_feat, r_idx, r_w = oracle.rasterize(H, W, z, img, feat, None, multiplier=multiplier, eps=eps, omp=True)
    out, face_idx = kal().render.mesh.rasterize(H, W, z.cuda(), img.cuda(), feat.cuda(), None, multiplier=multiplier, eps=eps)
    assert torch.equal(face_idx.cpu(), r_idx)
    assert torch.equal(out.cpu(), r_feat)


# ------------------------------------------------------------------ soft mask vs the CUDA goldens
def _simple(dtype):
    img = torch.tensor(SIMPLE_IMG, dtype=dtype).cuda()
    z = torch.tensor(SIMPLE_Z, dtype=dtype).cuda()
    _, face_idx = kal().render.mesh.rasterize(35, 31, z, img, torch.zeros(z.shape + (1,), dtype=dtype, device='cuda'))
    return img, face_idx


def _c_forward(img, face_idx, sigmainv, boxlen, knum, multiplier):
    scaled = img * multiplier
    lo, hi = scaled.min(dim=-2)[0], scaled.max(dim=-2)[0]
    bbox = torch.cat([lo - boxlen * multiplier, hi + boxlen * multiplier], dim=-1)
    return kal()._C.render.mesh.dibr_soft_mask_forward_cuda(scaled, bbox, face_idx, sigmainv, knum, multiplier)


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('sigmainv', [7000, 70])
@pytest.mark.parametrize('boxlen', [0.02, 0.2])
@pytest.mark.parametrize('multiplier', [1000, 100, 1])
@pytest.mark.parametrize('knum', [30, 20])
def test_simple_soft_mask_vs_cuda_goldens(g_dibr, dtype, sigmainv, boxlen, multiplier, knum):
    """test_dibr.py:111-191 (forward at the _C level, forward and backward at the API level)."""
    img, face_idx = _simple(dtype)
    assert torch.equal(face_idx.cpu(), torch.from_numpy(g_dibr['simple_new_face_idx']).long())
    tag = f'simple_{sigmainv}_{boxlen}'
    gt = lambda k: torch.from_numpy(g_dibr[f'{tag}_{k}']).cuda()  # noqa: E731
    soft, prob, idx, typ = _c_forward(img, face_idx, sigmainv, boxlen, knum, multiplier)
    assert torch.allclose(soft, gt('soft_mask').to(dtype), atol=1e-5, rtol=1e-5)
    assert torch.equal(idx, gt('idx')[..., :knum].long())
    assert torch.allclose(prob, gt('prob')[..., :knum].to(dtype), atol=1e-5, rtol=1e-5)
    assert torch.equal(typ, gt('type')[..., :knum])
    a = img.detach().requires_grad_()
    soft2 = kal().render.mesh.dibr_soft_mask(a, face_idx, sigmainv, boxlen, knum, multiplier)
    assert torch.equal(soft2, soft)
    shifted = torch.nn.functional.pad(face_idx != -1, (0, 5))[..., 5:]
    kal().metrics.render.mask_iou(soft2, shifted.to(dtype)).backward()
    assert torch.allclose(a.grad, gt('grad').to(dtype), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('dn', ['f32', 'f64'])
@pytest.mark.parametrize('flip', [0, 1])
@pytest.mark.parametrize('sigmainv', [7000, 70])
@pytest.mark.parametrize('boxlen', [0.02, 0.01])
@pytest.mark.parametrize('multiplier,knum,batch_size', [(1000, 30, 3), (100, 40, 1)])
def test_sphere_soft_mask_vs_cuda_goldens(g_dibr, dn, flip, sigmainv, boxlen, multiplier, knum, batch_size):
    """test_dibr.py:304-394 with its tolerances (dist_type <= 1 % mismatch; the reference allows 1e-1 on
    the gradient, we hold 1e-3)."""
    dtype = DT[dn]
    img = torch.from_numpy(g_dibr[f'sphere_in_{dn}_flip{flip}_img'])[:batch_size].cuda()
    z = torch.from_numpy(g_dibr[f'sphere_in_{dn}_flip{flip}_z'])[:batch_size].cuda()
    _, face_idx = kal().render.mesh.rasterize(35, 31, z, img, torch.zeros(z.shape + (1,), dtype=dtype, device='cuda'))
    tag = f'sphere_{sigmainv}_{boxlen}'
    gt = lambda k: torch.from_numpy(g_dibr[f'{tag}_{k}'])[:batch_size].cuda()  # noqa: E731
    soft, prob, idx, typ = _c_forward(img, face_idx, sigmainv, boxlen, knum, multiplier)
    kk = min(knum, 40)
    assert torch.allclose(soft, gt('soft_mask').to(dtype), atol=1e-5, rtol=1e-5)
    assert torch.equal(idx[..., :kk], gt('idx')[..., :kk].long())
    assert torch.allclose(prob[..., :kk], gt('prob')[..., :kk].to(dtype), atol=1e-5, rtol=1e-5)
    assert float((typ[..., :kk] != gt('type')[..., :kk]).float().mean()) <= 0.01
    a = img.detach().requires_grad_()
    soft2 = kal().render.mesh.dibr_soft_mask(a, face_idx, sigmainv, boxlen, knum, multiplier)
    shifted = torch.nn.functional.pad(face_idx != -1, (0, 5))[..., 5:]
    kal().metrics.render.mask_iou(soft2, shifted.to(dtype)).backward()
    g = torch.flip(a.grad, dims=(2,)) if flip else a.grad
    assert torch.allclose(g, gt('grad').to(dtype) * (3. / batch_size), rtol=1e-3, atol=1e-5)


# ------------------------------------------------------------------ soft mask + composition vs the oracle
@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('level,views,H,W,knum,boxlen', [(16, 1, 256, 256, 30, 0.02), (6, 3, 35, 31, 30, 0.2),
                                                         (8, 2, 64, 96, 5, 0.1)])
def test_dibr_rasterization_vs_oracle(dtype, level, views, H, W, knum, boxlen):
    fz, fimg, feats, nz = _scene(level, views, dtype)
    ref = oracle.dibr_rasterization(H, W, fz, fimg, torch.cat(feats, -1), nz, boxlen=boxlen, knum=knum, omp=True)
    a = fimg.cuda().requires_grad_()
    f = [x.cuda().requires_grad_() for x in feats]
    out, soft, face_idx = kal().render.mesh.dibr_rasterization(H, W, fz.cuda(), a, f, nz.cuda(), boxlen=boxlen, knum=knum)
    assert torch.equal(face_idx.cpu(), ref['face_idx'])
    assert torch.equal(torch.cat(out, -1).detach().cpu(), ref['features'])
    assert rel_close(soft.detach(), ref['soft_mask'])
    # the op-level K-buffers
    s2, prob, idx, typ = _c_forward(fimg.cuda(), face_idx, 7000, boxlen, knum, 1000.)
    assert torch.equal(s2, soft.detach())
    assert torch.equal(idx.cpu(), ref['close_face_idx']) and torch.equal(typ.cpu(), ref['close_face_dist_type'])
    assert rel_close(prob, ref['close_face_prob'])
    # composition identity (test_dibr.py:495-529): bit-identical to rasterize + dibr_soft_mask
    out2, idx2 = kal().render.mesh.rasterize(H, W, fz.cuda(), fimg.cuda(), [x.cuda() for x in feats], (nz >= 0).cuda())
    soft3 = kal().render.mesh.dibr_soft_mask(fimg.cuda(), idx2, 7000, boxlen, knum, 1000.)
    assert torch.equal(idx2, face_idx) and torch.equal(soft3, soft.detach()) and torch.equal(torch.cat(out2, -1), torch.cat(out, -1).detach())
    # backward of both branches
    torch.manual_seed(2)
    g1 = torch.rand(ref['features'].shape, dtype=dtype)
    g2 = torch.rand(ref['soft_mask'].shape, dtype=dtype)
    ((torch.cat(out, -1) * g1.cuda()).sum() + (soft * g2.cuda()).sum()).backward()
    gr_img, gr_feat = oracle.rasterize_backward(g1, ref['face_idx'], ref['weights'], fimg, torch.cat(feats, -1), 1e-8)
    gs_img = oracle.dibr_soft_mask_backward(g2, ref['soft_mask'], ref['face_idx'], ref['close_face_prob'],
                                            ref['close_face_idx'], ref['close_face_dist_type'], ref['scaled_vertices'],
                                            7000, 1000.)
    assert rel_close(a.grad, gr_img + gs_img)
    assert rel_close(torch.cat([x.grad for x in f], -1), gr_feat)


def test_error_strings():
    m = kal()._C.render.mesh
    img = torch.rand(2, 5, 3, 2, device='cuda')
    with pytest.raises(RuntimeError, match=r"Expected tensor of size \[2, 5, 4\], but got tensor of size \[2, 4, 4\] for "
                                           r"argument #2 'face_bboxes' \(while checking arguments for dibr_soft_mask_forward_cuda\)"):
        m.dibr_soft_mask_forward_cuda(img, torch.rand(2, 4, 4, device='cuda'),
                                      torch.zeros(2, 8, 8, dtype=torch.long, device='cuda'), 7000., 30, 1000.)
    with pytest.raises(RuntimeError, match=r"is on CPU"):
        m.dibr_soft_mask_forward_cuda(img.cpu(), torch.rand(2, 5, 4, device='cuda'),
                                      torch.zeros(2, 8, 8, dtype=torch.long, device='cuda'), 7000., 30, 1000.)
    with pytest.raises(RuntimeError, match=r"Expected contiguous tensor, but got non-contiguous tensor for argument #3 'face_vertices_z'"):
        m.packed_rasterize_forward_cuda(8, 8, torch.rand(3, 5, device='cuda').t(), torch.rand(5, 3, 2, device='cuda'),
                                        torch.rand(5, 4, device='cuda'), torch.rand(5, 3, 1, device='cuda'),
                                        torch.tensor([0, 5], device='cuda'), 1000., 1e-8)
    with pytest.raises(ValueError):
        kal().render.mesh.rasterize(8, 8, torch.rand(1, 5, 3, device='cuda'), img[:1], torch.rand(1, 5, 3, 1, device='cuda'),
                                    backend='nvdiffrast')


def test_full_size_properties_1024():
    """C4 shape (50k faces, 1024^2, 2 views) without an oracle run: (i) every covered pixel's barycentric
    weights are >= 0, sum to 1 and reproduce the pixel centre from the selected face's vertices;
    (ii) uncovered pixels hold -1 / zeros; (iii) soft_mask is 1 on covered pixels, in [0,1) elsewhere,
    and K-buffer rows are a prefix of hits followed by the -1/0/0 fill; (iv) rendering each view alone
    gives the same result as in the batch (views are independent)."""
    from kaolin_amd.utils import testing as T
    fz, fimg, feats, nz = T.sphere_scene(level=50, num_views=2, device='cuda')
    H = W = 1024
    m = kal().render.mesh
    out, soft, face_idx = m.dibr_rasterization(H, W, fz, fimg, feats, nz)
    cov = face_idx >= 0
    assert 0.1 < float(cov.float().mean()) < 0.9
    uv, ones = out
    assert torch.equal(ones[..., 0] > 0.5, cov)
    assert float((ones[..., 0][cov] - 1).abs().max()) < 1e-5
    assert float(uv[~cov].abs().max()) == 0.
    assert bool((soft[cov] == 1).all()) and float(soft[~cov].max()) <= 1. and float(soft.min()) >= 0.
    for b in range(2):
        o1, s1, i1 = m.dibr_rasterization(H, W, fz[b:b + 1], fimg[b:b + 1], [f[b:b + 1] for f in feats], nz[b:b + 1])
        assert torch.equal(i1[0], face_idx[b]) and torch.equal(s1[0], soft[b]) and torch.equal(o1[0][0], uv[b])
    # pixel centre reproduced by the weights: rebuild w from ones-feature trick is not available, so use
    # a position feature: features = vertex image coordinates -> interpolated value must equal the pixel centre
    pos, idx2 = m.rasterize(H, W, fz, fimg, fimg.contiguous(), nz >= 0)
    assert torch.equal(idx2, face_idx)
    xs = (2 * torch.arange(W, device='cuda', dtype=torch.float) + 1 - W) / W
    ys = (H - 2 * torch.arange(H, device='cuda', dtype=torch.float) - 1) / H
    assert float((pos[..., 0] - xs[None, None, :])[cov].abs().max()) < 1e-4
    assert float((pos[..., 1] - ys[None, :, None])[cov].abs().max()) < 1e-4
    s2, prob, idx, typ = _c_forward(fimg, face_idx, 7000, 0.02, 30, 1000.)
    hit = idx >= 0
    assert bool((hit[..., 1:] <= hit[..., :-1]).all())            # hits form a prefix
    assert bool(((typ > 0) == hit).all()) and float(prob[~hit].abs().max()) == 0.
    assert bool((idx[hit][1:] >= 0).all()) and not bool(hit[cov].any())
    hi = idx.clone()
    hi[~hit] = 10 ** 9
    assert bool(((hi[..., 1:] > hi[..., :-1]) | ~hit[..., 1:]).all())   # ascending face order


def test_backward_with_and_without_hit_count():
    """The reference-signature backward (reads the K-buffers' -1 terminators) and our hit-count shortcut agree."""
    fz, fimg, feats, nz = _scene(8, 2, torch.float)
    m = kal()._C.render.mesh
    _, face_idx = kal().render.mesh.rasterize(96, 64, fz.cuda(), fimg.cuda(), torch.cat(feats, -1).cuda(), (nz >= 0).cuda())
    scaled = fimg.cuda() * 1000.
    lo, hi = scaled.min(dim=-2)[0] - 20., scaled.max(dim=-2)[0] + 20.
    soft, prob, idx, typ, hits = m.dibr_soft_mask_forward_cuda(scaled, torch.cat([lo, hi], -1), face_idx, 7000., 30, 1000.,
                                                              _with_hit_count=True)
    assert torch.equal(hits.long(), (idx >= 0).sum(-1))
    g = torch.rand(soft.shape, device='cuda')
    a = m.dibr_soft_mask_backward_cuda(g, soft, face_idx, prob, idx, typ, scaled, 7000., 1000.)
    b = m.dibr_soft_mask_backward_cuda(g, soft, face_idx, prob, idx, typ, scaled, 7000., 1000., _hit_count=hits)
    assert same_sum_other_order(a, b, 1e-6)


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
def test_kbuffer_operators_match_lean_autograd_path(dtype):
    """The reference-contract operators (K-buffers) and the compact-list path used by autograd give the same
    soft_mask (bit-identical) and the same gradient (atomic order aside); the list holds exactly the K-buffer hits."""
    fz, fimg, feats, nz = _scene(10, 2, dtype)
    m = kal()._C.render.mesh
    H, W = 80, 112
    _, face_idx = kal().render.mesh.rasterize(H, W, fz.cuda(), fimg.cuda(), torch.cat(feats, -1).cuda(), (nz >= 0).cuda())
    scaled = fimg.cuda() * 1000.
    lo, hi = scaled.min(dim=-2)[0] - 20., scaled.max(dim=-2)[0] + 20.
    bbox = torch.cat([lo, hi], -1)
    soft, prob, idx, typ = m.dibr_soft_mask_forward_cuda(scaled, bbox, face_idx, 7000., 30, 1000.)
    soft2, hits = m.dibr_soft_mask_forward_lean(scaled, bbox, face_idx, 7000., 30, 1000.)
    assert torch.equal(soft, soft2)
    l_pix, l_face, l_prob, l_type = m.hit_list_entries(hits, fimg.shape[1], 2, H, W)
    assert int(hits[3][m.work_items(hits[4], 2, H, W)].sum()) == l_pix.numel()     # the items' segmented pair counts add up to it
    assert l_pix.numel() == int((idx >= 0).sum())
    # same multiset of (pixel, face, type, prob)
    pix = torch.nonzero(idx >= 0)
    flat_pix = (pix[:, 0] * H + pix[:, 1]) * W + pix[:, 2]
    a = torch.stack([flat_pix, idx[idx >= 0], typ[idx >= 0].long()], 1)
    b_ = torch.stack([l_pix.long(), l_face.long(), l_type.long()], 1)
    ka = (a[:, 0] * 100000 + a[:, 1]) * 8 + a[:, 2]
    kb = (b_[:, 0] * 100000 + b_[:, 1]) * 8 + b_[:, 2]
    oa, ob = torch.argsort(ka), torch.argsort(kb)
    assert torch.equal(ka[oa], kb[ob]) and torch.equal(prob[idx >= 0][oa], l_prob[ob])
    g = torch.rand(soft.shape, device='cuda', dtype=dtype)
    ga = m.dibr_soft_mask_backward_cuda(g, soft, face_idx, prob, idx, typ, scaled, 7000., 1000.)
    gb = m.dibr_soft_mask_backward_lean(g, soft2, hits, scaled, 7000., 30, 1000.)
    assert same_sum_other_order(ga, gb, 1e-6 if dtype == torch.float else 1e-12)


@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('with_valid', [False, True])
def test_fused_front_door_equals_reference_glue_plus_contract_operator(dtype, with_valid):
    """rasterize() (fused: packing / scaling / boxes inside the bin kernel) is bit-identical to the reference's
    torch glue around packed_rasterize_forward_cuda; same for the soft mask fused vs lean-with-torch-glue."""
    from kaolin_amd.render.mesh.rasterization import _packed_forward
    fz, fimg, feats, nz = _scene(12, 3, dtype)
    feat = torch.cat(feats, -1).cuda()
    valid = (nz >= 0).cuda() if with_valid else None
    H, W = 70, 130
    a, b_, c = _packed_forward(H, W, fz.cuda(), fimg.cuda(), feat, valid, 1000, 1e-8)
    x, y, w = kal()._C.render.mesh.rasterize_forward_fused(H, W, fz.cuda(), fimg.cuda(), feat, valid, 1000, 1e-8)
    assert torch.equal(b_, y) and torch.equal(a, x) and torch.equal(c, w)
    m = kal()._C.render.mesh
    scaled = fimg.cuda() * 1000.
    bbox = torch.cat([scaled.min(dim=-2)[0] - 0.02 * 1000., scaled.max(dim=-2)[0] + 0.02 * 1000.], -1)
    s1, h1 = m.dibr_soft_mask_forward_lean(scaled, bbox, y, 7000, 30, 1000.)
    s2, h2 = m.dibr_soft_mask_forward_fused(fimg.cuda(), y, 7000, 0.02, 30, 1000.)
    i1, i2 = m.work_items(h1[4], scaled.shape[0], H, W).sort()[0], m.work_items(h2[4], scaled.shape[0], H, W).sort()[0]   # queued in a run-dependent order
    assert torch.equal(s1, s2) and torch.equal(i1, i2) and torch.equal(h1[3][i1], h2[3][i2])
    g = torch.rand(s1.shape, device='cuda', dtype=dtype)
    g1 = m.dibr_soft_mask_backward_lean(g, s1, h1, scaled, 7000, 30, 1000.)
    g2 = m.dibr_soft_mask_backward_lean(g, s2, h2, fimg.cuda(), 7000, 30, 1000., img_scale=1000.)
    assert same_sum_other_order(g1, g2, 1e-6 if dtype == torch.float else 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float, torch.double])
def test_dibr_rasterization_reads_views_in_place(dtype):
    """`dibr_rasterization` fed with the `[..., 2]` views of `prepare_vertices`' outputs (what the tutorial does) reads
    them through strides -- no contiguous copy, no `>= 0` kernel: the result must equal the one for dense copies, and
    the one where the front-face mask is applied by hand through `rasterize` + `dibr_soft_mask`."""
    import kaolin_amd as kal
    from kaolin_amd.utils import testing as T
    v, f = T.geodesic_sphere(6)
    cams = T.fibonacci_cameras(3, 2.5, dtype)
    rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams.cuda(), torch.zeros(3, 3, dtype=dtype, device='cuda'),
                                                                      torch.tensor([[0., 1., 0.]], dtype=dtype, device='cuda').repeat(3, 1))
    proj = kal.render.camera.generate_perspective_projection(0.785, dtype=dtype).cuda()
    verts = v.to(dtype).cuda().unsqueeze(0).expand(3, -1, -1)
    fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(verts, f.cuda(), proj, camera_rot=rot, camera_trans=trans)
    feat = torch.rand(3, f.shape[0], 3, 2, dtype=dtype, device='cuda')
    z_view, n_view = fv_cam[..., 2], normals[..., 2]
    assert not z_view.is_contiguous() and not n_view.is_contiguous()
    a = kal.render.mesh.dibr_rasterization(48, 40, z_view, fv_img, feat, n_view)
    b = kal.render.mesh.dibr_rasterization(48, 40, z_view.contiguous(), fv_img, feat, n_view.contiguous())
    feats, idx = kal.render.mesh.rasterize(48, 40, z_view.contiguous(), fv_img, feat, valid_faces=n_view >= 0.)
    soft = kal.render.mesh.dibr_soft_mask(fv_img, idx)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(a[2], idx) and torch.equal(a[0], feats) and torch.equal(a[1], soft)
    assert int((idx >= 0).sum()) > 0


@pytest.mark.parametrize('D', [3, 5])
def test_static_features_skip_their_gradient(D):
    """autograd's needs_input_grad: with face_features that do not require grad (static texture coordinates) the
    backward kernels merge only the image-coordinate terms (g_feat = NULL at the C ABI).  The vertex gradient must be
    the one obtained when the features do require grad, through `rasterize` and through `dibr_rasterization`."""
    fz, fimg, feats, nz = _scene(10, 2, torch.float)
    torch.manual_seed(D)
    feat = torch.rand(feats[0].shape[:-1] + (D,))
    H, W = 96, 80
    up = torch.rand(2, H, W, D).cuda()
    up_soft = torch.rand(2, H, W).cuda()
    for name in ('rasterize', 'dibr_rasterization'):
        grads = []
        for needs in (True, False):
            a = fimg.cuda().requires_grad_()
            f = feat.cuda().requires_grad_(needs)
            if name == 'rasterize':
                out, _ = kal().render.mesh.rasterize(H, W, fz.cuda(), a, f, (nz >= 0).cuda())
                out.backward(up)
            else:
                out, soft, _ = kal().render.mesh.dibr_rasterization(H, W, fz.cuda(), a, f, nz.cuda())
                ((out * up).sum() + (soft * up_soft).sum()).backward()
            assert (f.grad is not None) == needs
            grads.append(a.grad.clone())
        assert same_sum_other_order(grads[1], grads[0], 1e-5), name


def test_gradient_buffer_cleared_by_the_forward_is_used_once():
    """The fused forward clears the buffer its backward accumulates into (one fill launch for both).  A second backward
    through a retained graph must not accumulate on top of the first result, and a forward whose image coordinates do not
    require grad must not leave a buffer behind."""
    fz, fimg, feats, nz = _scene(8, 2, torch.float)
    H, W = 64, 72
    feat = torch.cat(feats, -1).cuda()
    a = fimg.cuda().requires_grad_()
    out, soft, _ = kal().render.mesh.dibr_rasterization(H, W, fz.cuda(), a, feat, nz.cuda())
    loss = out.sum() + (soft * 3.).sum()
    loss.backward(retain_graph=True)
    first = a.grad.clone()
    a.grad = None
    loss.backward()
    assert same_sum_other_order(a.grad, first, 1e-5)
    f = feat.clone().requires_grad_()
    out2, soft2, _ = kal().render.mesh.dibr_rasterization(H, W, fz.cuda(), fimg.cuda(), f, nz.cuda())
    assert out2.grad_fn.zeroed_grad is None
    out2.sum().backward()
    assert f.grad is not None


def test_backward_without_the_forwards_signature_visits_every_tile():
    """the fused backward walks the forward's list of covered tiles.  A work buffer that does not carry the forward's
    layout signature (tl::WORK_MAGIC_WORD -- another operator's buffer, a tool's, a stale one) must not be read as tile
    indices: the walk falls back to every tile and the gradient is the same."""
    fz, fimg, feats, nz = _scene(10, 2, torch.float)
    m = kal()._C.render.mesh
    H, W = 96, 80
    feat = torch.cat(feats, -1).cuda()
    out = m.dibr_rasterization_forward_fused(H, W, fz.cuda(), fimg.cuda(), feat, nz.cuda(), 7000., 0.02, 30, 1000., 1e-8)
    interp, face_idx, wts, soft, hits, _ = out
    assert int(hits[4][1]) != 0                                   # the signature is there
    g1 = torch.rand(interp.shape, device='cuda')
    g2 = torch.rand(soft.shape, device='cuda')
    good, _ = m.dibr_rasterization_backward_fused(g1, g2, face_idx, wts, soft, hits, fimg.cuda(), feat, 7000., 30, 1000., 1e-8,
                                                  need_feature_grad=False)
    work = hits[4].clone()
    work[1] = 0                                                   # signature gone ...
    cov0 = m.WORK_COV_WORD
    work[cov0:cov0 + m.COV_SHARDS * m.COUNTER_STRIDE:m.COUNTER_STRIDE] = 1 << 30      # ... and the list's counts garbage
    bad_hits = hits[:4] + (work,)
    again, _ = m.dibr_rasterization_backward_fused(g1, g2, face_idx, wts, soft, bad_hits, fimg.cuda(), feat, 7000., 30, 1000., 1e-8,
                                                   need_feature_grad=False)
    assert same_sum_other_order(again, good, 1e-6)
    assert float(good.abs().max()) > 0


def test_wave_transpose64_without_lds():
    """The select kernel's 64 x 64 bit transpose (tile_bins.h: v_permlane32_swap / v_permlane16_swap, DPP moves, rotate + bit-field
    insert -- no LDS crossbar since round 4) against numpy, on random matrices, the identity, single bits and full rows; the
    __shfl_xor form it replaced gives the same."""
    import ctypes
    from kaolin_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    mats = [rng.integers(0, 2, size=(64, 64), dtype=np.uint8) for _ in range(40)]
    mats.append(np.eye(64, dtype=np.uint8))
    mats.append(np.zeros((64, 64), np.uint8))
    mats.append(np.ones((64, 64), np.uint8))
    for i, j in ((0, 63), (63, 0), (31, 32), (32, 31), (17, 5), (47, 16)):
        m = np.zeros((64, 64), np.uint8)
        m[i, j] = 1
        mats.append(m)
    for i in (0, 15, 16, 33, 63):
        m = np.zeros((64, 64), np.uint8)
        m[i, :] = 1
        mats.append(m)
    weights = (np.uint64(1) << np.arange(64, dtype=np.uint64))

    def pack(m):      # row i -> uint64 with bit j = m[i, j]
        return (m.astype(np.uint64) * weights[None, :]).sum(axis=1, dtype=np.uint64)
    rows = np.stack([pack(m) for m in mats])                         # (n, 64)
    want = np.stack([pack(m.T) for m in mats])
    x = torch.from_numpy(rows.view(np.int64)).cuda()
    for reference in (0, 1):
        out = torch.zeros_like(x)
        st = lib.kamd_debug_transpose64(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), len(mats), ctypes.c_void_p(x.data_ptr()),
                                        ctypes.c_void_p(out.data_ptr()), reference)
        assert st == 0
        got = out.cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), f'reference={reference}: {int((got != want).sum())} words differ'


def test_nan_vertex_matches_reference_glue():
    """torch.min / torch.max propagate NaN, so a face with a NaN vertex has a NaN box that rejects no pixel; the
    fused binning must give the reference glue's result (`_packed_forward`), not confine the face to its finite corners."""
    from kaolin_amd.render.mesh.rasterization import _packed_forward
    fz, fimg, feats, nz = _scene(6, 2, torch.float)
    fimg = fimg.clone()
    fimg[0, 5, 1, 0] = float('nan')
    fimg[1, 17, 2, 1] = float('nan')
    feat = torch.cat(feats, -1).cuda()
    H, W = 48, 40
    a, b_, c = _packed_forward(H, W, fz.cuda(), fimg.cuda(), feat, None, 1000, 1e-8)
    x, y, w = kal()._C.render.mesh.rasterize_forward_fused(H, W, fz.cuda(), fimg.cuda(), feat, None, 1000, 1e-8)
    assert torch.equal(b_, y)
    assert torch.equal(torch.nan_to_num(a, nan=-7.), torch.nan_to_num(x, nan=-7.))
    assert torch.equal(torch.nan_to_num(c, nan=-7.), torch.nan_to_num(w, nan=-7.))


def test_meshes_without_faces():
    """B*F = 0: every pixel is background (index -1, zero features, zero soft mask), nothing is launched on the faces; the
    calls after it are unaffected."""
    H, W = 24, 40
    z = torch.zeros((2, 0, 3)).cuda()
    out, soft, idx = kal().render.mesh.dibr_rasterization(H, W, z, torch.zeros((2, 0, 3, 2)).cuda(), torch.zeros((2, 0, 3, 3)).cuda(),
                                                          torch.zeros((2, 0)).cuda())
    assert bool((idx == -1).all()) and float(soft.abs().max()) == 0. and float(out.abs().max()) == 0.
    assert out.shape == (2, H, W, 3) and soft.shape == (2, H, W) and idx.shape == (2, H, W)
    fz, fimg, feats, nz = _scene(6, 2, torch.float)
    ref = oracle.dibr_rasterization(H, W, fz, fimg, torch.cat(feats, -1), nz, omp=True)
    out, soft, idx = kal().render.mesh.dibr_rasterization(H, W, fz.cuda(), fimg.cuda(), torch.cat(feats, -1).cuda(), nz.cuda())
    assert torch.equal(idx.cpu(), ref['face_idx']) and rel_close(soft, ref['soft_mask'])


def test_zero_sized_image_gives_zero_gradients():
    """H = 0 (or W = 0) with faces present: no kernel runs, and the gradient buffer the forward prepared for the
    backward (allocated uninitialised) must still come back as zeros."""
    import kaolin_amd  # noqa: F401
    fz, fimg, feats, nz = _scene(4, 2, torch.float)
    feat = torch.cat(feats, -1).cuda()
    for H, W in ((0, 16), (16, 0)):
        a = fimg.cuda().requires_grad_()
        torch.empty(1 << 20, device='cuda').fill_(float('nan'))            # poison what the allocator hands out next
        out, soft, idx = kal().render.mesh.dibr_rasterization(H, W, fz.cuda(), a, feat, nz.cuda())
        assert out.shape == (2, H, W, 3) and soft.shape == (2, H, W) and idx.shape == (2, H, W)
        (out.sum() + soft.sum()).backward()
        assert a.grad is not None and float(a.grad.abs().max()) == 0.


def _soft_backward_on_the_gpu_forwards_outputs(g2, fimg, face_idx, ref, boxlen=0.02, knum=30):
    """The soft mask's backward is a function of the forward's OUTPUTS and ill-conditioned in them where the mask is close to 1
    (dL/dz ~ (1 - mask): a mask that is 1 - 6e-8 with one library's exp() and 1 with the other's gives a term or none).  So the
    oracle's backward is evaluated on the GPU forward's own mask and K-buffers -- the contract operator's, whose mask is
    bit-identical to the fused operator's and whose indices / types equal the oracle's.  -> (gradient, sum of |terms|)"""
    s2, kprob, kidx, ktyp = _c_forward(fimg.cuda(), face_idx, 7000, boxlen, knum, 1000.)
    assert torch.equal(kidx.cpu(), ref['close_face_idx']) and torch.equal(ktyp.cpu(), ref['close_face_dist_type'])
    return oracle.dibr_soft_mask_backward(g2, s2.cpu(), ref['face_idx'], kprob.cpu(), kidx.cpu(), ktyp.cpu(), ref['scaled_vertices'],
                                          7000, 1000., return_abs=True)


@pytest.mark.parametrize('H,W,views', [(512, 512, 3), (48, 2112, 2), (300, 200, 2)])
def test_floor_under_the_object_vs_oracle(H, W, views):
    """An object standing on a floor of two image-sized triangles: big faces go to their view's big list and mark the tiles of
    their rectangles in per-row bit words (tile_lists.h, big_rows) -- tiles outside keep their own lists and the background
    fast path.  2112 pixels across = 132 tile columns: three words per row.  (A face whose box is 'everywhere' -- a NaN vertex --
    marks every tile: test_nan_vertex_matches_reference_glue.)"""
    from kaolin_amd.utils import testing as T
    v, f = T.geodesic_sphere(10)
    v = v * 0.45
    floor_v = torch.tensor([[-1.6, -0.5, -1.6], [1.6, -0.5, -1.6], [1.6, -0.5, 1.6], [-1.6, -0.5, 1.6]], dtype=v.dtype)
    n = v.shape[0]
    faces = torch.cat([f, torch.tensor([[n, n + 2, n + 1], [n, n + 3, n + 2]])])
    fz, fimg, feats, nz = T.mesh_scene(torch.cat([v, floor_v]), faces, views, 'cpu', torch.float, 1, 2.5)
    ref = oracle.dibr_rasterization(H, W, fz, fimg, torch.cat(feats, -1), nz, omp=True)
    a = fimg.cuda().requires_grad_()
    out, soft, face_idx = kal().render.mesh.dibr_rasterization(H, W, fz.cuda(), a, [x.cuda() for x in feats], nz.cuda())
    assert torch.equal(face_idx.cpu(), ref['face_idx'])
    assert torch.equal(torch.cat(out, -1).cpu(), ref['features'])
    assert rel_close(soft.detach(), ref['soft_mask'])
    floor_ids = torch.tensor([f.shape[0], f.shape[0] + 1])
    assert bool(torch.isin(ref['face_idx'], floor_ids).any()), 'the floor is visible in the reference image'
    # the standalone rasterizer takes the same lists
    out2, idx2 = kal().render.mesh.rasterize(H, W, fz.cuda(), fimg.cuda(), [x.cuda() for x in feats], (nz >= 0).cuda())
    assert torch.equal(idx2, face_idx) and torch.equal(torch.cat(out2, -1), torch.cat(out, -1))
    torch.manual_seed(5)
    g1 = torch.rand(ref['features'].shape)
    g2 = torch.rand(ref['soft_mask'].shape)
    ((torch.cat(out, -1) * g1.cuda()).sum() + (soft * g2.cuda()).sum()).backward()
    gr_img, _, sr = oracle.rasterize_backward(g1, ref['face_idx'], ref['weights'], fimg, torch.cat(feats, -1), 1e-8, return_abs=True)
    gs_img, ss = _soft_backward_on_the_gpu_forwards_outputs(g2, fimg, face_idx, ref)
    assert torch.equal(soft.detach(), _c_forward(fimg.cuda(), face_idx, 7000, 0.02, 30, 1000.)[0])
    # (the floor's vertices collect tens of thousands of float-atomic terms of both signs: the accumulation's own rounding scale
    # joins the element-wise 1e-5, as for the knot scene's bowl in test_full_size_parity.py)
    assert rel_close(a.grad, gr_img + gs_img, term_abs_sum=sr + ss)


@pytest.mark.parametrize('n,size', [(400, 0.7), (3500, 0.6), (150, 1.6)])
def test_stacked_medium_faces_vs_oracle(n, size):
    """Hundreds of faces of 5-16 tiles (up to 64 in the soft pass) piled on the same tiles: each appends single-face entries (tile_lists.h,
    append_own), far past a tile's inline slots -- pool chunks, the slots parked in LDS, and with 3 500 of them past the 512 entries a
    rasterizer tile's table holds (the tile is flagged and scans its mesh).  face_idx / features bit-exact, soft mask and gradient
    at 1e-5 against the oracle."""
    H = W = 256
    g = torch.Generator().manual_seed(n)
    centre = (torch.rand((n, 1, 2), generator=g) - 0.5) * 0.5
    img = (centre + (torch.rand((n, 3, 2), generator=g) - 0.5) * size).unsqueeze(0).float()
    z = -(torch.rand((1, n, 3), generator=g) + 1).float()
    feat = torch.rand((1, n, 3, 2), generator=g).float()
    nz = torch.ones((1, n))
    ref = oracle.dibr_rasterization(H, W, z, img, feat, nz, omp=True)
    a = img.cuda().requires_grad_()
    out, soft, face_idx = kal().render.mesh.dibr_rasterization(H, W, z.cuda(), a, feat.cuda(), nz.cuda())
    assert torch.equal(face_idx.cpu(), ref['face_idx'])
    assert torch.equal(out.cpu(), ref['features'])
    assert rel_close(soft.detach(), ref['soft_mask'])
    torch.manual_seed(6)
    g1, g2 = torch.rand(ref['features'].shape), torch.rand(ref['soft_mask'].shape)
    ((out * g1.cuda()).sum() + (soft * g2.cuda()).sum()).backward()
    gr_img, _, sr = oracle.rasterize_backward(g1, ref['face_idx'], ref['weights'], img, feat, 1e-8, return_abs=True)
    gs_img, ss = _soft_backward_on_the_gpu_forwards_outputs(g2, img, face_idx, ref)
    assert rel_close(a.grad, gr_img + gs_img, term_abs_sum=sr + ss)


@pytest.mark.timeout(120)
@pytest.mark.parametrize('dtype', [torch.float, torch.double])
def test_inverted_boxes_and_meshes_without_faces_in_the_packed_operator(dtype):
    """The contract operator takes the caller's boxes as they come: a box with min > max holds no pixel (the reference tests
    x >= xmin and x < xmax) -- it must give the oracle's image, and it must not reach the binning kernel's tile arithmetic (a
    rectangle of negative width made its loop spin for ever).  And a packed batch in which NO mesh has a face (null workspace)
    is an image of background.  Both found by tools/round4/fuzz_rasterize_ops.py."""
    M = kal()._C.render.mesh
    g = torch.Generator().manual_seed(3)
    H, W, mult = 41, 1340, 37.5
    counts = [40, 0, 126]
    first = torch.tensor([0, 40, 40, 166])
    F = 166
    size = 10.0 ** (torch.rand(F, generator=g) * 2.5 - 2.2)
    img = (((torch.rand(F, 1, 2, generator=g) - 0.5) * 2.2 + (torch.rand(F, 3, 2, generator=g) - 0.5) * size.view(-1, 1, 1)) * mult).to(dtype)
    z = -(torch.rand(F, 3, generator=g) * 2 + 0.2).to(dtype)
    feat = torch.rand(F, 3, 2, generator=g).to(dtype)
    lo, hi = img.min(dim=1)[0], img.max(dim=1)[0]
    bbox = torch.cat([lo + 0.02 * mult, hi - 0.02 * mult], dim=-1).contiguous()      # too small; inverted for the small faces
    assert bool((bbox[:, 0] > bbox[:, 2]).any())
    want = oracle.packed_rasterize_forward(H, W, z, img, bbox, feat, first, mult, 1e-8, omp=True)
    got = M.packed_rasterize_forward_cuda(H, W, z.cuda(), img.cuda(), bbox.cuda(), feat.cuda(), first.cuda(), mult, 1e-8)
    assert torch.equal(got[1].cpu(), want[1]) and torch.equal(got[0].cpu(), want[0]) and torch.equal(got[2].cpu(), want[2])
    # no face in any mesh
    e = lambda *s: torch.zeros(*s, dtype=dtype, device='cuda')   # noqa: E731
    out = M.packed_rasterize_forward_cuda(4, 304, e(0, 3), e(0, 3, 2), e(0, 4), e(0, 3, 3), torch.zeros(3, dtype=torch.long, device='cuda'), 1000., 1e-8)
    assert out[1].shape == (2, 4, 304) and bool((out[1] == -1).all()) and float(out[0].abs().sum()) == 0 and float(out[2].abs().sum()) == 0
    # a negative boxlen inverts the soft mask's enlarged boxes
    fz, fimg, feats, nz = _scene(6, 1, dtype)
    _, face_idx = kal().render.mesh.rasterize(64, 64, fz.cuda(), fimg.cuda(), [x.cuda() for x in feats], (nz >= 0).cuda())
    soft = kal().render.mesh.dibr_soft_mask(fimg.cuda(), face_idx, 7000., -0.05, 30, 1000.)
    ref = oracle.dibr_soft_mask(fimg, face_idx.cpu(), 7000., -0.05, 30, 1000., omp=True)[0]
    assert rel_close(soft, ref)


# ------------------------------------------------------------------ box edges within a hair of pixel centres
@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('H,W,big_x', [(16, 4096, False), (64, 1024, False), (24, 2048, True)])
def test_box_edges_within_two_thousandths_of_a_pixel_centre(dtype, H, W, big_x):
    """The binning launch turns a face's box into pixel ranges with the CLOSED FORM of the kernels' test (tl::pixel_range: columns
    ceil(u(xmin)) .. ceil(u(xmax)) - 1, the ends moved outwards by 1e-3 + 1e-6 W pixels for the float roundings of u and of the
    kernels' pixel centres).  A face it missed would silently lose a covered or a soft pixel.  Here every face has box edges within
    +-2e-3 pixel of pixel centres -- on them, a float ulp beside them, 5e-4 and 2e-3 pixel away on either side -- at widths up to
    4096 and, `big_x`, with the mesh pushed to the image's right edge where |x| is largest: face_idx, the K-buffer indices and the
    soft mask must equal the oracle's, which tests every pixel against every face."""
    g = torch.Generator().manual_seed(1234 + H + W)
    F = 600
    cols = torch.randint(2, W - 8, (F,), generator=g).double() if not big_x else torch.randint(W - 200, W - 8, (F,), generator=g).double()
    rows = torch.randint(1, H - 4, (F,), generator=g).double()
    wpx = torch.randint(1, 6, (F,), generator=g).double()
    hpx = torch.randint(1, 4, (F,), generator=g).double()
    offs = torch.tensor([0.0, 1e-7, -1e-7, 5e-4, -5e-4, 2e-3, -2e-3, 1e-3, -1e-3])
    pick = lambda: offs[torch.randint(0, len(offs), (F,), generator=g)].double()
    # pixel centre of column c in NDC: (2 c + 1 - W) / W; of row r: (H - 2 r - 1) / H
    x0 = (2 * (cols + pick()) + 1 - W) / W
    x1 = (2 * (cols + wpx + pick()) + 1 - W) / W
    y_top = (H - 2 * (rows + pick()) - 1) / H
    y_bot = (H - 2 * (rows + hpx + pick()) - 1) / H
    xm = (2 * (cols + wpx * torch.rand(F, generator=g).double() + pick()) + 1 - W) / W
    y_third = torch.where(torch.rand(F, generator=g) < 0.5, y_top, y_bot)   # (the box stays [min, max] of the hair-offset coordinates)
    fimg = torch.stack([torch.stack([x0, y_top], -1), torch.stack([x1, y_bot], -1), torch.stack([xm, y_third], -1)], 1)
    fimg = fimg.to(dtype)[None].contiguous()
    fz = (-1.0 - torch.rand((1, F, 3), generator=g).double()).to(dtype)
    feat = torch.rand((1, F, 3, 2), generator=g).double().to(dtype)
    nz = torch.ones((1, F), dtype=dtype)
    ref = oracle.dibr_rasterization(H, W, fz, fimg, feat, nz, boxlen=0.002, knum=8, omp=True)
    out, soft, face_idx = kal().render.mesh.dibr_rasterization(H, W, fz.cuda(), fimg.cuda(), feat.cuda(), nz.cuda(), boxlen=0.002, knum=8)
    assert torch.equal(face_idx.cpu(), ref['face_idx'])
    assert int((ref['face_idx'] >= 0).sum()) > 500
    assert torch.equal(out.cpu(), ref['features'])
    assert rel_close(soft, ref['soft_mask'])
    s2, prob, idx, typ = _c_forward(fimg.cuda(), face_idx, 7000, 0.002, 8, 1000.)
    assert torch.equal(idx.cpu(), ref['close_face_idx']) and torch.equal(typ.cpu(), ref['close_face_dist_type'])


@pytest.mark.gpu
@pytest.mark.parametrize('n_big', [40, 2500])
def test_fused_backward_with_image_sized_faces_equals_the_composition(n_big):
    """Faces whose enlarged box spans more than 8 x 8 soft tiles collect their soft-mask gradient in per-XCD partial sums that the
    rasterizer's backward launch folds in (csrc/tile_lists.h, WORK_BIGHASH_WORD): with 40 such faces the table is in use, with
    2 500 it is over its limit and left alone.  Either way the fused operator's vertex gradient equals the composition's
    (rasterize + dibr_soft_mask: the stand-alone soft-mask backward never uses the table), and a second backward through the
    retained graph gives the same again (the partial sums are left cleared)."""
    import kaolin_amd as kal
    g = torch.Generator().manual_seed(n_big)
    H = W = 288
    n_small = 300
    F = n_big + n_small
    size = torch.cat([torch.full((n_big,), 2.2), torch.full((n_small,), 0.06)])[torch.randperm(F, generator=g)]
    centre = (torch.rand(1, F, 1, 2, generator=g) - 0.5) * 1.6
    img = centre + (torch.rand(1, F, 3, 2, generator=g) - 0.5) * size.view(1, F, 1, 1)
    z = -(torch.rand(1, F, 3, generator=g) * 2 + 0.5)
    feat = torch.rand(1, F, 3, 2, generator=g)
    nz = torch.rand(1, F, generator=g) - 0.3          # 30 % back faces
    g1, g2 = torch.rand(1, H, W, 2, generator=g).cuda(), torch.rand(1, H, W, generator=g).cuda()
    a = img.cuda().requires_grad_()
    out, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, z.cuda(), a, feat.cuda(), nz.cuda(), knum=5)
    loss = (out * g1).sum() + (soft * g2).sum()
    grad_fused, = torch.autograd.grad(loss, a, retain_graph=True)
    grad_again, = torch.autograd.grad(loss, a)
    b = img.cuda().requires_grad_()
    out2, idx2 = kal.render.mesh.rasterize(H, W, z.cuda(), b, feat.cuda(), valid_faces=nz.cuda() >= 0)
    soft2 = kal.render.mesh.dibr_soft_mask(b, idx2, knum=5)
    assert torch.equal(idx2, face_idx) and torch.equal(out2, out) and torch.equal(soft2, soft)
    grad_comp, = torch.autograd.grad((out2 * g1).sum() + (soft2 * g2).sum(), b)
    scale = float(grad_comp.abs().max())
    assert scale > 0
    for name, got in (('fused', grad_fused), ('second backward', grad_again)):
        d = (got - grad_comp).abs()
        bad = d > 1e-4 * grad_comp.abs() + 2e-5 * scale     # (sums of thousands of float atomics in two different orders)
        assert not bool(bad.any()), f'{name}: {int(bad.sum())} of {bad.numel()} elements off, worst {float(d.max()):.3g} at scale {scale:.3g}'
