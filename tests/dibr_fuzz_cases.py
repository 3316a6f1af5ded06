"""Randomised parity cases of the fused DIB-R operator against the oracle (test infrastructure; used by tests/test_dibr_fuzz.py and
tools/round4/fuzz_dibr.py): image sizes that are not multiples of the tile sizes, batches, mixtures of tiny / medium / image-sized
faces (the three binning paths of tile_lists.h), faces partly or wholly outside the image, back faces, few and many faces per tile,
small knum, fp32 / fp64.  face_idx / features / K-buffer indices and types bit-exact, soft mask at 1e-5 (+ 4 eps absolute), the vertex
gradient at the element-wise 1e-5 against the oracle's backward evaluated on the GPU forward's own outputs."""
import torch

import oracle
from kaolin_amd.utils.testing import elementwise_mismatch


def check_case(case):
    """-> (description, list of mismatch messages) of case number `case` (deterministic in it)."""
    import kaolin_amd as kal
    msgs = []
    g = torch.Generator().manual_seed(case)
    r = lambda *s: torch.rand(*s, generator=g)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    dtype = torch.float64 if case % 5 == 4 else torch.float32
    H, W = ri(5, 300), ri(5, 300)
    if case % 7 == 0:
        W = ri(1100, 1500); H = ri(20, 60)          # more than 64 tile columns: two words per big-face tile row
    B = ri(1, 3)
    n_tiny, n_med, n_big = ri(0, 600), ri(0, 120), ri(0, 12)
    F = max(n_tiny + n_med + n_big, 1)
    size = torch.cat([torch.full((n_tiny,), 0.04), torch.full((n_med,), 0.35), torch.full((n_big,), 2.5), torch.full((F - n_tiny - n_med - n_big,), 0.1)])
    size = size[torch.randperm(F, generator=g)] * (0.3 + 1.4 * r(F))
    spread = [0.5, 1.0, 2.4][ri(0, 2)]               # 2.4: a good part of the faces lies outside the image
    centre = (r(B, F, 1, 2) - 0.5) * 2 * spread
    img = (centre + (r(B, F, 3, 2) - 0.5) * size.view(1, F, 1, 1)).to(dtype)
    z = -(r(B, F, 3) * 2 + 0.5).to(dtype)
    D = ri(1, 4)
    feat = r(B, F, 3, D).to(dtype)
    nz = (r(B, F) - (0.3 if case % 3 == 0 else -1.0)).to(dtype)     # a third of the cases: 30 % back faces
    knum = [30, 30, 5, 1][ri(0, 3)]
    boxlen = [0.02, 0.02, 0.1][ri(0, 2)]
    ref = oracle.dibr_rasterization(H, W, z, img, feat, nz, boxlen=boxlen, knum=knum, omp=True)
    a = img.cuda().requires_grad_()
    out, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, z.cuda(), a, feat.cuda(), nz.cuda(), boxlen=boxlen, knum=knum)
    if not torch.equal(face_idx.cpu(), ref['face_idx']):
        msgs.append(f'face_idx differs at {int((face_idx.cpu() != ref["face_idx"]).sum())} pixels')
    elif not torch.equal(out.cpu(), ref['features']):
        msgs.append('features differ')
    # soft = 1 - prod(1 - p): a value near 0 is a difference of two numbers near 1 and carries THEIR rounding (exp() differs by an
    # ulp between libraries): 4 eps absolute on top of the element-wise 1e-5
    eps = torch.finfo(dtype).eps
    d = (soft.detach().cpu().double() - ref['soft_mask'].double()).abs()
    okm = d <= 1e-5 * ref['soft_mask'].double().abs() + 4 * eps
    if not bool(okm.all()):
        msgs.append(f'soft mask: {int((~okm).sum())} elements off, worst {float(d.max()):.3g}')
    g1, g2 = r(*ref['features'].shape).to(dtype), r(*ref['soft_mask'].shape).to(dtype)
    ((out * g1.cuda()).sum() + (soft * g2.cuda()).sum()).backward()
    gr, _, sr = oracle.rasterize_backward(g1, ref['face_idx'], ref['weights'], img, feat, 1e-8, return_abs=True)
    # the soft mask's backward is a function of the forward's OUTPUTS, and ill-conditioned in them where the mask is close to 1
    # (dL/dz ~ (1 - mask): a mask that is 1 - 6e-8 in one library and 1 in the other gives a term or none): the oracle's backward
    # gets the GPU forward's own mask and K-buffers (the contract operator's, whose mask is bit-identical to the fused operator's)
    scaled = a.detach() * 1000.
    lo, hi = scaled.min(dim=-2)[0], scaled.max(dim=-2)[0]
    bbox = torch.cat([lo - boxlen * 1000., hi + boxlen * 1000.], dim=-1)
    s2, kprob, kidx, ktyp = kal._C.render.mesh.dibr_soft_mask_forward_cuda(scaled, bbox, face_idx, 7000., knum, 1000.)
    if not torch.equal(s2, soft.detach()):
        msgs.append('the contract operator and the fused operator disagree on the soft mask')
    if not (torch.equal(kidx.cpu(), ref['close_face_idx']) and torch.equal(ktyp.cpu(), ref['close_face_dist_type'])):
        msgs.append('K-buffer indices / types differ from the oracle')
    gs, ss = oracle.dibr_soft_mask_backward(g2, s2.cpu(), ref['face_idx'], kprob.cpu(), kidx.cpu(), ktyp.cpu(), ref['scaled_vertices'],
                                            7000, 1000., return_abs=True)
    m = elementwise_mismatch(a.grad, gr + gs, 1e-5 if dtype == torch.float32 else 1e-9, term_abs_sum=sr + ss)
    if m:
        msgs.append('vertex gradient: ' + m)
    desc = f'{H}x{W} B={B} F={F} tiny/med/big {n_tiny}/{n_med}/{n_big} {dtype} knum={knum} boxlen={boxlen} spread={spread}'
    return desc, msgs
