"""Randomised sweep of the STANDALONE soft-mask operators (an arbitrary selected_face_idx supplied by the caller, not the rasterizer's):
the contract operator with its K-buffers and the autograd function, against the oracle -- extreme aspect ratios (1 x W, H x 1, 3000 x 7),
random coverage patterns (all covered, none, noise, half planes), sigmainv / boxlen / knum / multiplier variants, fp32 / fp64; and the
standalone rasterizer with a bool valid_faces on the same scenes.
usage (GPU box): python tools/round4/fuzz_soft_mask.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
import kaolin_amd as kal
from kaolin_amd.utils.testing import elementwise_mismatch

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, t0 = 0, time.time()
def fail(case, what, msg):
    global bad
    bad += 1
    print(f'case {case} {what} FAILED: {msg}', flush=True)
for case in range(seed0, seed0 + n_cases):
    g = torch.Generator().manual_seed(case)
    r = lambda *s: torch.rand(*s, generator=g)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    dtype = torch.float64 if case % 4 == 3 else torch.float32
    H, W = [(ri(1, 3), ri(1, 600)), (ri(1, 600), ri(1, 3)), (ri(1000, 3000), ri(1, 9)), (ri(10, 200), ri(10, 200)), (ri(30, 90), ri(300, 1300))][case % 5]
    B, F = ri(1, 3), ri(1, 500)
    size = 10.0 ** (r(F) * 2.5 - 2.2)
    img = ((r(B, F, 1, 2) - 0.5) * 2.2 + (r(B, F, 3, 2) - 0.5) * size.view(1, F, 1, 1)).to(dtype)
    pattern = ri(0, 4)
    sel = torch.randint(0, F, (B, H, W), generator=g)
    if pattern == 0:
        sel[:] = -1
    elif pattern == 1:
        sel[r(B, H, W) < 0.5] = -1
    elif pattern == 2:
        sel[:, :, W // 2:] = -1
    elif pattern == 3:
        sel[r(B, H, W) < 0.97] = -1
    sigmainv, boxlen, knum, mult = [7000., 70., 30000.][ri(0, 2)], [0.02, 0.2, 0.005][ri(0, 2)], [30, 1, 7, 150][ri(0, 3)], [1000., 1., 100.][ri(0, 2)]
    s_ref, p_ref, i_ref, t_ref, scaled = oracle.dibr_soft_mask(img, sel, sigmainv, boxlen, knum, mult, omp=True)
    lo, hi = scaled.min(dim=-2)[0], scaled.max(dim=-2)[0]
    bbox = torch.cat([lo - boxlen * mult, hi + boxlen * mult], dim=-1)
    desc = f'({H}x{W} B={B} F={F} {dtype} pattern {pattern} sigmainv={sigmainv} boxlen={boxlen} knum={knum} mult={mult})'
    s2, kp, ki, kt = kal._C.render.mesh.dibr_soft_mask_forward_cuda(scaled.cuda(), bbox.cuda(), sel.cuda(), sigmainv, knum, mult)
    eps = torch.finfo(dtype).eps
    tol = 1e-5 if dtype == torch.float32 else 1e-10
    if not (torch.equal(ki.cpu(), i_ref) and torch.equal(kt.cpu(), t_ref)):
        fail(case, 'contract operator ' + desc, f'K-buffer index differs at {int((ki.cpu() != i_ref).sum())}, type at {int((kt.cpu() != t_ref).sum())}')
        continue
    d = (kp.cpu().double() - p_ref.double()).abs()
    if not bool((d <= tol * p_ref.double().abs() + 1e-37).all()):
        fail(case, 'contract operator ' + desc, f'probabilities differ by up to {float(d.max()):.3g}')
    d = (s2.cpu().double() - s_ref.double()).abs()
    if not bool((d <= tol * s_ref.double().abs() + 4 * eps).all()):
        fail(case, 'contract operator ' + desc, f'soft mask differs by up to {float(d.max()):.3g}')
    a = img.cuda().requires_grad_()
    soft = kal.render.mesh.dibr_soft_mask(a, sel.cuda(), sigmainv, boxlen, knum, mult)
    if not torch.equal(soft.detach(), s2):
        fail(case, 'autograd function ' + desc, 'soft mask differs from the contract operator')
    g2 = r(B, H, W).to(dtype)
    (soft * g2.cuda()).sum().backward()
    gs, ss = oracle.dibr_soft_mask_backward(g2, s2.cpu(), sel, kp.cpu(), ki.cpu(), kt.cpu(), scaled, sigmainv, mult, return_abs=True)
    # (the operator's gradient IS the one w.r.t. the unscaled vertices: its terms carry the 1 / multiplier of dibr_soft_mask_cuda.cu:299-302)
    m = elementwise_mismatch(a.grad, gs, tol, term_abs_sum=ss)
    if m:
        fail(case, 'autograd backward ' + desc, m)
    gc = kal._C.render.mesh.dibr_soft_mask_backward_cuda(g2.cuda(), s2, sel.cuda(), kp, ki, kt, scaled.cuda(), sigmainv, mult)
    m = elementwise_mismatch(gc, gs, tol, term_abs_sum=ss)
    if m:
        fail(case, 'contract backward ' + desc, m)
print(f'{n_cases} cases from seed {seed0}: {bad} failed checks, {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if bad else 0)
