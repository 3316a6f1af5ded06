"""Randomised sweep of the rasterizer's CONTRACT operators against the oracle: packed_rasterize_forward_cuda with ragged meshes (face
counts per mesh from 0 upwards through first_idx_face_per_mesh), caller-supplied boxes (exact, enlarged, deliberately too small -- the
operator must use THEM, as the reference does), multiplier / eps variants, fp32 / fp64, odd image sizes; and rasterize_backward_cuda on
an arbitrary selected_face_idx / weights pair (feature gradient included).
usage (GPU box): python tools/round4/fuzz_rasterize_ops.py [n_cases] [first_seed]"""
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
M = kal._C.render.mesh
for case in range(seed0, seed0 + n_cases):
    g = torch.Generator().manual_seed(case)
    r = lambda *s: torch.rand(*s, generator=g)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    dtype = torch.float64 if case % 4 == 3 else torch.float32
    H, W = [(ri(1, 4), ri(1, 500)), (ri(1, 500), ri(1, 4)), (ri(8, 260), ri(8, 260)), (ri(20, 70), ri(900, 1400))][case % 4]
    B = ri(1, 4)
    counts = [ri(0, 400) if ri(0, 4) else 0 for _ in range(B)]
    first = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.long)
    Ft = int(first[-1])
    mult, eps_ = [1000., 1., 37.5][ri(0, 2)], [1e-8, 1e-3, 0.3][ri(0, 2)]
    size = 10.0 ** (r(max(Ft, 1)) * 2.5 - 2.2)
    img = (((r(max(Ft, 1), 1, 2) - 0.5) * 2.2 + (r(max(Ft, 1), 3, 2) - 0.5) * size.view(-1, 1, 1)) * mult).to(dtype)[:Ft]
    z = -(r(max(Ft, 1), 3) * 2 + 0.2).to(dtype)[:Ft]
    D = ri(1, 5)
    feat = r(max(Ft, 1), 3, D).to(dtype)[:Ft]
    lo, hi = (img.min(dim=1)[0], img.max(dim=1)[0]) if Ft else (torch.zeros(0, 2, dtype=dtype), torch.zeros(0, 2, dtype=dtype))
    kind = ri(0, 2)
    pad = [0.0, 0.05 * mult, -0.02 * mult][kind]          # exact / enlarged / too small boxes
    bbox = torch.cat([lo - pad, hi + pad], dim=-1).contiguous()
    desc = f'({H}x{W} B={B} faces {counts} {dtype} mult={mult} eps={eps_} boxes {["exact", "enlarged", "too small"][kind]} D={D})'
    if os.environ.get('FUZZ_PROGRESS'):
        print('case', case, desc, flush=True)
    want = oracle.packed_rasterize_forward(H, W, z, img, bbox, feat, first, mult, eps_, omp=True)
    got = M.packed_rasterize_forward_cuda(H, W, z.cuda(), img.cuda(), bbox.cuda(), feat.cuda(), first.cuda(), mult, eps_)
    torch.cuda.synchronize()
    if os.environ.get('FUZZ_PROGRESS'):
        print('   forward done', flush=True)
    if not torch.equal(got[1].cpu(), want[1]):
        fail(case, 'packed_rasterize_forward_cuda ' + desc, f'selected_face_idx differs at {int((got[1].cpu() != want[1]).sum())} pixels')
        continue
    if not (torch.equal(got[0].cpu(), want[0]) and torch.equal(got[2].cpu(), want[2])):
        fail(case, 'packed_rasterize_forward_cuda ' + desc, 'features / weights differ')
    # ---- backward contract operator on a dense batch with an ARBITRARY (index, weights) pair
    Bd, F = ri(1, 3), ri(1, 300)
    imgd = ((r(Bd, F, 1, 2) - 0.5) * 2 + (r(Bd, F, 3, 2) - 0.5) * 0.3).to(dtype)
    featd = r(Bd, F, 3, D).to(dtype)
    Hb, Wb = ri(1, 90), ri(1, 90)
    sel = torch.randint(-1, F, (Bd, Hb, Wb), generator=g)
    wts = r(Bd, Hb, Wb, 3).to(dtype)
    grad = (r(Bd, Hb, Wb, D) - 0.5).to(dtype)
    interp = torch.zeros(Bd, Hb, Wb, D, dtype=dtype)
    gi, gf, sa = oracle.rasterize_backward(grad, sel, wts, imgd, featd, 1e-8, return_abs=True)
    o = M.rasterize_backward_cuda(grad.cuda(), interp.cuda(), sel.cuda(), wts.cuda(), imgd.cuda(), featd.cuda(), 1e-8)
    tol = 1e-5 if dtype == torch.float32 else 1e-10
    m = elementwise_mismatch(o[0], gi, tol, term_abs_sum=sa)
    if m:
        fail(case, f'rasterize_backward_cuda image gradient (B={Bd} F={F} {Hb}x{Wb} {dtype})', m)
    m = elementwise_mismatch(o[1], gf, tol)
    if m:
        fail(case, f'rasterize_backward_cuda feature gradient (B={Bd} F={F} {Hb}x{Wb} {dtype})', m)
print(f'{n_cases} cases from seed {seed0}: {bad} failed checks, {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if bad else 0)
