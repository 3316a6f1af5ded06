"""The randomised DIB-R sweep with non-finite and huge coordinates injected (NaN / +-inf / 1e30 in a few vertices or depths): the fused
operator against the oracle -- face_idx equal, features / soft mask equal up to NaN placement and 1e-5.  usage: N first_seed"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
import kaolin_amd as kal

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, t0 = 0, time.time()
for case in range(seed0, seed0 + n_cases):
    g = torch.Generator().manual_seed(case)
    r = lambda *s: torch.rand(*s, generator=g)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    dtype = torch.float64 if case % 4 == 3 else torch.float32
    H, W, B, F = ri(5, 150), ri(5, 150), ri(1, 2), ri(1, 300)
    size = 10.0 ** (r(F) * 2.5 - 2.2)
    img = ((r(B, F, 1, 2) - 0.5) * 2 + (r(B, F, 3, 2) - 0.5) * size.view(1, F, 1, 1)).to(dtype)
    z = -(r(B, F, 3) * 2 + 0.5).to(dtype)
    vals = [float('nan'), float('inf'), float('-inf'), 1e30, -1e30, 1e-30]
    for _ in range(ri(1, 6)):
        b, f, v = ri(0, B - 1), ri(0, F - 1), ri(0, 2)
        if ri(0, 2) == 0:
            z[b, f, v] = vals[ri(0, 5)]
        else:
            img[b, f, v, ri(0, 1)] = vals[ri(0, 5)]
    feat = r(B, F, 3, 2).to(dtype)
    nz = torch.ones(B, F, dtype=dtype)
    knum = [30, 3][ri(0, 1)]
    if os.environ.get('FUZZ_PROGRESS'):
        print('case', case, H, W, B, F, dtype, flush=True)
    ref = oracle.dibr_rasterization(H, W, z, img, feat, nz, knum=knum, omp=True)
    out, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, z.cuda(), img.cuda(), feat.cuda(), nz.cuda(), knum=knum)
    torch.cuda.synchronize()
    msgs = []
    if not torch.equal(face_idx.cpu(), ref['face_idx']):
        msgs.append(f'face_idx differs at {int((face_idx.cpu() != ref["face_idx"]).sum())} pixels')
    else:
        a, b_ = out.cpu().double(), ref['features'].double()
        same_nan = torch.equal(torch.isnan(a), torch.isnan(b_))
        if not same_nan or not torch.equal(torch.nan_to_num(a, nan=0., posinf=1e300, neginf=-1e300), torch.nan_to_num(b_, nan=0., posinf=1e300, neginf=-1e300)):
            msgs.append('features differ')
        a, b_ = soft.cpu().double(), ref['soft_mask'].double()
        if not torch.equal(torch.isnan(a), torch.isnan(b_)):
            msgs.append(f'soft mask NaN placement differs at {int((torch.isnan(a) != torch.isnan(b_)).sum())} pixels')
        else:
            d = (torch.nan_to_num(a) - torch.nan_to_num(b_)).abs()
            if not bool((d <= 1e-5 * torch.nan_to_num(b_).abs() + 4 * torch.finfo(dtype).eps).all()):
                msgs.append(f'soft mask differs by up to {float(d.max()):.3g}')
    if msgs:
        bad += 1
        print(f'case {case} FAILED ({H}x{W} B={B} F={F} {dtype} knum={knum}):', '; '.join(msgs), flush=True)
print(f'{n_cases} cases from seed {seed0}: {bad} failed, {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if bad else 0)
