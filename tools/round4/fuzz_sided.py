"""Randomised bit-exactness sweep of the exact grid search behind sided_distance (fp32 / fp64 / fp16) against the all-pairs oracle:
sizes around the thresholds of the grid path, degenerate clouds (a line, a single point repeated, extents of 1e-30 and 1e30, outliers,
duplicates, all queries far away).  usage (GPU box): python tools/round4/fuzz_sided.py [n_cases] [first_seed]"""
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
    dtype = [torch.float32, torch.float32, torch.float64, torch.float16][case % 4]
    B = ri(1, 2)
    N, M = ri(2040, 9000), ri(8180, 40000)
    kind = ['uniform', 'line', 'one_point', 'tiny', 'huge', 'outliers', 'duplicates', 'far_queries', 'plane'][ri(0, 8)]
    p1, p2 = r(B, N, 3), r(B, M, 3)
    if kind == 'line':
        p2 = r(B, M, 1) * torch.tensor([1., 2., -1.]) + 0.5
    elif kind == 'one_point':
        p2 = torch.ones(B, M, 3) * 0.3
        p2[:, M // 2:] += 1e-3 * r(B, M - M // 2, 3)
    elif kind == 'tiny':
        s = 1e-30 if dtype != torch.float16 else 1e-4
        p1, p2 = p1 * s, p2 * s
    elif kind == 'huge':
        s = 1e30 if dtype == torch.float64 else (1e15 if dtype == torch.float32 else 100.)
        p1, p2 = (p1 - 0.5) * s, (p2 - 0.5) * s
    elif kind == 'outliers':
        p2[:, ::997] *= (1e6 if dtype != torch.float16 else 200.)
        p1[:, ::13] -= 5.
    elif kind == 'duplicates':
        p2 = p2[:, torch.randint(0, M // 8, (M,), generator=g)]
    elif kind == 'far_queries':
        p1 = p1 + 50.
    elif kind == 'plane':
        p2[..., 1] = 0.125
        p1[..., 1] = 0.125
    # round 5: non-finite points, at the reference's tile starts (target indices 512 k: a NaN distance there hides the tile) and elsewhere,
    # in half of the cases; a third of the cases on the all-pairs kernels
    if case % 2 == 1:
        nbad = ri(1, 12)
        where = torch.randint(0, M // 512, (nbad,), generator=g) * 512
        where[nbad // 2:] += torch.randint(0, 512, (nbad - nbad // 2,), generator=g)
        vals = torch.tensor([float('nan'), float('inf'), float('-inf')])[torch.randint(0, 3, (nbad,), generator=g)]
        p2[torch.randint(0, B, (nbad,), generator=g), where.clamp(max=M - 1), torch.randint(0, 3, (nbad,), generator=g)] = vals
        p1[0, ri(0, N - 1), ri(0, 2)] = float('inf')
        p1[B - 1, ri(0, N - 1), ri(0, 2)] = float('nan')
    p1, p2 = p1.to(dtype), p2.to(dtype)
    d_ref, i_ref = oracle.sided_distance_forward(p1, p2, omp=True)
    if case % 3 == 2:
        os.environ['KAMD_SIDED_DISTANCE'] = 'brute'
    try:
        d, i = kal.metrics.pointcloud.sided_distance(p1.cuda(), p2.cuda())
    finally:
        os.environ.pop('KAMD_SIDED_DISTANCE', None)
    ok_i = torch.equal(i.cpu(), i_ref)
    ok_d = torch.equal(d.cpu(), d_ref) or bool(((d.cpu() == d_ref) | (torch.isnan(d.cpu()) & torch.isnan(d_ref))).all())
    if not (ok_i and ok_d):
        bad += 1
        print(f'case {case} FAILED ({kind} {dtype} B={B} N={N} M={M}): indices differ at {int((i.cpu() != i_ref).sum())}, distances at {int((d.cpu() != d_ref).sum())}', flush=True)
print(f'{n_cases} cases from seed {seed0}: {bad} failed, {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if bad else 0)
