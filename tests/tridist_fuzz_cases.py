"""Randomised bit-exactness cases of point_to_mesh_distance's two searches (the Hilbert-tile sweep and the all-pairs kernels, forced in
turn) against the all-pairs oracle (test infrastructure; tests/test_triangle_distance_fuzz.py, tools/round4/fuzz_tridist.py): triangle
soups of mixed sizes, DEGENERATE faces (two equal vertices, a point, collinear vertices -- the reference's computed distance to such a
face can be far below the true one, and a search that culls with bounds on the true distance loses it: found by this sweep in round 4),
duplicated faces, flat meshes, clusters with far queries, huge / tiny coordinates; fp32 / fp64; the sweep's three workgroup sizes."""
import os

import torch

import oracle


def _same(a, b):
    return torch.equal(torch.nan_to_num(a.double(), nan=-7.), torch.nan_to_num(b.double(), nan=-7.))


def check_case(case):
    """-> (description, list of mismatch messages)"""
    import kaolin_amd as kal
    g = torch.Generator().manual_seed(case)
    r = lambda *s: torch.rand(*s, generator=g)
    rn = lambda *s: torch.randn(*s, generator=g)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    dtype = torch.float64 if case % 3 == 2 else torch.float32
    N, F = ri(1000, 9000), ri(64, 4000)
    kind = ['soup', 'mixed_sizes', 'degenerate', 'duplicates', 'flat', 'clusters', 'huge', 'tiny'][ri(0, 7)]
    c = r(F, 1, 3)
    fv = c + (r(F, 3, 3) - 0.5) * 0.05
    pts = r(N, 3)
    if kind == 'soup':
        fv = rn(F, 3, 3)
        pts = rn(N, 3)
    elif kind == 'mixed_sizes':
        fv = c + (r(F, 3, 3) - 0.5) * (10.0 ** (r(F, 1, 1) * 4 - 3))
    elif kind == 'degenerate':
        fv[::3, 2] = fv[::3, 1]                     # two equal vertices
        fv[1::5] = fv[1::5, :1]                     # a point
        fv[2::7, 2] = (fv[2::7, 0] + fv[2::7, 1]) / 2   # collinear
        # Faces 512, 1024, ... take part: the reference re-seeds its running best at every block of 512 faces, so a block whose
        # FIRST face yields NaN for a query is ignored as a whole for it (unbatched_triangle_distance_cuda.cu:303,310) -- the GPU
        # follows since round 5 (td_reseed_*), and half of the cases put a face with v1 == v2 (NaN for the queries beyond v1) there
        if (case // 2) % 2 == 1:
            fv[512::512, 1] = fv[512::512, 0]
    elif kind == 'duplicates':
        fv = fv[torch.randint(0, max(F // 6, 1), (F,), generator=g)]
    elif kind == 'flat':
        fv[..., 2] = 0.5
    elif kind == 'clusters':
        fv = torch.cat([fv[:F // 2] * 0.05 + 4.0, fv[F // 2:] * 0.05 - 3.0])
        pts = torch.cat([r(N // 2, 3) * 10 - 5, rn(N - N // 2, 3) * 0.1 + 4.0])
    elif kind == 'huge':
        s = 1e12 if dtype == torch.float64 else 1e6
        fv, pts = fv * s, pts * s
    elif kind == 'tiny':
        fv, pts = fv * 1e-6, pts * 1e-6
    fv, pts = fv.to(dtype), pts.to(dtype)
    d_ref, i_ref, t_ref = oracle.triangle_distance_forward(pts, fv, omp=True)
    os.environ['KAMD_TRIANGLE_DISTANCE'] = 'sweep' if case % 2 == 0 else 'brute'
    os.environ['KAMD_TS_THREADS'] = str([64, 128, 256][case % 3])
    try:
        d, i, t = kal.metrics.trianglemesh._UnbatchedTriangleDistanceCuda.apply(pts.cuda(), fv.cuda())
    finally:
        del os.environ['KAMD_TRIANGLE_DISTANCE']
        del os.environ['KAMD_TS_THREADS']
    msgs = []
    if not torch.equal(i.cpu(), i_ref):
        msgs.append(f'face index differs at {int((i.cpu() != i_ref).sum())} queries')
    if not torch.equal(t.cpu().to(t_ref.dtype), t_ref):
        msgs.append(f'dist_type differs at {int((t.cpu().to(t_ref.dtype) != t_ref).sum())}')
    if not _same(d.cpu(), d_ref):
        msgs.append('distances differ')
    return f'{kind} {dtype} N={N} F={F}', msgs
