"""Pins the oracle (oracle/) against the REFERENCE'S OWN pure-torch implementations on random inputs -- beyond the committed golden
fixtures (tests/golden/make_golden.py, same loader).  Runs in the build container only (it imports /root/reference by path, on CPU):
  * sided distance           vs  kaolin.metrics.pointcloud._sided_distance                      (values)
  * triangle distance        vs  kaolin.metrics.trianglemesh._unbatched_naive_point_to_mesh_distance  (distance, face, type)
  * voxelizer                vs  kaolin.ops.conversions.trianglemeshes_to_voxelgrids            (torch.equal)
  * deftet_sparse_render     vs  kaolin.render.mesh.deftet._naive_deftet_sparse_render          (face_idx, features; knum >= hits)
  * rasterize                vs  _naive_deftet_sparse_render(knum=1) as the reference's own tests do (face_idx)
usage: python tools/round4/oracle_vs_reference.py [n_cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch
import oracle
from oracle import voxelgrid as vox_oracle
import _refload
ref = _refload.load_reference()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = {k: 0 for k in ('sided', 'tridist', 'vox', 'deftet', 'raster')}
worst = {k: 0.0 for k in bad}
t0 = time.time()
for case in range(n_cases):
    g = torch.Generator().manual_seed(case)
    r = lambda *s: torch.rand(*s, generator=g)
    rn = lambda *s: torch.randn(*s, generator=g)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    dtype = torch.float64 if case % 2 else torch.float32
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    # ---- sided distance
    B, N, M = ri(1, 3), ri(1, 300), ri(1, 300)
    p1, p2 = (rn(B, N, 3) * 3).to(dtype), (rn(B, M, 3) * 3).to(dtype)
    d, i = oracle.sided_distance_forward(p1, p2)
    want = ref['pointcloud']._sided_distance(p1, p2)
    e = float(((d - want).abs() / want.abs().clamp_min(1e-30)).max())
    worst['sided'] = max(worst['sided'], e)
    # the index must achieve the reference's minimum
    picked = ((p1 - torch.gather(p2, 1, i[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    if e > tol or float(((picked - want).abs() / want.abs().clamp_min(1e-30)).max()) > tol:
        bad['sided'] += 1
    # ---- triangle distance (non-degenerate soups: the naive version and the kernel differ on faces without area by construction)
    Np, F = ri(1, 400), ri(1, 200)
    if F == 3:
        F = 4      # (the naive version's `torch.cross(e21, e13)` without `dim` crosses along the FACE axis when there are exactly 3 faces)
    fv = (r(F, 1, 3) + (r(F, 3, 3) - 0.5) * (10.0 ** (r(F, 1, 1) * 1.5 - 1.5))).to(dtype)
    pts = r(Np, 3).to(dtype)
    do, io, to = oracle.triangle_distance_forward(pts, fv)
    dn, in_, tn = ref['trianglemesh']._unbatched_naive_point_to_mesh_distance(pts, fv)
    e = float(((do - dn).abs() / dn.abs().clamp_min(1e-30)).max())
    worst['tridist'] = max(worst['tridist'], e)
    same = (io == in_) & (to.long() == tn.long())
    # where the face or the type differs, the two candidates must be equally near (a tie within rounding)
    # (the reference's CUDA kernel, and the oracle with it, keeps the squared distance in a `float` even for double inputs
    # (unbatched_triangle_distance_cuda.cu:301): 6e-8 against the naive version's double)
    if e > (1.2e-7 if dtype == torch.float64 else 5e-4) or float((~same).float().mean()) > (0.0 if dtype == torch.float64 else 0.02):
        bad['tridist'] += 1
        print(f'case {case} tridist {dtype}: worst rel {e:.3g}, face/type differs at {int((~same).sum())} of {Np}', flush=True)
    # ---- voxelizer
    Bv, V, Fv, R = ri(1, 2), ri(3, 60), ri(1, 80), ri(2, 40)
    verts = (r(Bv, V, 3) * (10.0 ** (r(Bv, 1, 1) * 2 - 1))).to(dtype)
    faces = torch.randint(0, V, (Fv, 3), generator=g)
    wantv = ref['conv_trianglemesh'].trianglemeshes_to_voxelgrids(verts, faces, R)
    gotv = vox_oracle.trianglemeshes_to_voxelgrids(verts, faces, R)
    if not torch.equal(gotv, wantv.to(gotv.dtype)):
        bad['vox'] += 1
        print(f'case {case} vox: {int((gotv != wantv.to(gotv.dtype)).sum())} voxels differ (B={Bv} V={V} F={Fv} R={R} {dtype})', flush=True)
    # ---- deftet (knum above the deepest pixel: the naive oracle keeps the NEAREST knum, the operator the first knum in mesh order)
    Bd, Fd, P, D = 1, ri(1, 60), ri(1, 300), ri(1, 3)
    img = ((r(Bd, Fd, 1, 2) - 0.5) * 2 + (r(Bd, Fd, 3, 2) - 0.5) * 0.6).to(dtype)
    z = -(r(Bd, Fd, 3) * 3 + 0.1).to(dtype)
    feat = r(Bd, Fd, 3, D).to(dtype)
    pix = ((r(Bd, P, 2) - 0.5) * 2.2).to(dtype)
    rng = torch.cat([torch.full((Bd, P, 1), -10.), torch.zeros(Bd, P, 1)], -1).to(dtype)
    K = Fd
    od = oracle.deftet_sparse_render(pix, rng, z, img, feat, knum=K)
    nf, ni = ref['deftet']._naive_deftet_sparse_render(pix, rng, z, img, feat, K)
    if not torch.equal(od['face_idx'], ni):
        # equal depths may be ordered either way by the reference's argsort: compare as sets per pixel
        if not torch.equal(od['face_idx'].sort(-1)[0], ni.sort(-1)[0]):
            bad['deftet'] += 1
            print(f'case {case} deftet: face sets differ', flush=True)
    else:
        e = float((od['features'] - nf).abs().max())
        worst['deftet'] = max(worst['deftet'], e)
        # (the naive renderer and the operator treat the eps of the barycentric normaliser differently -- ~eps / area ~ 1e-6 whatever the
        # dtype; the reference's own test compares them at 1e-4-ish tolerances: test_deftet.py:452-489)
        if e > 2e-5:
            bad['deftet'] += 1
            print(f'case {case} deftet {dtype}: features differ by {e:.3g}', flush=True)
    # ---- rasterize vs the naive renderer with knum = 1 on a pixel grid (test_rasterization.py's oracle)
    H, W = ri(4, 24), ri(4, 24)
    xs = (2 * torch.arange(W, dtype=dtype) + 1 - W) / W
    ys = (H - 1 - 2 * torch.arange(H, dtype=dtype)) / H
    grid = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W)], -1).reshape(1, H * W, 2)
    rr = torch.cat([torch.full((1, H * W, 1), -1e6), torch.zeros(1, H * W, 1)], -1).to(dtype)
    _, ni = ref['deftet']._naive_deftet_sparse_render(grid, rr, z, img, feat, 1)
    _, oi, _ = oracle.rasterize(H, W, z, img, feat, None, omp=False)
    diff = int((oi.reshape(-1) != ni.reshape(-1)).sum())
    if diff > 0:
        # the naive renderer keeps the NEAREST hit; the rasterizer too (largest z): a difference means a tie or an edge pixel
        bad['raster'] += 1
        print(f'case {case} raster: {diff} of {H * W} pixels differ ({dtype})', flush=True)
print(f'{n_cases} cases: failures {bad}; worst relative differences {dict((k, float(f"{v:.3g}")) for k, v in worst.items())}; {time.time() - t0:.0f} s', flush=True)
