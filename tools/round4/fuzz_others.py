"""Randomised parity sweeps of the path's other operators against their oracles: trianglemeshes_to_voxelgrids (torch.equal), check_sign
(torch.equal), deftet_sparse_render (face_idx equal, features 1e-5), unbatched_mesh_to_spc (octree / face ids equal).
usage (GPU box): python tools/round4/fuzz_others.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
from oracle import voxelgrid as vox_oracle
import kaolin_amd as kal
from kaolin_amd.utils.testing import geodesic_sphere

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
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
    # ---- voxelizer: random soups with huge and degenerate faces, several meshes, odd resolutions, given origin / scale
    B, V, F, R = ri(1, 3), ri(3, 200), ri(1, 300), ri(2, 97)
    verts = (r(B, V, 3) * (10.0 ** (r(B, 1, 1) * 2 - 1))).to(dtype)
    faces = torch.randint(0, V, (F, 3), generator=g)
    org = sc = None
    if case % 3 == 1:
        org, sc = (r(B, 3) * 0.2 - 0.1).to(dtype), (r(B) * 2 + 0.5).to(dtype) * verts.abs().amax(dim=(1, 2))
    want = vox_oracle.trianglemeshes_to_voxelgrids(verts, faces, R, org, sc)
    got = kal.ops.conversions.trianglemeshes_to_voxelgrids(verts.cuda(), faces.cuda(), R, None if org is None else org.cuda(), None if sc is None else sc.cuda())
    if not torch.equal(got.cpu(), want):
        fail(case, f'voxelgrid (B={B} V={V} F={F} R={R} {dtype})', f'{int((got.cpu() != want).sum())} voxels differ')
    # ---- check_sign: a closed mesh (sphere, scaled / shifted) + random points incl. exactly on vertices / far away
    v, f = geodesic_sphere(ri(1, 6))
    v = (v * torch.tensor([1.0, 0.5 + r(1).item(), 1.5]) + r(3) - 0.5).to(dtype)[None]
    pts = torch.cat([(r(1, ri(1, 3000), 3) * 4 - 2).to(dtype), v[:, :5], (v[:, :5] + v[:, 5:10]) / 2], dim=1)
    want = oracle.check_sign(v, f, pts, omp=True)
    got = kal.ops.mesh.check_sign(v.cuda(), f.cuda(), pts.cuda())
    if not torch.equal(got.cpu(), want):
        fail(case, f'check_sign ({dtype}, F={f.shape[0]}, N={pts.shape[1]})', f'{int((got.cpu() != want).sum())} points differ')
    # ---- deftet_sparse_render: random faces and pixels, small knum (overflow), restricted depth ranges
    B, F, P, K, D = ri(1, 2), ri(1, 400), ri(1, 3000), [1, 4, 20, 64][ri(0, 3)], ri(1, 4)
    img = ((r(B, F, 1, 2) - 0.5) * 2 + (r(B, F, 3, 2) - 0.5) * (10.0 ** (r(B, F, 1, 1) * 2 - 2))).to(dtype)
    z = -(r(B, F, 3) * 3 + 0.1).to(dtype)
    feat = r(B, F, 3, D).to(dtype)
    pix = ((r(B, P, 2) - 0.5) * 2.2).to(dtype)
    lo = -(r(B, P, 1) * 2 + 1.5)
    rng = torch.cat([lo, lo + r(B, P, 1) * 3], dim=-1).to(dtype)
    want = oracle.deftet_sparse_render(pix, rng, z, img, feat, knum=K, omp=True)
    gf, gi = kal.render.mesh.deftet_sparse_render(pix.cuda(), rng.cuda(), z.cuda(), img.cuda(), feat.cuda(), knum=K)
    if not torch.equal(gi.cpu(), want['face_idx']):
        fail(case, f'deftet (B={B} F={F} P={P} K={K} {dtype})', f'face_idx differs at {int((gi.cpu() != want["face_idx"]).sum())} slots')
    elif not torch.allclose(gf.cpu(), want['features'], rtol=1e-5 if dtype == torch.float32 else 1e-10, atol=1e-6 if dtype == torch.float32 else 1e-12):
        fail(case, f'deftet (B={B} F={F} P={P} K={K} {dtype})', f'features differ by up to {float((gf.cpu() - want["features"]).abs().max()):.3g}')
    # ---- mesh_to_spc: random soups inside [-1, 1]^3 (the operator's domain), levels 1-7
    F, level = ri(1, 300), ri(1, 7)
    fv = ((r(F, 1, 3) - 0.5) * 1.6 + (r(F, 3, 3) - 0.5) * (10.0 ** (r(F, 1, 1) * 2.5 - 2.5))).clamp(-1, 1).float()
    o_oct, o_ids, o_bary = oracle.mesh_to_spc(fv, level, omp=True)
    g_oct, g_ids, g_bary = kal.ops.conversions.unbatched_mesh_to_spc(fv.cuda(), level)
    if not (torch.equal(g_oct.cpu(), o_oct) and torch.equal(g_ids.cpu(), o_ids)):
        fail(case, f'mesh_to_spc (F={F} level={level})', f'octree bytes {g_oct.numel()} vs {o_oct.numel()}, ids {g_ids.numel()} vs {o_ids.numel()}')
print(f'{n_cases} cases from seed {seed0}: {bad} failed checks, {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if bad else 0)
