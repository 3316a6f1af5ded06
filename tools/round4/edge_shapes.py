"""Zero-sized and non-finite inputs through every operator of the path (must neither hang nor fault; compared with the oracle / the
definition where one exists).  usage (GPU box): timeout 60 python tools/round4/edge_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
import kaolin_amd as kal
from kaolin_amd.utils.testing import geodesic_sphere
bad = 0
def check(name, fn):
    global bad
    try:
        r = fn()
        torch.cuda.synchronize()
        print('ok  ', name, '' if r is None else r, flush=True)
    except Exception as e:      # noqa: BLE001
        bad += 1
        print('FAIL', name, type(e).__name__, str(e)[:200], flush=True)
c = lambda *s: torch.rand(*s).cuda()
pc, tm = kal.metrics.pointcloud, kal.metrics.trianglemesh
def sd(N, M, B=1):
    d, i = pc.sided_distance(c(B, N, 3), c(B, M, 3))
    return tuple(d.shape)
check('sided_distance N=0', lambda: sd(0, 5))
check('sided_distance M=0', lambda: sd(5, 0))
check('sided_distance B=0', lambda: sd(5, 5, 0))
check('sided_distance grid path with NaN / inf points', lambda: (lambda p1, p2: [p1.__setitem__((0, 3), float('nan')), p2.__setitem__((0, 7, 1), float('inf')), torch.equal(pc.sided_distance(p1.cuda(), p2.cuda())[1].cpu(), oracle.sided_distance_forward(p1, p2, omp=True)[1])][-1])(torch.rand(1, 3000, 3), torch.rand(1, 9000, 3)))
check('chamfer N=1 M=1', lambda: float(pc.chamfer_distance(c(2, 1, 3), c(2, 1, 3)).sum()))
check('f_score tiny', lambda: tuple(pc.f_score(c(1, 3, 3), c(1, 4, 3)).shape))
v, f = geodesic_sphere(3)
fv = v.float()[f].cuda()
check('point_to_mesh N=0', lambda: tuple(tm.point_to_mesh_distance(c(1, 0, 3), fv[None])[0].shape))
check('point_to_mesh one face one point', lambda: float(tm.point_to_mesh_distance(c(1, 1, 3), fv[None, :1])[0]))
check('point_to_mesh 70000 points, NaN face, inf point (sweep)', lambda: (lambda pts, ff: [ff.__setitem__((5,), float('nan')), pts.__setitem__((9, 0), float('inf')), torch.equal(kal.metrics.trianglemesh._UnbatchedTriangleDistanceCuda.apply(pts.cuda(), ff.cuda())[1].cpu(), oracle.triangle_distance_forward(pts, ff, omp=True)[1])][-1])(torch.rand(70000, 3) * 2 - 1, geodesic_sphere(12)[0].float()[geodesic_sphere(12)[1]].clone()))
conv = kal.ops.conversions
check('voxelgrid F=0', lambda: float(conv.trianglemeshes_to_voxelgrids(c(1, 5, 3), torch.zeros(0, 3, dtype=torch.long).cuda(), 8).sum()))
check('voxelgrid NaN vertex', lambda: (lambda vv: [vv.__setitem__((0, 2, 1), float('nan')), float(conv.trianglemeshes_to_voxelgrids(vv.cuda(), f.cuda(), 16).sum())][-1])(v.float()[None].clone()))
check('voxelgrid resolution 2', lambda: float(conv.trianglemeshes_to_voxelgrids(v.float()[None].cuda(), f.cuda(), 2).sum()))
check('mesh_to_spc level 1, faces outside the cube', lambda: tuple(x.numel() for x in conv.unbatched_mesh_to_spc((fv * 3).contiguous(), 1)))
check('mesh_to_spc NaN face', lambda: (lambda ff: [ff.__setitem__((0, 0, 0), float('nan')), tuple(x.numel() for x in conv.unbatched_mesh_to_spc(ff, 4))][-1])(fv.clone()))
check('check_sign P=0', lambda: tuple(kal.ops.mesh.check_sign(v.float()[None].cuda(), f.cuda(), c(1, 0, 3)).shape))
check('check_sign NaN point', lambda: (lambda p: [p.__setitem__((0, 0, 0), float('nan')), int(kal.ops.mesh.check_sign(v.float()[None].cuda(), f.cuda(), p.cuda()).sum())][-1])(torch.rand(1, 50, 3)))
dr = kal.render.mesh
fz, fimg, feats, nz = kal.utils.testing.sphere_scene(level=4, num_views=1, device='cuda')
feat = torch.cat(feats, -1)
check('deftet P=0', lambda: tuple(dr.deftet_sparse_render(c(1, 0, 2), c(1, 0, 2), fz, fimg, feat, knum=4)[1].shape))
check('deftet inverted ranges', lambda: int((dr.deftet_sparse_render(c(1, 50, 2) * 2 - 1, torch.tensor([[[-1., -3.]]]).cuda().expand(1, 50, 2).contiguous(), fz, fimg, feat, knum=4)[1] >= 0).sum()))
check('deftet NaN pixel', lambda: (lambda p: [p.__setitem__((0, 3, 0), float('nan')), tuple(dr.deftet_sparse_render(p.cuda(), torch.tensor([[[-10., 0.]]]).cuda().expand(1, 50, 2).contiguous(), fz, fimg, feat, knum=4)[1].shape)][-1])(torch.rand(1, 50, 2) * 2 - 1))
check('dibr 1x1 image', lambda: tuple(dr.dibr_rasterization(1, 1, fz, fimg, feat, nz)[1].shape))
check('dibr knum=1 boxlen=0', lambda: float(dr.dibr_rasterization(33, 17, fz, fimg, feat, nz, knum=1, boxlen=0.)[1].sum()))
check('dibr sigmainv=0', lambda: float(dr.dibr_rasterization(33, 17, fz, fimg, feat, nz, sigmainv=0.)[1].sum()))
check('dibr huge boxlen (every box everywhere)', lambda: float(dr.dibr_rasterization(40, 40, fz, fimg, feat, nz, boxlen=50.)[1].sum()))
check('dibr multiplier<0 via soft mask', lambda: float(dr.dibr_soft_mask(fimg, dr.rasterize(20, 20, fz, fimg, feat)[1], 7000., 0.02, 30, -1000.).sum()))
check('rasterize eps<0 multiplier tiny', lambda: int((dr.rasterize(20, 20, fz, fimg, feat, multiplier=1e-20, eps=-1.)[1] >= 0).sum()))
print('failures:', bad, flush=True)
