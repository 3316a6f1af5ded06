"""Development probe: raster_backward on an image with no covered pixel (the cost of the 85 % background workgroups) and on
the C4 scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
lib = _lib.load()
V, H, W = 8, 1024, 1024
fz, fimg, feats, nz = T.sphere_scene(level=50, num_views=V, device='cuda')
feat = torch.cat(feats, -1).contiguous()
out, idx, wts = kal._C.render.mesh.rasterize_forward_fused(H, W, fz, fimg, feat, nz >= 0, 1000, 1e-8)
g = torch.rand_like(out)
none = torch.full_like(idx, -1)
for name, fi in (('no covered pixel', none), ('C4 scene', idx)):
    for need in (False, True):
        for _ in range(3):
            kal._C.render.mesh.rasterize_backward_cuda(g, out, fi, wts, fimg, feat, 1e-8, need_feature_grad=need)
        torch.cuda.synchronize()
        lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
        for _ in range(10):
            kal._C.render.mesh.rasterize_backward_cuda(g, out, fi, wts, fimg, feat, 1e-8, need_feature_grad=need)
        torch.cuda.synchronize()
        lib.kamd_profile_enable(0)
        print(name, 'feature grad' if need else 'no feature grad', {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
