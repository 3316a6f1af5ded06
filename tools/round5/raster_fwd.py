"""Development probe: the fused dibr_rasterization FORWARD alone on the C4 scene (8 views, 1024^2, 50k faces), N calls -- the workload of
the rasterizer's counter passes (tools/round5/call5.sh).  usage: python tools/round5/raster_fwd.py [N] [scene]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
scene = sys.argv[2] if len(sys.argv) > 2 else 'sphere'
V, H, W = 8, 1024, 1024
fz, fimg, feats, nz = (T.knot_scene(num_views=V, device='cuda') if scene == 'knot' else T.sphere_scene(level=50, num_views=V, device='cuda'))
feat = torch.cat(feats, -1).contiguous()
lib = _lib.load()
for _ in range(3):
    kal.render.mesh.dibr_rasterization(H, W, fz, fimg, feat, nz)
torch.cuda.synchronize()
lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
t0 = time.time()
for _ in range(n):
    kal.render.mesh.dibr_rasterization(H, W, fz, fimg, feat, nz)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
lib.kamd_profile_enable(0)
print(scene, 'forward ms', round(dt * 1e3, 4), {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
