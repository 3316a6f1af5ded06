"""Development probe: the rasterizer's kernels at the C4 shape under ablations (no valid faces = background only)."""
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


def run(name, fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    lib.kamd_profile_enable(0)
    print(name, {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})


none = torch.zeros(nz.shape, dtype=torch.bool, device='cuda')
front = nz >= 0
run('rasterize, no valid face ', lambda: kal.render.mesh.rasterize(H, W, fz, fimg, feat, none))
run('rasterize, front faces   ', lambda: kal.render.mesh.rasterize(H, W, fz, fimg, feat, front))
run('rasterize, all faces     ', lambda: kal.render.mesh.rasterize(H, W, fz, fimg, feat))
run('dibr_rasterization (fwd) ', lambda: kal.render.mesh.dibr_rasterization(H, W, fz, fimg, feat, nz))
x = torch.empty(V * H * W * 8, dtype=torch.float32, device='cuda')
def fill():
    x.zero_()
for _ in range(3): fill()
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(20): fill()
torch.cuda.synchronize()
print('torch zero_ of 268 MB: %.1f us' % ((time.perf_counter() - t) / 20 * 1e6))
