"""Development probe: a few DIB-R steps at the C4 shape (8 views, 1024^2, 50k faces) for rocprofv3
(--kernel-trace --stats, or --pmc ... in its own run).  Prints the library's per-kernel event timings."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
lib = _lib.load()
V, H, W = 8, 1024, 1024
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
fz, fimg, feats, nz = T.sphere_scene(level=50, num_views=V, device='cuda')
feat = torch.cat(feats, -1).contiguous()
a = fimg.clone().requires_grad_()
G1 = torch.rand(V, H, W, 3, device='cuda').reshape(-1)
G2 = torch.rand(V, H, W, device='cuda').reshape(-1)


def step():
    a.grad = None
    f, soft, idx = kal.render.mesh.dibr_rasterization(H, W, fz, a, feat, nz)
    (torch.dot(f.reshape(-1), G1) + torch.dot(soft.reshape(-1), G2)).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
for _ in range(steps):
    step()
torch.cuda.synchronize()
lib.kamd_profile_enable(0)
print({k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
