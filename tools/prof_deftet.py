"""Workload for `rocprofv3 --kernel-trace --stats`: deftet forward at 1M pixels x 50k faces (random and grid pixels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd.utils import testing as T
fz, fimg, feats, nz = T.sphere_scene(level=50, num_views=1, device='cuda')
feat = torch.cat(feats, -1).contiguous()
P = 1 << 20
kind = sys.argv[1] if len(sys.argv) > 1 else 'rand'
if kind == 'rand':
    pix = torch.rand(1, P, 2, device='cuda') * 2 - 1
else:
    n = 1024
    x = (2 * torch.arange(n, device='cuda', dtype=torch.float) + 1 - n) / n
    pix = torch.stack(torch.meshgrid(x, -x, indexing='xy'), -1).reshape(1, P, 2).contiguous()
ranges = torch.tensor([[[-10., 0.]]], device='cuda').repeat(1, P, 1)
for _ in range(10):
    out, idx = kal.render.mesh.deftet_sparse_render(pix, ranges, fz, fimg, feat, 30)
torch.cuda.synchronize()
