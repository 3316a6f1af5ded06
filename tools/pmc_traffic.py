"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass):
two calibration launches of known size (a 1 GiB torch clone = 16 B/lane streaming read+write; the K-buffer fill
kernel = pure 16-B stores) followed by a few bench steps.  tools/parse_traffic.py turns the two CSVs into
profiles/traffic.json."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd.utils import testing as T
x = torch.empty(1 << 28, dtype=torch.float32, device='cuda').normal_()
for _ in range(3):
    y = x.clone()
torch.cuda.synchronize()
fz, fimg, feats, nz = T.sphere_scene(level=50, num_views=8, device='cuda')
H = W = 1024
feat = torch.cat(feats, -1).contiguous()
a = fimg.clone().requires_grad_()
G1 = torch.rand(8, H, W, 3, device='cuda').reshape(-1); G2 = torch.rand(8, H, W, device='cuda').reshape(-1)
for _ in range(3):
    a.grad = None
    f, soft, idx = kal.render.mesh.dibr_rasterization(H, W, fz, a, feat, nz)
    (torch.dot(f.reshape(-1), G1) + torch.dot(soft.reshape(-1), G2)).backward()
# reference-contract K-buffer operators (fill kernel = write calibration)
scaled = fimg * 1000.
bbox = torch.cat([scaled.min(-2)[0] - 20., scaled.max(-2)[0] + 20.], -1).contiguous()
for _ in range(2):
    kal._C.render.mesh.dibr_soft_mask_forward_cuda(scaled, bbox, idx, 7000., 30, 1000.)
torch.cuda.synchronize()
