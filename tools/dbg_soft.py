import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kaolin_amd as kal
from kaolin_amd.utils import testing as T
fz, fimg, feats, nz = T.sphere_scene(level=50, num_views=8, device='cuda')
H = W = 1024
m = kal._C.render.mesh
_, face_idx = kal.render.mesh.rasterize(H, W, fz, fimg, torch.cat(feats, -1), nz >= 0)
scaled = fimg * 1000.
lo, hi = scaled.min(dim=-2)[0] - 20., scaled.max(dim=-2)[0] + 20.
bbox = torch.cat([lo, hi], -1).contiguous()
for _ in range(3):
    soft, hits = m.dibr_soft_mask_forward_lean(scaled, bbox, face_idx, 7000., 30, 1000.)
    g = torch.rand_like(soft)
    m.dibr_soft_mask_backward_lean(g, soft, hits, scaled, 7000., 30, 1000.)
torch.cuda.synchronize()
pix = m.hit_list_entries(hits, 30)[0].long()
n = pix.numel()
upix = torch.unique(pix)
print('hits', n, 'pixels with hits', upix.numel(), 'hits/pixel', n / upix.numel())
sub = ((upix // (H * W)) * 100000000 + ((upix % (H * W)) // W // 4) * 10000 + ((upix % W) // 16))
print('sub-tiles with hits', torch.unique(sub).numel(), 'uncovered frac', float((face_idx < 0).float().mean()))
