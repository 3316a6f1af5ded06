import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, kaolin_amd as kal
from test_dibr_oracle import SIMPLE_IMG, SIMPLE_Z
m = kal._C.render.mesh
img = torch.tensor(SIMPLE_IMG).cuda(); z = torch.tensor(SIMPLE_Z).cuda()
_, face_idx = kal.render.mesh.rasterize(35, 31, z, img, torch.zeros(z.shape + (1,), device='cuda'))
scaled = img * 1000.
bbox = torch.cat([scaled.min(-2)[0] - 200., scaled.max(-2)[0] + 200.], -1)
soft, prob, idx, typ = m.dibr_soft_mask_forward_cuda(scaled, bbox, face_idx, 7000., 30, 1000.)
soft2, hits = m.dibr_soft_mask_forward_lean(scaled, bbox, face_idx, 7000., 30, 1000.)
print('soft equal', torch.equal(soft, soft2), 'n_items', int(hits[5]), 'counts', hits[4][:int(hits[5])].tolist())
lp, lf, lpr, lt = m.hit_list_entries(hits, 30)
print('entries', lp.numel(), 'expected', int((idx >= 0).sum()))
g = torch.ones_like(soft)
a = m.dibr_soft_mask_backward_cuda(g, soft, face_idx, prob, idx, typ, scaled, 7000., 1000.)
b = m.dibr_soft_mask_backward_lean(g, soft2, hits, scaled, 7000., 30, 1000.)
print(a.flatten().tolist()); print(b.flatten().tolist())
