"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass):
calibration launches of known size first (a 1 GiB torch clone = 16 B/lane streaming read+write; the K-buffer fill
kernel = pure 16-B stores), then exactly STEPS steps of bench.py's DIB-R step (config C4, static features).
tools/parse_traffic.py turns the two CSVs into profiles/traffic.json; everything dispatched after the last
fill_regions_kernel belongs to the steps."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd.utils import testing as T
STEPS = 3
dev = 'cuda'
V, H, W = 8, 1024, 1024
verts, faces = T.geodesic_sphere(50)
verts = verts.float().to(dev).requires_grad_()
faces = faces.to(dev)
F = faces.shape[0]
cams = T.fibonacci_cameras(V, 2.5).to(dev)
rot, trans = kal.render.camera.generate_rotate_translate_matrices(
    cams, torch.zeros((V, 3), device=dev), torch.tensor([[0., 1., 0.]], device=dev).repeat(V, 1))
proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
g = torch.Generator().manual_seed(0)
feats3 = torch.cat([torch.rand((1, F, 3, 2), generator=g).to(dev).expand(V, -1, -1, -1), torch.ones((V, F, 3, 1), device=dev)], -1).contiguous()
G1 = torch.rand((V, H, W, 3), generator=g).to(dev)
G2 = torch.rand((V, H, W), generator=g).to(dev)


def step():
    verts.grad = None
    cam, img, nrm = kal.render.mesh.prepare_vertices(verts.unsqueeze(0).expand(V, -1, -1), faces, proj, camera_rot=rot, camera_trans=trans)
    feat, soft, idx = kal.render.mesh.dibr_rasterization(H, W, cam[..., 2], img, feats3, nrm[..., 2])
    kal.metrics.render.weighted_sum(feat, G1, soft, G2).backward()
    return img, idx


img, idx = step()                      # allocator warm-up (its dispatches precede the calibration marker)
x = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
for _ in range(3):
    y = x.clone()
torch.cuda.synchronize()
del x, y
scaled = img.detach() * 1000.
bbox = torch.cat([scaled.min(-2)[0] - 20., scaled.max(-2)[0] + 20.], -1).contiguous()
for _ in range(2):                     # reference-contract K-buffer operator: fill kernel = write calibration AND the marker
    kal._C.render.mesh.dibr_soft_mask_forward_cuda(scaled, bbox, idx, 7000., 30, 1000.)
torch.cuda.synchronize()
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
