"""Host-side cost of the DIB-R bench step (config C4): enqueue time per stage and a cProfile of a few hundred eager steps."""
import cProfile, pstats, sys, os, io, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd.utils import testing as T

dev = torch.device('cuda')
V, H, W = 8, 1024, 1024
verts, faces = T.geodesic_sphere(int(os.environ.get('FREQ', '50')))
verts = verts.float().to(dev).requires_grad_()
faces = faces.to(dev)
F = faces.shape[0]
cams = T.fibonacci_cameras(V, 2.5).to(dev)
rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams, torch.zeros((V, 3), device=dev),
                                                                  torch.tensor([[0., 1., 0.]], device=dev).repeat(V, 1))
proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
g = torch.Generator().manual_seed(0)
feats = torch.rand((V, F, 3, 3), generator=g).to(dev)
G1 = torch.rand((V, H, W, 3), generator=g).to(dev)
G2 = torch.rand((V, H, W), generator=g).to(dev)
marks = {}


def step(timing=False):
    t0 = time.perf_counter()
    verts.grad = None
    fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(verts.unsqueeze(0).expand(V, -1, -1), faces, proj, camera_rot=rot, camera_trans=trans)
    t1 = time.perf_counter()
    feat, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, feats, normals[..., 2])
    t2 = time.perf_counter()
    loss = kal.metrics.render.weighted_sum(feat, G1, soft, G2)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    if timing:
        for k, v in (('prepare_vertices', t1 - t0), ('dibr_rasterization', t2 - t1), ('weighted_sum', t3 - t2), ('backward', t4 - t3)):
            marks[k] = marks.get(k, 0.0) + v


for _ in range(20):
    step()
torch.cuda.synchronize()
N = 200
# host-only cost: let the GPU drain between steps so that the enqueue never blocks on a full queue
for _ in range(N):
    step(True)
    torch.cuda.synchronize()
print('host enqueue per stage (us, GPU idle when enqueuing):', {k: round(v / N * 1e6, 1) for k, v in marks.items()},
      'total', round(sum(marks.values()) / N * 1e6, 1), flush=True)
t = time.perf_counter()
for _ in range(N):
    step()
enq = (time.perf_counter() - t) / N
torch.cuda.synchronize()
tot = (time.perf_counter() - t) / N
print(f'back to back: host enqueue {enq * 1e6:.1f} us/step, wall {tot * 1e6:.1f} us/step', flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
    torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
print(s.getvalue()[:8000])
