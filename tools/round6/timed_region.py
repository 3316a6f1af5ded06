"""Development probe: the bench's timed region with K = 20 steps after W = 5 (the driver's flags), several times in a row -- host time of
every enqueue, wall time of the region, and the GPU's own intervals.  usage: python tools/round6/timed_region.py"""
import gc, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd.utils import testing as T
V, H, W = 8, 1024, 1024
dev = 'cuda'
v, f = T.scene_mesh('sphere', 50)
verts = v.float().to(dev).requires_grad_()
faces = f.to(dev)
cams = T.fibonacci_cameras(V, 2.5).to(dev)
rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams, torch.zeros_like(cams), torch.tensor([[0., 1., 0.]], device=dev).expand(V, -1))
proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
g = torch.Generator().manual_seed(0)
feats3 = torch.cat([torch.rand((1, faces.shape[0], 3, 2), generator=g), torch.ones((1, faces.shape[0], 3, 1))], -1).to(dev).expand(V, -1, -1, -1).contiguous()
G1, G2 = torch.rand((V, H, W, 3), generator=g).to(dev), torch.rand((V, H, W), generator=g).to(dev)


def step():
    verts.grad = None
    fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(verts.unsqueeze(0).expand(V, -1, -1), faces, proj, camera_rot=rot, camera_trans=trans)
    feat, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, feats3, normals[..., 2])
    kal.metrics.render.weighted_sum(feat, G1, soft, G2).backward()


K, WU = 20, 5
for rep in range(8):
    mode = ('gc.collect + synchronize', 'synchronize only', 'gc.collect, 5 more steps, synchronize', 'synchronize twice')[rep % 4]
    for _ in range(WU):
        step()
    if rep % 4 == 0:
        gc.collect()
    if rep % 4 == 2:
        gc.collect()
        for _ in range(5):
            step()
    gc.disable()
    torch.cuda.synchronize()
    if rep % 4 == 3:
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        a = time.perf_counter()
        step()
        ev[i + 1].record()
        host.append((time.perf_counter() - a) * 1e3)
    enq = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    gc.enable()
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    print(f'rep {rep} [{mode}]: wall {wall:.3f} ms = {wall / K:.4f} per step; enqueue {enq:.3f} ms; host per step first 4 {[round(x, 3) for x in host[:4]]} median {sorted(host)[K // 2]:.3f} max {max(host):.3f}; '
          f'gpu intervals first 4 {[round(x, 3) for x in gpu[:4]]} median {sorted(gpu)[K // 2]:.4f} sum {sum(gpu):.3f}')
