"""Development probe: where the HOST time of the bench's steps goes (cProfile over the C4 step and the C3 chamfer step / operator,
the GPU left to run behind).  usage: python tools/round6/host_profile.py [steps]"""
import cProfile, io, math, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib, distributed as D
from kaolin_amd.utils import testing as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
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
reducer = D.SharedGradientReducer([verts])


def dibr_step():
    verts.grad = None
    feats3.grad = None
    fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(verts.unsqueeze(0).expand(V, -1, -1), faces, proj, camera_rot=rot, camera_trans=trans)
    feat, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, feats3, normals[..., 2])
    kal.metrics.render.weighted_sum(feat, G1, soft, G2).backward()
    reducer.wait()


N = 100000
base = torch.rand((1, N, 3), device=dev)
p2 = torch.rand((1, N, 3), device=dev).requires_grad_()
offset = torch.zeros(3, device=dev, requires_grad=True)
creducer = D.SharedGradientReducer([offset])
p1_leaf = base.clone().requires_grad_()
upstream = torch.ones(1, device=dev)


def chamfer_step():
    offset.grad = None
    p2.grad = None
    kal.metrics.pointcloud.chamfer_distance(base + offset, p2).sum().backward()
    creducer.wait()


def chamfer_operator():
    p1_leaf.grad = None
    p2.grad = None
    kal.metrics.pointcloud.chamfer_distance(p1_leaf, p2).backward(upstream)


def idle_ms(fn, steps=50):
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(steps):
        t0 = time.perf_counter()
        fn()
        tot += time.perf_counter() - t0
        torch.cuda.synchronize()
    return tot / steps * 1e3


for name, fn in (('dibr_step', dibr_step), ('chamfer_step', chamfer_step), ('chamfer_operator', chamfer_operator)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    enq = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e3
    print(f'== {name}: {tot:.4f} ms/step, host enqueue {enq:.4f} ms back to back, {idle_ms(fn):.4f} ms with the GPU idle')
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()     # (the GPU idle: no queue-full waits inside the host's frames)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
    print('\n'.join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
