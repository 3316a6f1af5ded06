"""Development probe: the chamfer and DIB-R training steps captured into a HIP graph (torch.cuda.graph) and replayed.

Every operator enqueues on torch's current stream through the C ABI and allocates through torch's allocator, so a whole
forward + backward step is capturable.  Replay removes the host from the step: what remains is the GPU-side floor."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd.utils import testing as T

dev = torch.device('cuda')


def eager_and_graph(name, step, outputs, reps=50):
    """`step()` reads static inputs and leaves its results in the tensors `outputs()` returns."""
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        step()
    enq = (time.perf_counter() - t) / reps
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / reps
    want = [o.clone() for o in outputs()]
    print(f'{name}: eager ok', flush=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(f'{name}: side-stream warm-up ok', flush=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    torch.cuda.synchronize()
    print(f'{name}: capture ok', flush=True)
    graph.replay()
    torch.cuda.synchronize()
    print(f'{name}: first replay ok', flush=True)
    got = outputs()
    # difference to the eager result relative to the largest entry, next to the same figure between two eager runs
    # (float atomics make the gradient sums order-dependent)
    same = [float((g - w).abs().max() / w.abs().max()) for g, w in zip(got, want)]
    got = [g.clone() for g in got]
    step()
    torch.cuda.synchronize()
    noise = [float((o - w).abs().max() / w.abs().max()) for o, w in zip(outputs(), want)]
    t = time.perf_counter()
    for _ in range(reps):
        graph.replay()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t) / reps
    print(f'{name}: eager {eager*1e3:.4f} ms/step (host enqueue {enq*1e3:.4f}), graph replay {replay*1e3:.4f} ms/step, '
          f'max rel. difference to eager: {same} (eager vs eager: {noise})', flush=True)


def chamfer():
    n = 100000
    g = torch.Generator().manual_seed(0)
    base = torch.rand((1, n, 3), generator=g).to(dev)
    p2 = torch.rand((1, n, 3), generator=g).to(dev).requires_grad_()
    offset = torch.zeros(3, device=dev, requires_grad=True)
    grads = [torch.zeros_like(offset), torch.zeros_like(p2)]

    def step():
        loss = kal.metrics.pointcloud.chamfer_distance(base + offset, p2).sum()
        go, gp = torch.autograd.grad(loss, [offset, p2])
        grads[0].copy_(go)
        grads[1].copy_(gp)
    eager_and_graph('chamfer 100k x 100k fwd+bwd', step, lambda: grads)


def dibr():
    V, H, W = 8, 1024, 1024
    verts, faces = T.geodesic_sphere(50)
    verts = verts.float().to(dev).requires_grad_()
    faces = faces.to(dev)
    F = faces.shape[0]
    cams = T.fibonacci_cameras(V, 2.5).to(dev)
    up = torch.tensor([[0., 1., 0.]], device=dev).repeat(V, 1)
    rot, trans = kal.render.camera.generate_rotate_translate_matrices(cams, torch.zeros((V, 3), device=dev), up)
    proj = kal.render.camera.generate_perspective_projection(math.pi / 4).to(dev)
    g = torch.Generator().manual_seed(0)
    uv = torch.rand((1, F, 3, 2), generator=g).to(dev).expand(V, -1, -1, -1).contiguous()
    feats3 = torch.cat([uv, torch.ones((V, F, 3, 1), device=dev)], dim=-1).contiguous()
    G1f = torch.rand((V, H, W, 3), generator=g).to(dev).reshape(-1)
    G2f = torch.rand((V, H, W), generator=g).to(dev).reshape(-1)
    grad = [torch.zeros_like(verts)]

    def step():
        fv_cam, fv_img, normals = kal.render.mesh.prepare_vertices(
            verts.unsqueeze(0).expand(V, -1, -1), faces, proj, camera_rot=rot, camera_trans=trans)
        feat, soft, _ = kal.render.mesh.dibr_rasterization(H, W, fv_cam[..., 2], fv_img, feats3, normals[..., 2])
        loss = torch.dot(feat.reshape(-1), G1f) + torch.dot(soft.reshape(-1), G2f)
        grad[0].copy_(torch.autograd.grad(loss, [verts])[0])
    eager_and_graph('DIB-R 8 views 1024^2 fwd+bwd', step, lambda: grad)


if __name__ == '__main__':
    which = sys.argv[1:] or ['chamfer', 'dibr']
    for fn in [f for f in (chamfer, dibr) if f.__name__ in which]:
        try:
            fn()
        except Exception as e:     # a failed capture must not hide the other measurement
            print(f'{fn.__name__}: FAILED {type(e).__name__}: {e}', flush=True)
            torch.cuda.synchronize()
