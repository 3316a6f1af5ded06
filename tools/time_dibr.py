"""Quick timing of the DIB-R path at C2 / C4 shapes (not the bench: a development probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd.utils import testing as T


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


for level, views, H in ((16, 1, 256), (50, 8, 1024)):
    fz, fimg, feats, nz = T.sphere_scene(level=level, num_views=views, device='cuda')
    W = H
    a = fimg.clone().requires_grad_()
    f = [x.clone().requires_grad_() for x in feats]
    g1 = torch.rand(views, H, W, 3, device='cuda')
    g2 = torch.rand(views, H, W, device='cuda')

    def step():
        a.grad = None
        out, soft, idx = kal.render.mesh.dibr_rasterization(H, W, fz, a, f, nz)
        ((torch.cat(out, -1) * g1).sum() + (soft * g2).sum()).backward()

    def fwd():
        with torch.no_grad():
            kal.render.mesh.dibr_rasterization(H, W, fz, a, f, nz)

    def rast():
        with torch.no_grad():
            kal.render.mesh.rasterize(H, W, fz, a, f, nz >= 0)

    ms = timeit(step)
    print(f'level {level} views {views} {H}x{W}: fwd+bwd {ms:.3f} ms  {views*H*W/ms/1e3:.1f} Mpix/s | fwd {timeit(fwd):.3f} ms | rasterize only {timeit(rast):.3f} ms')
