"""Development probe: deftet_sparse_render (forward fused + backward) on the 50k-face sphere for several pixel counts,
random and image-ordered pixel coordinates; per-kernel times from the library's HIP-event hooks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T

lib = _lib.load()
for level in (50, 100):
    fz, fimg, feats, nz = T.sphere_scene(level=level, num_views=1, device='cuda')
    feat = torch.cat(feats, -1).contiguous()
    F = fz.shape[1]
    for P, kind in ((4096, 'rand'), (65536, 'rand'), (1 << 20, 'rand'), (1 << 20, 'grid')):
        torch.manual_seed(0)
        if kind == 'rand':
            pix = torch.rand(1, P, 2, device='cuda') * 2 - 1
        else:
            n = int(P ** 0.5)
            x = (2 * torch.arange(n, device='cuda', dtype=torch.float) + 1 - n) / n
            pix = torch.stack(torch.meshgrid(x, -x, indexing='xy'), -1).reshape(1, P, 2).contiguous()
        ranges = torch.tensor([[[-10., 0.]]], device='cuda').repeat(1, P, 1)
        a = fimg.clone().requires_grad_(); u = feat.clone().requires_grad_()
        K = 30
        G = torch.rand(1, P, K, 3, device='cuda')
        def step():
            a.grad = None; u.grad = None
            out, idx = kal.render.mesh.deftet_sparse_render(pix, ranges, fz, a, u, K)
            out.backward(G)
            return idx
        for _ in range(3): idx = step()
        lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.time() - t) / 10
        lib.kamd_profile_enable(0)
        prof = {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()}
        print(f'F={F} P={P} {kind}: fwd+bwd {dt*1e3:.3f} ms  hits/pixel {float((idx != -1).sum()) / P:.2f}  '
              f'brute-force box tests {P * F / 1e9:.1f} G  {prof}', flush=True)
