"""Development probe: chamfer_distance fwd + bwd at 100k x 100k (the fused operator) with the library's per-kernel events.
usage: python tools/round6/chamfer_kernels.py [steps] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = _lib.load()
for dist in ('uniform', 'sphere'):
    torch.manual_seed(0)
    if dist == 'uniform':
        p1, p2 = torch.rand(B, 100000, 3, device='cuda'), torch.rand(B, 100000, 3, device='cuda')
    else:
        a, b = torch.randn(B, 100000, 3, device='cuda'), torch.randn(B, 100000, 3, device='cuda')
        p1, p2 = a / a.norm(dim=-1, keepdim=True), b / b.norm(dim=-1, keepdim=True) * 1.01
    p1.requires_grad_(); p2.requires_grad_()
    up = torch.ones(B, device='cuda')

    def step():
        p1.grad = None; p2.grad = None
        kal.metrics.pointcloud.chamfer_distance(p1, p2).backward(up)

    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): step()
    torch.cuda.synchronize(); plain = (time.time() - t0) / n
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    for _ in range(n): step()
    torch.cuda.synchronize(); lib.kamd_profile_enable(0)
    print(dist, 'B', B, 'chamfer fwd+bwd ms', round(plain * 1e3, 4), {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
