"""Development probe: sided distance / chamfer timing at 100k x 100k, grid vs brute."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3


for dist in ('uniform', 'sphere'):
    torch.manual_seed(0)
    if dist == 'uniform':
        p1, p2 = torch.rand(1, 100000, 3, device='cuda'), torch.rand(1, 100000, 3, device='cuda')
    else:
        a, b = torch.randn(1, 100000, 3, device='cuda'), torch.randn(1, 100000, 3, device='cuda')
        p1, p2 = a / a.norm(dim=-1, keepdim=True), b / b.norm(dim=-1, keepdim=True) * 1.01
    lib = _lib.load(); lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    ms = t(lambda: kal._C.metrics.sided_distance_forward_cuda(p1, p2))
    lib.kamd_profile_enable(0)
    print(dist, 'grid fwd ms', round(ms, 4), {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
    os.environ['KAMD_SIDED_DISTANCE'] = 'brute'
    print(dist, 'brute fwd ms', round(t(lambda: kal._C.metrics.sided_distance_forward_cuda(p1, p2)), 4))
    del os.environ['KAMD_SIDED_DISTANCE']
