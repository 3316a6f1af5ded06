"""One line per library build (KAMD_LIB_PATH): chamfer 100k x 100k -- the bench's step, the operator alone, kernel times
(uniform cube and points on a sphere), each checked against the all-pairs kernels' nearest indices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib

dev = 'cuda'
lib = _lib.load()


def timed(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = [os.environ.get('AB_LABEL', '?')]
for name in ('uniform', 'sphere'):
    g = torch.Generator().manual_seed(0)
    n = 100000
    if name == 'uniform':
        base, p2 = torch.rand((1, n, 3), generator=g).to(dev), torch.rand((1, n, 3), generator=g).to(dev)
    else:
        a, b = torch.randn((1, n, 3), generator=g), torch.randn((1, n, 3), generator=g)
        base, p2 = (a / a.norm(dim=-1, keepdim=True)).to(dev), (b / b.norm(dim=-1, keepdim=True) * 1.01).to(dev)
    p2 = p2.requires_grad_()
    offset = torch.zeros(3, device=dev, requires_grad=True)
    p1_leaf = base.clone().requires_grad_()
    up = torch.ones(1, device=dev)

    def step():
        offset.grad = None
        p2.grad = None
        kal.metrics.pointcloud.chamfer_distance(base + offset, p2).sum().backward()

    def op():
        p1_leaf.grad = None
        p2.grad = None
        kal.metrics.pointcloud.chamfer_distance(p1_leaf, p2).backward(up)

    s_ms, o_ms = timed(step), timed(op)
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    timed(op, 10, 0)
    lib.kamd_profile_enable(0)
    k = {kk.split('(')[0]: round(v[0] / v[1] * 1e3, 1) for kk, v in _lib.kernel_profile(reset=True).items()}
    # exactness: nearest indices equal the all-pairs kernels'
    d, i = kal.metrics.pointcloud.sided_distance(base, p2)
    os.environ['KAMD_SIDED_DISTANCE'] = 'brute'
    d2, i2 = kal.metrics.pointcloud.sided_distance(base, p2)
    del os.environ['KAMD_SIDED_DISTANCE']
    ok = bool(torch.equal(i, i2) and torch.equal(d, d2))
    out.append(f'{name}: step {s_ms:.4f} op {o_ms:.4f} ms {k} exact={ok}')
print(' | '.join(out))
