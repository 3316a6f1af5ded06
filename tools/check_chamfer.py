"""First-contact check of the fused chamfer operator (persistent grid build with grid barriers): small and large clouds,
B > 1, against the composition of two sided_distance calls; prints timings.  Run under `timeout`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
pc = kal.metrics.pointcloud
lib = _lib.load()
for (B, N, M) in [(1, 9000, 9000), (2, 20000, 30010), (1, 100000, 100000), (3, 8192, 70000)]:
    g = torch.Generator().manual_seed(N)
    p1, p2 = torch.rand(B, N, 3, generator=g).cuda(), torch.rand(B, M, 3, generator=g).cuda()
    for squared in (True, False):
        a, b = p1.clone().requires_grad_(), p2.clone().requires_grad_()
        up = torch.rand(B, generator=g).cuda()
        out = pc.chamfer_distance(a, b, 0.7, 1.3, squared=squared)
        out.backward(up)
        torch.cuda.synchronize()
        a2, b2 = p1.clone().requires_grad_(), p2.clone().requires_grad_()
        d1, d2 = pc.sided_distance(a2, b2)[0], pc.sided_distance(b2, a2)[0]
        if not squared:
            d1, d2 = d1.sqrt(), d2.sqrt()
        ref = 0.7 * d1.mean(-1) + 1.3 * d2.mean(-1)
        ref.backward(up)
        torch.cuda.synchronize()
        ev = float(((out - ref).abs() / ref.abs()).max())
        ea = float((a.grad - a2.grad).abs().max() / a2.grad.abs().max())
        eb = float((b.grad - b2.grad).abs().max() / b2.grad.abs().max())
        print(f'B={B} N={N} M={M} squared={squared}: value rel err {ev:.2e}, grad rel err {ea:.2e} {eb:.2e}', flush=True)
        assert ev < 2e-6 and ea < 1e-5 and eb < 1e-5
    # forward-only (no gradient pieces) and the plain pair search
    with torch.no_grad():
        v = pc.chamfer_distance(p1, p2)
    d12, d21 = pc._nearest_both_ways(p1, p2)
    assert torch.allclose(v, d12.mean(-1) + d21.mean(-1), rtol=2e-6)
n = 100000
gen = torch.Generator().manual_seed(0)
base = torch.rand((1, n, 3), generator=gen).cuda()
p2 = torch.rand((1, n, 3), generator=gen).cuda().requires_grad_()
offset = torch.zeros(3, device='cuda', requires_grad=True)


def step():
    offset.grad = None
    p2.grad = None
    pc.chamfer_distance(base + offset, p2).sum().backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
enq = (time.perf_counter() - t0) / 50
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
def prof(fn, label):
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    lib.kamd_profile_enable(0)
    print(label, {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()}, flush=True)


def fwd_only():
    with torch.no_grad():
        pc.chamfer_distance(base, p2)


if os.environ.get('KAMD_CHECK_SPLIT'):
    prof(fwd_only, 'forward, value only:')
    prof(lambda: pc.chamfer_distance(base, p2), 'forward with gradient pieces:')
    prof(lambda: pc._nearest_both_ways(base, p2), 'plain pair search:')
lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
for _ in range(10):
    step()
torch.cuda.synchronize()
lib.kamd_profile_enable(0)
print(f'chamfer 100k x 100k step: {dt * 1e3:.4f} ms (host enqueue {enq * 1e3:.4f} ms)',
      {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
print('CHAMFER OK')
