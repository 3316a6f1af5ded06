"""Host-side cost of the chamfer (and DIB-R) step: cProfile over a few hundred eager steps (the chamfer step is host-bound)."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal

dev = torch.device('cuda')
n = 100000
g = torch.Generator().manual_seed(0)
base = torch.rand((1, n, 3), generator=g).to(dev)
p2 = torch.rand((1, n, 3), generator=g).to(dev).requires_grad_()
offset = torch.zeros(3, device=dev, requires_grad=True)


def step():
    offset.grad = None
    p2.grad = None
    kal.metrics.pointcloud.chamfer_distance(base + offset, p2).sum().backward()


def step_min():
    p2.grad = None
    kal.metrics.pointcloud.chamfer_distance(base, p2).backward(ones)


ones = torch.ones(1, device=dev)
for name, fn in (('bench step (offset + sum)', step), ('operator only (p2 grad, given upstream gradient)', step_min)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(300):
        fn()
    enq = (time.perf_counter() - t) / 300
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t) / 300
    print(f'{name}: host enqueue {enq * 1e6:.1f} us/step, wall {tot * 1e6:.1f} us/step', flush=True)

pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:6000])
