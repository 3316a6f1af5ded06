"""First GPU contact: lib loads against torch's HIP runtime, sided distance parity + timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd
from kaolin_amd import _lib, _C
import oracle

print('lib', _lib.load().kamd_version(), 'torch', torch.__version__, torch.cuda.get_device_name(0))
dev = 'cuda'
ok = True
for dtype in (torch.float32, torch.float64, torch.float16):
    for (B, N, M) in [(2, 1000, 777), (1, 5000, 4099), (3, 1, 1), (1, 2048, 6000)]:
        torch.manual_seed(0)
        p1 = torch.rand(B, N, 3).to(dtype)
        p2 = torch.rand(B, M, 3).to(dtype)
        d_ref, i_ref = oracle.sided_distance_forward(p1, p2)
        d, i = _C.metrics.sided_distance_forward_cuda(p1.to(dev), p2.to(dev))
        e_d = torch.equal(d.cpu(), d_ref)
        e_i = torch.equal(i.cpu(), i_ref)
        print(dtype, (B, N, M), 'dist bit-exact', e_d, 'idx equal', e_i)
        ok &= e_d and e_i
# fast path parity on a bigger case
torch.manual_seed(1)
p1 = torch.rand(1, 20000, 3); p2 = torch.rand(1, 30011, 3)
t = time.time(); d_ref, i_ref = oracle.sided_distance_forward(p1, p2, omp=True); print('oracle omp s', time.time() - t)
d, i = _C.metrics.sided_distance_forward_cuda(p1.to(dev), p2.to(dev))
print('fast path 20000x30011 dist', torch.equal(d.cpu(), d_ref), 'idx', torch.equal(i.cpu(), i_ref))
ok &= torch.equal(d.cpu(), d_ref) and torch.equal(i.cpu(), i_ref)
# duplicates -> ties must give lowest index
p2d = torch.rand(1, 4096, 3).repeat(1, 4, 1)
p1d = torch.rand(1, 4096, 3)
d_ref, i_ref = oracle.sided_distance_forward(p1d, p2d)
d, i = _C.metrics.sided_distance_forward_cuda(p1d.to(dev), p2d.to(dev))
print('ties', torch.equal(d.cpu(), d_ref), torch.equal(i.cpu(), i_ref), int(i.max()))
ok &= torch.equal(i.cpu(), i_ref)

# timing 100k x 100k
p1 = torch.rand(1, 100000, 3, device=dev); p2 = torch.rand(1, 100000, 3, device=dev)
for _ in range(3):
    _C.metrics.sided_distance_forward_cuda(p1, p2)
torch.cuda.synchronize()
t = time.time()
for _ in range(10):
    d, i = _C.metrics.sided_distance_forward_cuda(p1, p2)
torch.cuda.synchronize()
dt = (time.time() - t) / 10
print(f'sided fwd 100k x 100k: {dt*1e3:.3f} ms  {1e10/dt/1e6:.3e} Mpairs/s  VALU lane-ops/s {6.7e10/dt/1e12:.1f} T')
g = torch.rand(1, 100000, device=dev)
t = time.time()
for _ in range(10):
    _C.metrics.sided_distance_backward_cuda(g, p1, p2, i)
torch.cuda.synchronize()
print('bwd ms', (time.time() - t) / 10 * 1e3)
print('ALL OK' if ok else 'MISMATCH')
