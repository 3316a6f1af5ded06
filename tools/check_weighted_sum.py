"""weighted_sum on the C4 G-buffer sizes: value / gradients against torch, forward and backward times (events)."""
import torch
import kaolin_amd as kal
dev = 'cuda'
g = torch.Generator().manual_seed(0)
x1 = torch.rand((8, 1024, 1024, 3), generator=g).to(dev).requires_grad_()
w1 = torch.rand((8, 1024, 1024, 3), generator=g).to(dev)
x2 = torch.rand((8, 1024, 1024), generator=g).to(dev).requires_grad_()
w2 = torch.rand((8, 1024, 1024), generator=g).to(dev)
out = kal.metrics.render.weighted_sum(x1, w1, x2, w2)
ref = (x1.double() * w1.double()).sum() + (x2.double() * w2.double()).sum()
print('value rel err', abs(float(out) - float(ref)) / float(ref))
out.backward()
print('grads equal', torch.equal(x1.grad, w1), torch.equal(x2.grad, w2))


def t(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


from kaolin_amd import _C
go = torch.ones((), device=dev)
xd1, xd2 = x1.detach(), x2.detach()
print('fused forward us', round(t(lambda: _C.render.mesh.weighted_sum2_forward(xd1, w1, xd2, w2)), 1),
      'backward us', round(t(lambda: _C.render.mesh.weighted_sum2_backward(go, w1, w2)), 1))
print('torch dots us', round(t(lambda: torch.dot(xd1.reshape(-1), w1.reshape(-1)) + torch.dot(xd2.reshape(-1), w2.reshape(-1))), 1),
      'products us', round(t(lambda: (go * w1, go * w2)), 1))
print('bytes', (x1.numel() + x2.numel()) * 8 / 1e6, 'MB each way')
