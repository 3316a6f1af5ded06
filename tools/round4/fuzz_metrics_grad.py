"""Randomised sweeps of the metrics' autograd paths against the oracle: chamfer_distance (value + both gradients; sizes on either side of
the grid search's thresholds, weights, squared or not), sided_distance's backward, point_to_mesh_distance's backward (K8).
usage (GPU box): python tools/round4/fuzz_metrics_grad.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
import kaolin_amd as kal
from kaolin_amd.utils.testing import elementwise_mismatch

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, t0 = 0, time.time()
def fail(case, what, msg):
    global bad
    bad += 1
    print(f'case {case} {what} FAILED: {msg}', flush=True)
for case in range(seed0, seed0 + n_cases):
    g = torch.Generator().manual_seed(case)
    r = lambda *s: torch.rand(*s, generator=g)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    # ---- chamfer: value and gradients
    dtype = torch.float64 if case % 4 == 3 else torch.float32
    B = ri(1, 3)
    N, M = [ri(1, 300), ri(1500, 2600), ri(7000, 9500)][ri(0, 2)], [ri(1, 300), ri(1500, 2600), ri(7900, 12000)][ri(0, 2)]
    p1 = (r(B, N, 3) * (1 + 3 * r(1))).to(dtype)
    p2 = (r(B, M, 3) + 0.3 * r(1, 1, 3)).to(dtype)
    if case % 5 == 0:
        p2[:, : M // 2] = p2[:, M // 2: M // 2 + M // 2][:, : M // 2]       # duplicated targets: ties
    w1, w2, squared = [1., 0.3][ri(0, 1)], [1., 2.5][ri(0, 1)], bool(ri(0, 1))
    a, b = p1.cuda().requires_grad_(), p2.cuda().requires_grad_()
    val = kal.metrics.pointcloud.chamfer_distance(a, b, w1, w2, squared)
    up = r(B).to(dtype) + 0.5
    (val * up.cuda()).sum().backward()
    d1, i1 = oracle.sided_distance_forward(p1, p2, omp=True)
    d2, i2 = oracle.sided_distance_forward(p2, p1, omp=True)
    t1, t2 = (d1 if squared else d1.sqrt()), (d2 if squared else d2.sqrt())
    want = w1 * t1.double().mean(-1) + w2 * t2.double().mean(-1)
    if not torch.allclose(val.detach().cpu().double(), want, rtol=5e-6 if dtype == torch.float32 else 1e-12, atol=0):
        fail(case, f'chamfer value (B={B} N={N} M={M} {dtype} squared={squared})', f'{val.tolist()} vs {want.tolist()}')
    # gradient through the definition with the oracle's indices (torch autograd in double)
    q1, q2 = p1.double().requires_grad_(), p2.double().requires_grad_()
    e1 = ((q1 - torch.gather(q2, 1, i1[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    e2 = ((q2 - torch.gather(q1, 1, i2[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    ref = w1 * (e1 if squared else e1.sqrt()).mean(-1) + w2 * (e2 if squared else e2.sqrt()).mean(-1)
    (ref * up.double()).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 1e-9
    for name, got, wantg in (('p1', a.grad, q1.grad), ('p2', b.grad, q2.grad)):
        finite = torch.isfinite(wantg)
        m = elementwise_mismatch(got.cpu().double()[finite], wantg[finite], tol)
        if m:
            fail(case, f'chamfer grad {name} (B={B} N={N} M={M} {dtype} squared={squared})', m)
    # ---- K8: point_to_mesh_distance backward on a random non-degenerate soup
    Np, F = ri(10, 3000), ri(1, 500)
    fv = (r(F, 1, 3) + (r(F, 3, 3) - 0.5) * 0.3).to(dtype)
    pts = r(Np, 3).to(dtype)
    pa, fa = pts.cuda().requires_grad_(), fv.cuda().requires_grad_()
    dist, idx, typ = kal.metrics.trianglemesh._UnbatchedTriangleDistanceCuda.apply(pa, fa)
    gd = r(Np).to(dtype)
    dist.backward(gd.cuda())
    d_ref, i_ref, t_ref = oracle.triangle_distance_forward(pts, fv, omp=True)
    if not (torch.equal(idx.cpu(), i_ref) and torch.equal(typ.cpu().to(t_ref.dtype), t_ref)):
        fail(case, f'K7 (N={Np} F={F} {dtype})', 'index / type differ')
    else:
        gp, gf = oracle.triangle_distance_backward(gd, pts, fv, i_ref, t_ref)
        for name, got, wantg, tl in (('points', pa.grad, gp, 1e-5), ('faces', fa.grad, gf, 2e-5)):
            m = elementwise_mismatch(got, wantg, tl if dtype == torch.float32 else 1e-9)
            if m:
                fail(case, f'K8 grad {name} (N={Np} F={F} {dtype})', m)
print(f'{n_cases} cases from seed {seed0}: {bad} failed checks, {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if bad else 0)
