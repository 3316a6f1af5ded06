"""Development probe: point_to_mesh 200k x 50k on the sphere with 500 faces without area (test_gpu_sweep_with_one_percent_degenerate_faces),
call by call: time, kernel table and (experiment builds, KAMD_TS_STATS=1) the sweep's counters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils.testing import geodesic_sphere
lib = _lib.load()
v, f = geodesic_sphere(50)
fv = v[f].float()
g = torch.Generator().manual_seed(1)
bad = torch.randperm(fv.shape[0], generator=g)[:500]
fvd = fv.clone()
fvd[bad[:250], 1] = fvd[bad[:250], 0]
fvd[bad[250:], 2] = fvd[bad[250:], 1]
pts = (torch.rand(200000, 3, generator=g) * 1.2 - 0.6).cuda()
fvd, fv = fvd.cuda(), fv.cuda()
print('block-first bad faces:', sorted(int(b) for b in bad if int(b) % 512 == 0))
for name, mesh in (('degenerate', fvd), ('clean', fv)):
    for it in range(8):
        lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d, i, t = kal.metrics.trianglemesh._UnbatchedTriangleDistanceCuda.apply(pts, mesh)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        lib.kamd_profile_enable(0)
        print(name, it, f'{dt * 1e3:.2f} ms', {k: round(v[0] / v[1], 3) for k, v in _lib.kernel_profile(reset=True).items()}, flush=True)
