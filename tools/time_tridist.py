"""Development probe: point_to_mesh_distance at the C5 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
lib = _lib.load()
from kaolin_amd.utils.testing import geodesic_sphere

v, f = geodesic_sphere(50)
fv = v.float()[f].cuda()[None]
torch.manual_seed(0)
for n in ([int(a) for a in sys.argv[1:]] or [100000, 1000000]):
    pts = (torch.rand(1, n, 3) * 1.2 - 0.1).cuda() - 0.5
    for _ in range(2):
        kal.metrics.trianglemesh.point_to_mesh_distance(pts, fv)
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        d, i, ty = kal.metrics.trianglemesh.point_to_mesh_distance(pts, fv)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 3
    lib.kamd_profile_enable(0)
    print({k: round(v[0] / v[1], 3) for k, v in _lib.kernel_profile(reset=True).items()})
    print(f'point_to_mesh {n} x 50000: {dt*1e3:.2f} ms  {n*50000/dt/1e9:.1f} Gpairs/s  types {torch.bincount(ty[0].long(), minlength=7).tolist()}')
