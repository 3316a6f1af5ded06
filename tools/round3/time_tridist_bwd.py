"""Development probe: point_to_mesh_distance forward + backward at the C5 shape (gradients to the points and the face vertices)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils.testing import geodesic_sphere
lib = _lib.load()
v, f = geodesic_sphere(50)
for n in ([int(a) for a in sys.argv[1:]] or [100000, 1000000]):
    torch.manual_seed(0)
    fv = v.float()[f].cuda()[None].requires_grad_()
    pts = ((torch.rand(1, n, 3) * 1.2 - 0.1).cuda() - 0.5).requires_grad_()
    def step():
        fv.grad = None; pts.grad = None
        d, i, ty = kal.metrics.trianglemesh.point_to_mesh_distance(pts, fv)
        d.sum().backward()
    for _ in range(2):
        step()
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3):
        step()
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    lib.kamd_profile_enable(0)
    print(f'point_to_mesh fwd+bwd {n} x 50000: {dt*1e3:.2f} ms', {k: round(v[0] / v[1], 3) for k, v in _lib.kernel_profile(reset=True).items()})
