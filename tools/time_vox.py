"""Development probe: trianglemeshes_to_voxelgrids at the C5 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils.testing import geodesic_sphere
for level, res in ((50, 256), (2, 256), (50, 512)):
    v, f = geodesic_sphere(level)
    v = v.float()[None].cuda(); f = f.cuda()
    for _ in range(3): kal.ops.conversions.trianglemeshes_to_voxelgrids(v, f, res)
    lib = _lib.load(); lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): g = kal.ops.conversions.trianglemeshes_to_voxelgrids(v, f, res)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    lib.kamd_profile_enable(0); print({k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
    print(f'voxelgrid F={f.shape[0]} res={res}: {dt*1e6:.1f} us  occupied {int(g.sum())}  write GB/s {res**3*4/dt/1e9:.0f}')
