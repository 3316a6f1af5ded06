"""Development probe: trianglemeshes_to_voxelgrids at the C5 shape (and a coarse mesh, and 512^3).  KAMD_VOX_FUSED=2: the two-launch form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils.testing import geodesic_sphere
lib = _lib.load()
conv = kal.ops.conversions.trianglemeshes_to_voxelgrids
for level, res in ((50, 256), (2, 256), (50, 512)):
    v, f = geodesic_sphere(level)
    v = v.float()[None].cuda(); f = f.cuda()
    for _ in range(5): conv(v, f, res)
    # the operator as users call it: events around 50 calls
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): g = conv(v, f, res)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    for _ in range(10): g = conv(v, f, res)
    torch.cuda.synchronize()
    lib.kamd_profile_enable(0)
    print(f'voxelgrid F={f.shape[0]} res={res}: {us:.1f} us per call  occupied {int(g.sum())}  write GB/s {res**3*4/us/1e3:.0f}  kernels',
          {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
