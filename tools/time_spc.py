"""Development probe: unbatched_mesh_to_spc on the 50k-face sphere at several levels (wall clock incl. the host reads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils.testing import geodesic_sphere
lib = _lib.load()
v, f = geodesic_sphere(50)
fv = (v.float() * 1.2)[f].contiguous().cuda()
for level in (6, 8, 9, 10):
    for _ in range(2): out = kal.ops.conversions.unbatched_mesh_to_spc(fv, level)
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): out = kal.ops.conversions.unbatched_mesh_to_spc(fv, level)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    lib.kamd_profile_enable(0)
    prof = {k: (round(v[0] / 5 * 1e3, 1), v[1] // 5) for k, v in _lib.kernel_profile(reset=True).items()}
    print(f'level {level}: {dt*1e3:.3f} ms  voxels {out[1].numel()}  octree bytes {out[0].numel()}  per call (us, launches): {prof}', flush=True)
