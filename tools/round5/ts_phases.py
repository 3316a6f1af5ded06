"""Development probe: per-phase wave-time shares of the point_to_mesh sweep kernel (library built with -DKAMD_PHASE_PROF, selected
through KAMD_LIB_PATH).  usage: KAMD_LIB_PATH=kaolin_amd/libkaolin_amd_prof.so python tools/round5/ts_phases.py [queries]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils.testing import geodesic_sphere
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
v, f = geodesic_sphere(50)
fv = v.float()[f].cuda()[None]
torch.manual_seed(0)
pts = (torch.rand(1, n, 3) * 1.2 - 0.1).cuda() - 0.5
for _ in range(2):
    kal.metrics.trianglemesh.point_to_mesh_distance(pts, fv)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
raw.kamd_debug_phase_cycles_ts(buf, 1)
reps = 3
for _ in range(reps):
    kal.metrics.trianglemesh.point_to_mesh_distance(pts, fv)
torch.cuda.synchronize()
raw.kamd_debug_phase_cycles_ts(buf, 0)
names = ['0 setup + workgroup candidate list', '1 A: upper bound from tile spheres', '2 A: faces of the best tile (global)', '3 A: tiles needed + hard append',
         '4 B: marks of a 64-candidate chunk', '5 B: staging a tile', '6 B: walk, lane = query', '7 B: walk, lane = face', '8 loop tails']
tot = sum(buf[:10]) or 1
for i, nm in enumerate(names):
    print(f'  {nm:42s} {buf[i] / reps / 1e6:10.2f} Mticks/call  {100.0 * buf[i] / tot:5.1f} %')
