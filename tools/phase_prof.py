"""Development probe: per-phase cycle sums of the instrumented soft-mask kernels (library built with `make prof`,
selected through KAMD_LIB_PATH).  usage: KAMD_LIB_PATH=kaolin_amd/libkaolin_amd_prof.so python tools/phase_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
lib = _lib.load()
V, H, W = 8, 1024, 1024
scene = os.environ.get('KAMD_PROF_SCENE', 'sphere')   # sphere | knot | knot_shuffled | bowl (the knot scene's ~170 image-sized faces alone)
if scene == 'bowl':
    kv, kf = T.knot_mesh()
    fz, fimg, feats, nz = T.mesh_scene(kv, kf[2 * 440 * 48 + 20 * 12 * 12 + 20 * 14 * 14:], V, 'cuda', torch.float, 0, 2.5)
elif scene == 'knot_shuffled':
    fz, fimg, feats, nz = T.mesh_scene(*T.scene_mesh('knot_shuffled'), V, 'cuda', torch.float, 0, 2.5)
else:
    fz, fimg, feats, nz = (T.knot_scene(num_views=V, device='cuda') if scene == 'knot' else T.sphere_scene(level=50, num_views=V, device='cuda'))
feat = torch.cat(feats, -1).contiguous()
for _ in range(2):
    kal.render.mesh.dibr_rasterization(H, W, fz, fimg, feat, nz)
torch.cuda.synchronize()
buf, buf2, buf3 = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 16)()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.kamd_debug_phase_cycles(buf, 1)
raw.kamd_debug_phase_cycles_raster(buf2, 1)
raw.kamd_debug_phase_cycles_bin(buf3, 1)
n = 5
lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
for _ in range(n):
    kal.render.mesh.dibr_rasterization(H, W, fz, fimg, feat, nz)
torch.cuda.synchronize()
lib.kamd_profile_enable(0)
print('kernel us in this (instrumented) build:', {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()})
raw.kamd_debug_phase_cycles(buf, 0)
raw.kamd_debug_phase_cycles_raster(buf2, 0)
raw.kamd_debug_phase_cycles_bin(buf3, 0)
print('bin_faces: phases (index math | vertex loads | raster record incl. z loads | raster range | soft record | soft range | raster lists | soft lists) Mticks/step', [round(buf3[i] / n / 1e6, 1) for i in range(8)],
      'of the lists: big-list appends', round(buf3[8] / n / 1e6, 1), 'medium faces', round(buf3[9] / n / 1e6, 1), 'merging loop', round(buf3[15] / n / 1e6, 1), ';',
      'longest wavefront', buf3[10], 'ticks (10 ns each); wavefronts > 1000 ticks', buf3[11] / n, '> 2500', buf3[12] / n, 'of', buf3[13] / n, '; mean wavefront', round(buf3[14] / max(buf3[13], 1) / 100.0, 2), 'us')
for title, b, names in (
        ('soft_select', buf, ['setup', 'order entries', 'stream+cull', 'chunk masks', 'accept+transposes', 'pair write', 'tail']),
        ('raster_tile', buf2, ['setup (count known)', 'entries, ids, records -> LDS', 'cull + uniform edge loop', 'exact walks', 'loop tail (barriers)',
                               'feature gather + idx store', 'output barrier', 'LDS staging + row stores', 'worklist tickets + drain', 'background tile'])):
    tot = sum(b[:10]) or 1
    print(title)
    for i, nm in enumerate(names):
        print(f'  {nm:28s} {b[i] / n / 1e6:10.2f} Mticks/step  {100.0 * b[i] / tot:5.1f} %')
    if title == 'raster_tile' and b[10]:
        print(f'  wavefronts of tiles with candidates {b[10] / n:.0f}: {sum(b[:9]) / b[10]:.0f} ticks each; of background tiles {b[11] / n:.0f}: {b[9] / max(b[11], 1):.0f} ticks each')
