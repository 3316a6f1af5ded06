"""Development probe: raster_tile_kernel2<float, true> (the fused operator's tile kernel) at C4 under its KAMD_RASTER_MODE
ablations -- time per launch (profile table) or, under rocprofv3 --pmc, one launch pair per mode so that the dispatch
order identifies the mode.  Usage: python tools/round3/raster_insts.py [time|pmc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
lib = _lib.load()
V, H, W = 8, 1024, 1024
fz, fimg, feats, nz = T.sphere_scene(level=50, num_views=V, device='cuda')
feat = torch.cat(feats, -1).contiguous()
MODES = [int(x) for x in os.environ.get('RI_MODES', '0,32,8,1,3,7').split(',') if x != '']   # (the ablation modes need a library built with -DKAMD_RASTER_DEBUG)
what = sys.argv[1] if len(sys.argv) > 1 else 'time'


def call(valid=None):
    if valid is None:
        return kal.render.mesh.dibr_rasterization(H, W, fz, fimg, feat, nz)
    return kal.render.mesh.dibr_rasterization(H, W, fz, fimg, feat, valid)


for mode in MODES + ['nofaces']:
    if mode == 'nofaces':
        os.environ.pop('KAMD_RASTER_MODE', None)
        arg = torch.full_like(nz, -1.0)     # every face back-facing
    else:
        os.environ['KAMD_RASTER_MODE'] = str(mode) if mode else ''
        arg = None
    if what == 'pmc':
        call(arg); call(arg)
        torch.cuda.synchronize()
        continue
    for _ in range(3):
        call(arg)
    torch.cuda.synchronize()
    lib.kamd_profile_reset(); lib.kamd_profile_enable(1)
    for _ in range(10):
        call(arg)
    torch.cuda.synchronize()
    lib.kamd_profile_enable(0)
    prof = {k: round(v[0] / v[1] * 1e3, 1) for k, v in _lib.kernel_profile(reset=True).items()}
    print('mode', mode, {k: prof[k] for k in prof if 'raster' in k or 'bin' in k})
