"""Development probe (profiling build: make -C kaolin_amd/csrc prof; KAMD_LIB_PATH=kaolin_amd/libkaolin_amd_prof.so): the timeline of ONE
raster_tile launch at C4 from per-tile begin / end stamps of the 100 MHz counter -- lifetimes by candidate count, tiles in flight over
time, when the last tile of each kind ends.  usage: python tools/round6/tile_timeline.py [scene]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
scene = sys.argv[1] if len(sys.argv) > 1 else 'sphere'
V, H, W = 8, 1024, 1024
fz, fimg, feats, nz = (T.knot_scene(num_views=V, device='cuda') if scene == 'knot' else T.sphere_scene(level=50, num_views=V, device='cuda'))
feat = torch.cat(feats, -1).contiguous()
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
for _ in range(3):
    kal.render.mesh.dibr_rasterization(H, W, fz, fimg, feat, nz)
torch.cuda.synchronize()
n = V * (H // 16) * (W // 16)
buf = (ctypes.c_ulonglong * (4 * n))()
raw.kamd_debug_tile_times(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
t0 = t[:, 0].min()
beg, end, cand = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, t[:, 2]      # us
life = end - beg
print(f'{scene}: launch spans {end.max():.1f} us (instrumented build); tiles {n}, with candidates {(cand > 0).sum()}')
for lo, hi in ((0, 0), (1, 16), (17, 32), (33, 64), (65, 128), (129, 192), (193, 100000)):
    m = (cand >= lo) & (cand <= hi)
    if m.any():
        print(f'  candidates {lo:4d}..{min(hi, 9999):4d}: {m.sum():6d} tiles, lifetime mean {life[m].mean():6.2f} us  p50 {np.percentile(life[m], 50):6.2f}  p95 {np.percentile(life[m], 95):6.2f}  max {life[m].max():6.2f};'
              f' first begins {beg[m].min():6.1f}, last begins {beg[m].max():6.1f}, last ends {end[m].max():6.1f}')
face = cand > 0
grid = np.arange(0, end.max() + 2, 2.0)
print('  time us  | tiles with candidates in flight | background tiles in flight | tiles begun so far')
for g in grid:
    print(f'  {g:7.1f}  | {int(((beg <= g) & (end > g) & face).sum()):6d} | {int(((beg <= g) & (end > g) & ~face).sum()):6d} | {int((beg <= g).sum()):6d}')
