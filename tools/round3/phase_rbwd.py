"""Development probe: per-phase tick sums of the rasterizer's backward tile function (library built with `make prof`).
usage: KAMD_LIB_PATH=kaolin_amd/libkaolin_amd_prof.so python tools/round3/phase_rbwd.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
from kaolin_amd import _lib
from kaolin_amd.utils import testing as T
lib = _lib.load()
V, H, W = 8, 1024, 1024
fz, fimg, feats, nz = T.sphere_scene(level=50, num_views=V, device='cuda')
feat = torch.cat(feats, -1).contiguous()
a = fimg.clone().requires_grad_()
go = torch.rand((V, H, W, 3), device='cuda')


def step():
    a.grad = None
    out, soft, idx = kal.render.mesh.dibr_rasterization(H, W, fz, a, feat, nz)
    out.backward(go)


for _ in range(2):
    step()
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
raw.kamd_debug_phase_cycles_rbwd(buf, 1)
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
raw.kamd_debug_phase_cycles_rbwd(buf, 0)
names = ['face_idx load + any-covered barrier', 'LDS table init + barrier', 'loads + jacobian (drained)', 'DPP run merge', 'hash insert + LDS adds (drained)',
         'barrier', 'flush: global atomics issued', 'flush: atomics acknowledged']
tot = sum(buf[:8]) or 1
for i, nm in enumerate(names):
    print(f'  {nm:40s} {buf[i] / n / 1e6:9.3f} Mticks/step  {100.0 * buf[i] / tot:5.1f} %')
print('  (ticks of 10 ns summed over wavefronts; %d wavefront-ticks per step = %.1f us x 8192 resident wavefronts)' % (tot / n, tot / n / 100.0 / 8192))
