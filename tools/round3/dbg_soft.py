"""Development probe: where dibr_rasterization's soft mask differs from the oracle's on the small ragged case of tests/test_dibr_gpu.py."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import torch
import oracle
import kaolin_amd as kal
from kaolin_amd.utils import testing as T
for dtype in (torch.float, torch.double):
    level, views, H, W, knum, boxlen = 6, 3, 35, 31, 30, 0.2
    fz, fimg, feats, nz = T.sphere_scene(level=level, num_views=views, dtype=dtype, seed=0)
    ref = oracle.dibr_rasterization(H, W, fz, fimg, torch.cat(feats, -1), nz, boxlen=boxlen, knum=knum, omp=True)
    out, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, fz.cuda(), fimg.cuda(), [x.cuda() for x in feats], nz.cuda(), boxlen=boxlen, knum=knum)
    d = (soft.cpu() - ref['soft_mask']).abs()
    bad = (d > 1e-5 * ref['soft_mask'].abs().clamp(min=1e-3)).nonzero()
    print(dtype, 'bad pixels', bad.shape[0], 'of', soft.numel())
    for i in bad.tolist():
        print('  ', i, 'got %.9g ref %.9g' % (float(soft[tuple(i)]), float(ref['soft_mask'][tuple(i)])))
    s2 = kal.render.mesh.dibr_soft_mask(fimg.cuda(), face_idx, 7000., boxlen, knum, 1000.)
    print('operator path equal to fused:', bool(torch.equal(s2, soft)), 'operator vs oracle max', float((s2.cpu() - ref['soft_mask']).abs().max()))
    for v in range(0):
        m = (d[v] > 1e-6)
        print('view', v, 'covered rows', (face_idx[v] >= 0).any(1).nonzero().flatten().tolist()[:1], '..', 'bad map:')
        for y in range(H):
            print(''.join('X' if m[y, x] else ('#' if face_idx[v, y, x] >= 0 else '.') for x in range(W)))
