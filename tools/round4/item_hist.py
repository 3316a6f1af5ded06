"""How many (pixel, face) pairs do the soft mask's work items hold?  Histogram of the pair counts the select kernel leaves in the item
records (4th word), per scene."""
import sys, torch
sys.path.insert(0, '.')
import kaolin_amd as kal
from kaolin_amd._C.render import mesh as M
from kaolin_amd.utils import testing as T
H = W = 1024; V = 8
for name in ('sphere', 'knot'):
    fz, fimg, feats, nz = (T.knot_scene(num_views=V, device='cuda') if name == 'knot' else T.sphere_scene(level=50, num_views=V, device='cuda'))
    feat = torch.cat(feats, -1).contiguous()
    interp, face_idx, wts, soft, hits, _ = M.dibr_rasterization_forward_fused(H, W, fz, fimg, feat, nz, 7000., 0.02, 30, 1000., 1e-8)
    torch.cuda.synchronize()
    work = hits[4]
    n_groups = V * (H // 16) * (W // 16)
    shard_cap = 4 * ((n_groups + 7) // 8)
    counts = work[0:8 * M.COUNTER_STRIDE:M.COUNTER_STRIDE].tolist()
    items = work[M.WORK_HEADER:M.WORK_HEADER + 8 * shard_cap * 4].view(8, shard_cap, 4)
    pairs = torch.cat([items[s, :min(c, shard_cap), 3] for s, c in enumerate(counts)]).long().cpu()
    n = pairs.numel()
    edges = [0, 1, 17, 33, 65, 129, 257, 513, 1025, 1921]
    print(f'{name}: {n} items, {int(pairs.sum())} pairs; items by pair count:',
          ' '.join(f'[{a},{b}): {int(((pairs >= a) & (pairs < b)).sum())} ({int(pairs[(pairs >= a) & (pairs < b)].sum())} pairs)' for a, b in zip(edges[:-1], edges[1:])))
