"""The rasterizer kernels' workgroup -> tile map (kaolin_amd/csrc/raster2.inc KAMD_RASTER_ORDER, rasterize.hip
KAMD_RBWD_ORDER): views interleaved, a view's tile rows heaviest first -- the order the binning launch leaves
(tile_lists.h sort_tile_rows: rows ranked by the faces their tiles list, ties by row) -- or, without it, from the middle of
the image outwards.  Restated in Python and checked for what correctness needs -- every (view, tile) exactly once -- and
for what the order is for: heavy rows first whatever the object's position.  The GPU test reads the order a forward pass
left behind."""
import pytest
import torch


def rank_rows(row_work):
    """sort_tile_rows for one view: order[rank] = row, rank = rows with more work (or as much and a smaller index)."""
    n = len(row_work)
    order = [None] * n
    for r in range(n):
        rank = sum(1 for q in range(n) if row_work[q] > row_work[r] or (row_work[q] == row_work[r] and q < r))
        order[rank] = r
    return order


def tile_of(block, B, tiles_x, tiles_y, columns_too=False, row_order=None):
    b, k = block % B, block // B
    kr, tx = k // tiles_x, k % tiles_x
    mid = tiles_y >> 1
    ty = mid - ((kr + 1) >> 1) if kr & 1 else mid + (kr >> 1)
    if row_order is not None:
        ty = row_order[b][kr]
    if columns_too:
        midx = tiles_x >> 1
        tx = midx - ((tx + 1) >> 1) if tx & 1 else midx + (tx >> 1)
    return b, ty * tiles_x + tx


@pytest.mark.parametrize('columns_too', [False, True])
@pytest.mark.parametrize('B', [1, 3, 8])
@pytest.mark.parametrize('tiles_x,tiles_y', [(1, 1), (1, 2), (3, 1), (2, 3), (5, 4), (7, 7), (64, 64), (12, 9), (3, 16)])
def test_every_tile_of_every_view_exactly_once(B, tiles_x, tiles_y, columns_too):
    n = tiles_x * tiles_y
    seen = set()
    for block in range(B * n):
        b, tile = tile_of(block, B, tiles_x, tiles_y, columns_too)
        assert 0 <= b < B and 0 <= tile < n
        seen.add((b, tile))
    assert len(seen) == B * n


@pytest.mark.parametrize('tiles_y', [1, 2, 5, 8, 64, 65])
def test_rows_leave_the_middle_monotonically(tiles_y):
    tiles_x, B = 4, 2
    rows = [tile_of(block, B, tiles_x, tiles_y)[1] // tiles_x for block in range(0, B * tiles_x * tiles_y, B * tiles_x)]
    assert sorted(rows) == list(range(tiles_y))
    dist = [abs(r - (tiles_y >> 1)) for r in rows]
    assert dist == sorted(dist) and rows[0] == tiles_y >> 1
    # consecutive workgroups are the B views of one tile (a view stays on one XCD when B % 8 == 0), then the row's next tile
    first = [tile_of(block, B, tiles_x, tiles_y) for block in range(B * 2)]
    assert [b for b, _ in first] == [0, 1, 0, 1] and first[0][1] == first[1][1] and first[2][1] == first[0][1] + 1


@pytest.mark.parametrize('seed', [0, 1, 2])
@pytest.mark.parametrize('B,tiles_x,tiles_y', [(1, 3, 1), (2, 4, 7), (8, 64, 64), (3, 5, 256)])
def test_data_driven_row_order_is_a_bijection_heaviest_first(B, tiles_x, tiles_y, seed):
    g = torch.Generator().manual_seed(seed)
    work = torch.randint(0, 5, (B, tiles_y), generator=g) * torch.randint(0, 2, (B, tiles_y), generator=g)   # many ties and zeros
    order = [rank_rows(work[b].tolist()) for b in range(B)]
    for b in range(B):
        assert sorted(order[b]) == list(range(tiles_y))
        w = [int(work[b, r]) for r in order[b]]
        assert w == sorted(w, reverse=True)
        assert all(order[b][i] < order[b][i + 1] for i in range(tiles_y - 1) if w[i] == w[i + 1])    # ties keep the row order
    n = tiles_x * tiles_y
    seen = {tile_of(block, B, tiles_x, tiles_y, row_order=order) for block in range(B * n)}
    assert len(seen) == B * n


@pytest.mark.gpu
@pytest.mark.parametrize('shift', [(0., 0.), (0.5, -0.55), (-0.3, 0.9)])
def test_forward_leaves_a_row_order_that_starts_where_the_object_is(shift):
    """dibr_rasterization's binning launch ranks every view's 16-pixel tile rows by the faces their tiles list and leaves
    the order in the operator's work buffer (what raster_tile / raster_backward follow): a permutation per view, and every
    row that holds a covered pixel comes before every row no face box reaches -- wherever the object sits."""
    import kaolin_amd as kal
    from kaolin_amd.utils import testing as T
    fz, fimg, feats, nz = T.sphere_scene(level=12, num_views=3, device='cuda')
    fimg = (fimg + torch.tensor(shift, device='cuda')).contiguous()
    H, W = 272, 200
    feat = torch.cat(feats, -1).contiguous()
    a = fimg.clone().requires_grad_()
    out, soft, face_idx = kal.render.mesh.dibr_rasterization(H, W, fz, a, feat, nz)
    work = out.grad_fn.saved_tensors[-1]
    B, tiles_x, tiles_y = 3, (W + 15) // 16, (H + 15) // 16
    lib = kal._lib.load()
    assert work.numel() == lib.kamd_dibr_soft_mask_work_words(B, H, W)
    n_groups = B * tiles_x * tiles_y
    off = kal._C.render.mesh.WORK_HEADER + 8 * (4 * ((n_groups + 7) // 8)) * 4 + (n_groups + 3) // 4         # header, items, coverage bytes (tile_lists.h)
    order = work[off:off + (B * tiles_y + 1) // 2].view(torch.int16)[:B * tiles_y].view(B, tiles_y).long().cpu()
    covered_rows = (face_idx >= 0).any(dim=2).cpu()                                  # (B, H)
    for b in range(B):
        assert sorted(order[b].tolist()) == list(range(tiles_y))
        has_cov = [bool(covered_rows[b, r * 16:(r + 1) * 16].any()) for r in range(tiles_y)]
        n_cov = sum(has_cov)
        if n_cov == 0:
            continue
        # the scaled boxes of the faces: rows no box comes near have no work and must all come after the covered rows
        first = set(order[b, :n_cov].tolist())
        ys = fimg[b, :, :, 1]
        lo, hi = float(ys.min()), float(ys.max())
        reach = [r for r in range(tiles_y) if not ((H - 1 - (hi + 0.05) * H) / 2 > (r + 1) * 16 or (H - 1 - (lo - 0.05) * H) / 2 < r * 16 - 1)]
        assert all(has_cov[r] or r in reach for r in first)
        assert all(order[b, i] in reach for i in range(n_cov)), 'a row without any face ranked among the heaviest'
        pos = {int(r): i for i, r in enumerate(order[b].tolist())}
        unreached = [r for r in range(tiles_y) if r not in reach]
        assert all(pos[c] < pos[u] for c in range(tiles_y) if has_cov[c] for u in unreached)
    (out.sum() + soft.sum()).backward()          # the backward follows the same order: must simply work
    assert torch.isfinite(a.grad).all()
