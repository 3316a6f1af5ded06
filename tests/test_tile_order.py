"""The rasterizer kernels' workgroup -> tile map (kaolin_amd/csrc/raster2.inc KAMD_RASTER_ORDER, rasterize.hip
KAMD_RBWD_ORDER): views interleaved, a view's tile rows visited outwards from a centre row -- the middle of the rows the
mesh's boxes cover, which the binning launch leaves per view (tile_lists.h note_row_span / row_centre / row_from_centre), or
the middle of the image.  Restated in Python and checked for what correctness needs -- every (view, tile) exactly once --
and for what the order is for: the rows come in non-decreasing distance from the centre until one side of the image is
used up.  The GPU test reads the centre a forward pass left behind."""
import pytest
import torch


def row_from_centre(k, c, tiles_y):
    """tile_lists.h row_from_centre: c, c - 1, c + 1, c - 2, ... and, once one side is used up, on along the other."""
    left, right = c, tiles_y - 1 - c
    m = min(left, right)
    if k <= 2 * m:
        return c - ((k + 1) >> 1) if k & 1 else c + (k >> 1)
    return k if right > left else tiles_y - 1 - k


def tile_of(block, B, tiles_x, tiles_y, columns_too=False, centres=None):
    b, k = block % B, block // B
    kr, tx = k // tiles_x, k % tiles_x
    ty = row_from_centre(kr, tiles_y >> 1 if centres is None else centres[b], tiles_y)
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


@pytest.mark.parametrize('tiles_y', [1, 2, 3, 8, 63, 64])
def test_rows_from_any_centre_are_a_bijection_in_order_of_distance(tiles_y):
    for c in range(tiles_y):
        rows = [row_from_centre(k, c, tiles_y) for k in range(tiles_y)]
        assert sorted(rows) == list(range(tiles_y)) and rows[0] == c
        dist = [abs(r - c) for r in rows]
        assert dist == sorted(dist)
    B, tiles_x = 3, 5
    centres = [0, tiles_y - 1, tiles_y // 3]
    seen = {tile_of(block, B, tiles_x, tiles_y, centres=centres) for block in range(B * tiles_x * tiles_y)}
    assert len(seen) == B * tiles_x * tiles_y


@pytest.mark.gpu
@pytest.mark.parametrize('shift', [(0., 0.), (0.5, -0.55), (-0.3, 0.9)])
def test_forward_starts_where_the_object_is(shift):
    """dibr_rasterization's binning launch notes, per view, the tile rows the kept faces' boxes cover; the tile kernel starts
    from the middle of them and leaves that row in the operator's work buffer for the backward pass.  Wherever the object
    sits, the centre row must lie inside the rows that hold covered pixels (give or take the boxes' slack)."""
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
    off = kal._C.render.mesh.WORK_HEADER + 8 * (4 * ((n_groups + 7) // 8)) * 4 + (n_groups + 3) // 4   # header, items, coverage bytes
    centres = work[off:off + B].tolist()
    covered_rows = (face_idx >= 0).any(dim=2).cpu()                                  # (B, H)
    for b in range(B):
        assert 0 <= centres[b] < tiles_y
        rows = [r for r in range(tiles_y) if bool(covered_rows[b, r * 16:(r + 1) * 16].any())]
        if rows:
            assert rows[0] - 1 <= centres[b] <= rows[-1] + 1, (centres[b], rows)
            assert abs(centres[b] - (rows[0] + rows[-1]) / 2) <= 1.5
    (out.sum() + soft.sum()).backward()          # the backward starts from the same rows: must simply work
    assert torch.isfinite(a.grad).all()
