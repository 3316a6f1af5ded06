"""The rasterizer kernels' workgroup -> tile map (kaolin_amd/csrc/raster2.inc KAMD_RASTER_ORDER, rasterize.hip
KAMD_RBWD_ORDER): tile rows are visited from the middle of the image outwards, views interleaved.  Restated in Python and
checked for what correctness needs -- every (view, tile) exactly once -- and for what the order is for: the rows come in
non-decreasing distance from the middle row.  (No GPU; the GPU parity tests run the kernels with this map.)"""
import pytest


def tile_of(block, B, tiles_x, tiles_y, columns_too=False):
    b, k = block % B, block // B
    kr, tx = k // tiles_x, k % tiles_x
    mid = tiles_y >> 1
    ty = mid - ((kr + 1) >> 1) if kr & 1 else mid + (kr >> 1)
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
