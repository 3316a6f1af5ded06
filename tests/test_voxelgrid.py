"""trianglemeshes_to_voxelgrids.

CPU part: pins oracle/voxelgrid.py (numpy restatement) against golden outputs of the reference's own torch
implementation (tests/golden/make_golden.py: its test inputs + seeded random / sphere meshes), torch.equal
as in tests/python/kaolin/ops/conversions/test_trianglemesh.py:45-242, and the reference's docstring KAT.
GPU part: the HIP kernel through the C ABI vs goldens and vs the oracle, bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import voxelgrid as vox_oracle
from conftest import GOLDEN_DIR

CASES = ['batched', 'origins', 'scale', 'res4', 'rect', 'rand', 'rand_out', 'sphere64', 'sphere40_f64']


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLDEN_DIR, 'voxelgrid.npz'))


def load_case(gold, name):
    v, f, res = torch.from_numpy(gold[name + '_vertices']), torch.from_numpy(gold[name + '_faces']), int(gold[name + '_res'])
    org = torch.from_numpy(gold[name + '_origin']) if name + '_origin' in gold else None
    sc = torch.from_numpy(gold[name + '_scale']) if name + '_scale' in gold else None
    n = v.shape[0] * res ** 3
    dense = np.unpackbits(gold[name + '_packed'])[:n].reshape(v.shape[0], res, res, res)
    return v, f, res, org, sc, torch.from_numpy(dense).to(v.dtype)


@pytest.mark.parametrize('name', CASES)
def test_oracle_vs_reference_golden(gold, name):
    v, f, res, org, sc, expected = load_case(gold, name)
    out = vox_oracle.trianglemeshes_to_voxelgrids(v, f, res, org, sc)
    assert out.dtype == v.dtype and torch.equal(out, expected)


def test_oracle_docstring_kat():
    """kaolin/ops/conversions/trianglemesh.py:65-82."""
    v = torch.tensor([[[0, 0, 0], [1, 0, 0], [0, 0, 1]]], dtype=torch.float)
    out = vox_oracle.trianglemeshes_to_voxelgrids(v, torch.tensor([[0, 1, 2]]), 3, torch.zeros(1, 3), torch.ones(1))
    exp = torch.tensor([[[[1., 1., 1.], [0., 0., 0.], [0., 0., 0.]], [[1., 1., 0.], [0., 0., 0.], [0., 0., 0.]],
                         [[1., 0., 0.], [0., 0., 0.], [0., 0., 0.]]]])
    assert torch.equal(out, exp)
    with pytest.raises(TypeError, match=r'Expected resolution to be int but got .*'):
        vox_oracle.trianglemeshes_to_voxelgrids(v, torch.tensor([[0, 1, 2]]), 2.3)


# ------------------------------------------------------------------ GPU
def _conv():
    from kaolin_amd.ops import conversions
    return conversions


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('return_sparse', [False, True])
def test_gpu_vs_reference_golden(gold, name, return_sparse):
    v, f, res, org, sc, expected = load_case(gold, name)
    args = [None if org is None else org.cuda(), None if sc is None else sc.cuda()]
    out = _conv().trianglemeshes_to_voxelgrids(v.cuda(), f.cuda(), res, *args, return_sparse=return_sparse)
    if return_sparse:
        assert out.is_sparse
        out = out.to_dense()
    assert out.dtype == v.dtype and out.shape == expected.shape and torch.equal(out.cpu(), expected)


@pytest.mark.gpu
def test_gpu_sparse_result_never_allocates_the_dense_grid():
    """return_sparse=True builds the COO tensor from the marked voxels (a bit grid, compacted) -- the reference builds it from the
    unique voxel indices and never holds R^3 scalars (kaolin/ops/conversions/pointcloud.py:66-73); round 5 returned
    dense.to_sparse(): 4 GB at R = 1024.  Same indices, same order, and the call's peak memory stays far below one dense grid."""
    from kaolin_amd.utils.testing import geodesic_sphere
    v, f = geodesic_sphere(16)
    v = (v[None] * torch.tensor([1.0, 0.8, 1.2])).repeat(2, 1, 1)
    v[1] *= 0.5
    res = 384
    vc, fc = v.cuda(), f.cuda()
    dense = _conv().trianglemeshes_to_voxelgrids(vc, fc, res)
    want = dense.to_sparse()
    del dense
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    got = _conv().trianglemeshes_to_voxelgrids(vc, fc, res, return_sparse=True)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert got.is_sparse and got.is_coalesced() and got.shape == want.shape and got.dtype == want.dtype
    assert torch.equal(got.indices(), want.indices()) and torch.equal(got.values(), want.values())
    assert peak < 0.25 * 2 * res ** 3 * 4, peak        # (two dense grids would be 453 MB; the bit grids are 14 MB)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float, torch.double])
@pytest.mark.parametrize('level,res', [(8, 128), (2, 200), (16, 96)])
def test_gpu_vs_oracle_spheres(dtype, level, res):
    from kaolin_amd.utils.testing import geodesic_sphere
    v, f = geodesic_sphere(level)
    v = v.to(dtype)[None] * torch.tensor([1.0, 0.7, 1.3], dtype=dtype)
    out = _conv().trianglemeshes_to_voxelgrids(v.cuda(), f.cuda(), res)
    assert torch.equal(out.cpu(), vox_oracle.trianglemeshes_to_voxelgrids(v, f, res))


@pytest.mark.gpu
def test_gpu_edge_cases():
    conv = _conv().trianglemeshes_to_voxelgrids
    v = torch.tensor([[[0, 0, 0], [1, 0, 0], [0, 0, 1]]], dtype=torch.float, device='cuda')
    f = torch.tensor([[0, 1, 2]], device='cuda')
    with pytest.raises(TypeError, match=r'Expected resolution to be int but got .*'):
        conv(v, f, 2.3)
    # a vertex no face references still lands in the grid; vertices outside [0,1] are dropped
    v2 = torch.tensor([[[0, 0, 0], [1, 0, 0], [0, 0, 1], [0.5, 1.0, 0.5], [3., 3., 3.]]], dtype=torch.float)
    out = conv(v2.cuda(), f, 8, torch.zeros(1, 3).cuda(), torch.ones(1).cuda())
    assert torch.equal(out.cpu(), vox_oracle.trianglemeshes_to_voxelgrids(v2, f.cpu(), 8, torch.zeros(1, 3), torch.ones(1)))
    # one huge triangle: deep recursion from a single face (load balancing path)
    big = torch.tensor([[[0, 0, 0], [1, 0.2, 0], [0.1, 1, 1]]], dtype=torch.float)
    out = conv(big.cuda(), f, 256)
    assert torch.equal(out.cpu(), vox_oracle.trianglemeshes_to_voxelgrids(big, f.cpu(), 256))


@pytest.mark.gpu
def test_gpu_full_size_c5():
    """C5: 50k-face sphere at 256^3, bit-exact against the oracle (it finishes in seconds)."""
    from kaolin_amd.utils.testing import geodesic_sphere
    v, f = geodesic_sphere(50)
    v = v.float()[None]
    out = _conv().trianglemeshes_to_voxelgrids(v.cuda(), f.cuda(), 256)
    ref = vox_oracle.trianglemeshes_to_voxelgrids(v, f, 256)
    assert out.shape == (1, 256, 256, 256) and torch.equal(out.cpu(), ref)
    assert 100000 < int(ref.sum()) < 400000


@pytest.mark.gpu
def test_gpu_1000_repetitions_on_dirty_memory():
    """1 000 back-to-back calls over shapes with one and several meshes, the output landing on dirty memory every time (the
    allocator hands the previous result back, overwritten with 0.5), every seventh call on a second stream: a voxel the clear
    missed, a mark lost to the clear or an extent read before it was reduced would show up as a difference from the oracle."""
    from kaolin_amd.utils.testing import geodesic_sphere
    conv = _conv().trianglemeshes_to_voxelgrids
    cases = []
    for level, res, batch, dtype in ((50, 256, 1, torch.float), (6, 128, 3, torch.float), (2, 96, 2, torch.double), (10, 64, 5, torch.float)):
        v, f = geodesic_sphere(level)
        v = torch.stack([v.to(dtype) * torch.tensor([1.0, 0.6 + 0.1 * b, 1.2], dtype=dtype) for b in range(batch)])
        cases.append((v.cuda(), f.cuda(), res, vox_oracle.trianglemeshes_to_voxelgrids(v, f, res).cuda()))
    side = torch.cuda.Stream()
    bad = 0
    for rep in range(1000):
        v, f, res, expected = cases[rep % len(cases)]
        if rep % 7 == 3:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = conv(v, f, res)
            torch.cuda.current_stream().wait_stream(side)
        else:
            out = conv(v, f, res)
        out.record_stream(torch.cuda.current_stream())
        bad += int(not torch.equal(out, expected))
        out.fill_(0.5)       # the next result lands on dirty memory
        del out
    assert bad == 0
