"""Host-side helpers of bench.py that decide what the one JSON line reports (no GPU needed)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _k(share, gbps=100.0):
    return {'share_of_instrumented_step': share, 'algorithmic_GBps': gbps}


def test_dominant_kernel_is_the_largest_share_with_a_byte_count():
    kernels = {'raster_backward_kernel': _k(0.30), 'soft_search_kernel': _k(0.16), 'pv_forward_kernel': _k(0.5, None)}
    assert bench.pick_dominant(kernels) == 'raster_backward_kernel'
    assert bench.pick_dominant({'pv_forward_kernel': _k(0.5, None)}) is None
    assert bench.pick_dominant({}) is None


def test_near_ties_go_to_a_kernel_that_is_alone_on_the_stream():
    # the two backward kernels overlap (side stream): a kernel that runs alone wins a near tie
    kernels = {'raster_backward_kernel': _k(0.1567), 'soft_select_kernel': _k(0.1547), 'soft_eval_kernel': _k(0.07)}
    assert bench.pick_dominant(kernels) == 'soft_select_kernel'
    kernels['soft_select_kernel'] = _k(0.14)                     # more than 5 % behind: no tie any more
    assert bench.pick_dominant(kernels) == 'raster_backward_kernel'
    only_overlapped = {'soft_mask_backward_list_kernel': _k(0.2), 'raster_backward_kernel': _k(0.199)}
    assert bench.pick_dominant(only_overlapped) == 'soft_mask_backward_list_kernel'


def test_algorithmic_bytes_follow_the_survey_per_unit_figures():
    """SURVEY 8(d) per-unit figures x the units one launch processes: spot values at C4."""
    B, P, F, Fv = 8, 1024 * 1024, 50000, 25000
    assert bench.algorithmic_bytes('raster_tile_kernel', B, P, F, Fv, 3, 30) == B * (P * 32 + Fv * 88)   # P (20 + 4D) + F' (52 + 12D)
    assert bench.algorithmic_bytes('soft_select_kernel', B, P, F, Fv, 3, 30) == B * (P * 8 + F * 40)
    assert bench.algorithmic_bytes('soft_eval_kernel', B, P, F, Fv, 3, 30) is None
    assert bench.algorithmic_bytes('fill_regions_kernel', B, P, F, Fv, 3, 30) == B * P * 30 * 13
    assert bench.algorithmic_bytes('pv_forward_kernel', B, P, F, Fv, 3, 30) is None
    # the fused backward walks the covered tiles only: charged the pixels of those tiles, not of the image (VERDICT r03 weak #7b)
    p_cov = 0.2 * P
    assert bench.algorithmic_bytes('raster_backward_kernel', B, P, F, Fv, 3, 30, P_cov=p_cov) == B * (p_cov * 32 + Fv * 48)
    assert bench.algorithmic_bytes('raster_backward_kernel', B, P, F, Fv, 3, 30) == B * (P * 32 + Fv * 48)


def test_traffic_parser_knows_the_list_walking_backward():
    """VERDICT r03 weak #7b: the PMC parser's pattern for the rasterizer's backward did not match raster_backward_list_kernel."""
    import re
    src = open(os.path.join(ROOT, 'tools', 'parse_traffic.py')).read()
    pat = re.search(r"'raster_backward_kernel': r'([^']+)'", src).group(1)
    assert re.search(pat, 'void (anonymous namespace)::raster_backward_list_kernel<float, 3, false>(int, int)')
    assert re.search(pat, 'void (anonymous namespace)::raster_backward_kernel<float, 3, false>(int, int)')


def test_cpu_reference_extras_small_sample():
    """The opt-in CPU readings (bench.py --cpu-extras): the C oracle and the dense torch oracle of chamfer agree to rounding,
    the torch oracle of the rasterizer picks the faces the restated kernel picks."""
    r = bench.cpu_reference_extras(n_points=1500, slice_rows=300, raster_res=8, sphere_frequency=4)
    assert r['chamfer_restated_kernel']['value'] > 0 and r['chamfer_restated_kernel']['unit'] == 'Mpoint-pairs/s'
    assert r['chamfer_torch_oracle']['max_rel_diff_vs_restated_kernel'] < 1e-5
    assert r['rasterize_torch_oracle']['face_idx_equals_restated_kernel'] is True
    assert all(v['kind'] == 'port' and v['cores'] >= 1 for v in r.values())


def test_knot_shuffled_scene_is_a_permutation_of_the_knot():
    """bench.py --scene knot_shuffled / scene_variants.knot_shuffled and the full-size parity test render the knot's faces in a
    fixed random order: the same triangles (as a multiset of vertex-index triples), the same vertices, another list order."""
    import torch
    from kaolin_amd.utils import testing as T
    v0, f0 = T.scene_mesh('knot')
    v1, f1 = T.scene_mesh('knot_shuffled')
    assert torch.equal(v0, v1) and f0.shape == f1.shape and not torch.equal(f0, f1)
    key = lambda f: torch.sort((f[:, 0] * v0.shape[0] + f[:, 1]) * v0.shape[0] + f[:, 2]).values
    assert torch.equal(key(f0), key(f1))
    _, f2 = T.scene_mesh('knot_shuffled')
    assert torch.equal(f1, f2)  # (a fixed seed: every run renders the same order)
