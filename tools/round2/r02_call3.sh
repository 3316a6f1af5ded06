#!/bin/bash
# Round 2, GPU call 3: first contact of the persistent grid build / fused chamfer, the two-launch voxelizer and the slab
# bounds of point_to_mesh: a guarded sanity run first (a grid barrier that never completes must not eat the call), then
# the tests of those operators, the rest of the suite, and the bench.  Output -> gpurun_out/r02g/.
set -u
out=gpurun_out/r02g; mkdir -p $out
timeout 180 python tools/check_chamfer.py > $out/check_chamfer.txt 2>&1; echo "check_chamfer rc=$?"; tail -12 $out/check_chamfer.txt
if ! grep -q "CHAMFER OK" $out/check_chamfer.txt; then echo "chamfer sanity failed: stopping"; exit 0; fi
timeout 900 python -m pytest tests/test_sided_distance.py tests/test_graph_capture.py tests/test_voxelgrid.py tests/test_triangle_distance.py tests/test_render_fused.py tests/test_prepare_vertices.py -q -x -m gpu --timeout 300 > $out/pytest_new.log 2>&1; tail -3 $out/pytest_new.log
timeout 1200 python -m pytest tests -q -x -m gpu --timeout 600 --deselect tests/test_sided_distance.py --deselect tests/test_graph_capture.py --deselect tests/test_voxelgrid.py --deselect tests/test_triangle_distance.py --deselect tests/test_render_fused.py --deselect tests/test_prepare_vertices.py > $out/pytest_rest.log 2>&1; tail -3 $out/pytest_rest.log
timeout 600 python bench.py 2> $out/bench.err | tail -1 > $out/bench.json
python - <<'PY'
import json
try:
    j = json.load(open('gpurun_out/r02g/bench.json'))
    print('dibr ms/step', j['ms_per_step'], j['per_step_ms'])
    print('chamfer', j['chamfer']['ms_per_step'], j['chamfer']['host_enqueue_ms_per_step'], j['chamfer']['kernels_avg_us'])
    print('c5', {k: v for k, v in j['c5'].items() if 'deftet' not in k})
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r02g/bench.err').read()[-2000:])
PY
