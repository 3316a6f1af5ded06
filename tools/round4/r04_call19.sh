#!/bin/bash
# round 4, call 19: the binning changes (medium faces a lane per face, rasterizer big list from 16 tiles, pool x8, 512 entries per tile) --
# parity suite of the renderer, then the full bench line with per-scene work units
set -u
out=gpurun_out/r04c19; mkdir -p $out
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_tile_order.py tests/test_full_size_parity.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -2 | tee $out/pytest.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r04c19/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j.get('median_ms_per_step'), j['work_units_per_view'])
k = j['scene_variants']['knot']
for key in ('ms_per_step', 'work_units_per_view', 'vs_headline_scene_work_ratio', 'vs_headline_scene_kernel_ratio', 'kernel_ratio_over_work_ratio'):
    print(key, k.get(key))
PY
timeout 200 python tools/round4/knot_parts.py 2>&1 | grep -E "shuffled|^all" | tee $out/knot_parts.txt | cut -c1-260
