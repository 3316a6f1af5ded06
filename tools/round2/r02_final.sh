#!/bin/bash
# last call of the round: the full GPU suite on HEAD, then the bench line (no CPU baseline: the minutes left are few)
set -u
out=gpurun_out/r02zz; mkdir -p $out
timeout 200 python -m pytest tests -q -x -m gpu --timeout 120 > $out/pytest_gpu.log 2>&1; tail -1 $out/pytest_gpu.log
timeout 120 python bench.py --no-cpu-baseline 2> $out/bench.err | tail -1 > $out/bench.json
python - $out/bench.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print('  ms/step', j['ms_per_step'], j['per_step_ms'], 'roofline', j['roofline']['frac'], j['roofline']['avg_launch_us'])
    print('  ', {k.replace('_kernel', ''): v['avg_us'] for k, v in j['kernels'].items()})
    print('  chamfer', j['chamfer']['ms_per_step'], 'c5', j['c5']['voxelgrid_256_us'], j['c5']['point_to_mesh_1Mx50k_ms'])
except Exception as e:
    print('bench failed', e)
PY
