#!/bin/bash
# quick DIB-R check: tests of the fused path + the bench's DIB-R section
set -u
out=gpurun_out/${1:-r02q}; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -1 $out/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-chamfer --no-c5 2>> $out/bench.err | tail -1 > $out/bench.json
python - $out/bench.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print('  ms/step', j['ms_per_step'], j['per_step_ms'], 'fg', j['feature_grad_variant']['per_step_ms']['median'])
    print('  ', {k.replace('_kernel', ''): v['avg_us'] for k, v in j['kernels'].items()})
except Exception as e:
    print('bench failed', e)
PY
