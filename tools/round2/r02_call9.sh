#!/bin/bash
# Round 2, GPU call 9: pipelined soft-mask backward / eval (DIB-R tests + bench A/B), chamfer value from partials.
set -u
out=gpurun_out/r02m; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_sided_distance.py tests/test_graph_capture.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for m in 1 2; do
  echo "KAMD_EVAL_PIPE=$m"
  KAMD_EVAL_PIPE=$m timeout 300 python bench.py --no-cpu-baseline --no-chamfer --no-c5 2>> $out/bench.err | tail -1 > $out/bench_pipe$m.json
  python - $out/bench_pipe$m.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print('  ms/step', j['ms_per_step'], j['per_step_ms'], 'fg', j['feature_grad_variant']['per_step_ms']['median'])
    print('  ', {k.replace('_kernel', ''): v['avg_us'] for k, v in j['kernels'].items()})
except Exception as e:
    print('bench failed', e)
PY
done
tail -3 $out/bench.err
KAMD_CHECK_SPLIT=1 timeout 180 python tools/check_chamfer.py 2>&1 | grep "forward\|plain\|step\|OK\|rror" | tee $out/chamfer.txt
