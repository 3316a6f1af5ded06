#!/bin/bash
# round 5, call 31: soft_eval takes the places from the end when the launch is only a few generations long (product) vs always forwards (_evalfwd)
set -u
repo=$(pwd); out=$repo/gpurun_out/r05ah; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -2 $out/pytest_dibr.log
q() { echo "== $* ${EXTRA:-}"; env "$@" timeout 200 python bench.py --quick --steps 40 ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'median', d['median_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"; }
{
for i in 1 2; do
for sc in "" "--scene knot" "--scene knot_shuffled"; do
EXTRA="$sc" q KAMD_X=product_eval_order
EXTRA="$sc" q KAMD_LIB_PATH=${L}_evalfwd.so
done
done
} > $out/eval_order_ab.txt 2>&1
cat $out/eval_order_ab.txt
