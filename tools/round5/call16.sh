#!/bin/bash
# round 5, call 16: long runs of a face summed per workgroup in LDS (the soft mask's flat backward), contiguous shares of the rounds
set -u
repo=$(pwd); out=$repo/gpurun_out/r05r; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_full_size_parity.py tests/test_render_fused.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
q() { echo "== $* ${EXTRA:-}"; env "$@" timeout 200 python bench.py --quick --steps 50 ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'median', d['median_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"; }
{
for sc in "" "--scene knot"; do
EXTRA="$sc" q KAMD_X=product_min24
EXTRA="$sc" q KAMD_LIB_PATH=${L}_nobig.so
EXTRA="$sc" q KAMD_LIB_PATH=${L}_big8.so
EXTRA="$sc" q KAMD_LIB_PATH=${L}_big48.so
done
} > $out/bwd_big_runs_ab.txt 2>&1
cat $out/bwd_big_runs_ab.txt
