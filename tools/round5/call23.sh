#!/bin/bash
# round 5, call 23: the soft mask's backward with fewer vector instructions (two selects per slot, the row merge as fused multiply-adds,
# two rounds per trip of the pipeline, no zero fills of a round's unused fields)
set -u
repo=$(pwd); out=$repo/gpurun_out/r05z; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
{
timeout 600 python tools/round4/fuzz_dibr.py 150 3000 2>&1 | tail -3
timeout 600 python tools/round4/fuzz_soft_mask.py 100 3000 2>&1 | tail -3
timeout 600 python tools/round4/fuzz_dibr_nonfinite.py 60 3000 2>&1 | tail -3
} | grep -v amdgpu.ids > $out/fuzz.txt; cat $out/fuzz.txt
q() { echo "== $* ${EXTRA:-}"; env "$@" timeout 200 python bench.py --quick --steps 50 ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'median', d['median_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"; }
{
for i in 1 2; do
for sc in "" "--scene knot"; do
EXTRA="$sc" q KAMD_X=product_soft_backward_diet
EXTRA="$sc" q KAMD_LIB_PATH=${L}_base.so
done
done
} > $out/soft_backward_diet_ab.txt 2>&1
cat $out/soft_backward_diet_ab.txt
