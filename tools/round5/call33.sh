#!/bin/bash
# round 5, call 33: soft_eval walks the items heaviest class first (dealt by pair count in the rounds launch) vs in the worklist's order (_base)
set -u
repo=$(pwd); out=$repo/gpurun_out/r05ai; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 900 python -m pytest tests/test_full_size_parity.py tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
{
timeout 600 python tools/round4/fuzz_dibr.py 100 12000 2>&1 | tail -3
timeout 600 python tools/round4/fuzz_soft_mask.py 100 12000 2>&1 | tail -3
} | grep -v amdgpu.ids > $out/fuzz.txt; cat $out/fuzz.txt
q() { echo "== $* ${EXTRA:-}"; env "$@" timeout 200 python bench.py --quick --steps 40 ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'median', d['median_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"; }
{
for i in 1 2; do
for sc in "" "--scene knot" "--scene knot_shuffled"; do
EXTRA="$sc" q KAMD_X=product_eval_heaviest_first
EXTRA="$sc" q KAMD_LIB_PATH=${L}_base.so
done
done
} > $out/eval_lpt_ab.txt 2>&1
cat $out/eval_lpt_ab.txt
