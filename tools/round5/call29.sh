#!/bin/bash
# round 5, call 29: stress of soft_select's rounds launch -- a build with 64 ordered slots instead of 192 sends every tile of more than 64
# entries through it: the DIB-R suite and the sweeps must still be bit-identical
set -u
repo=$(pwd); out=$repo/gpurun_out/r05af; mkdir -p $out
export KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd_ecap64.so
timeout 900 python -m pytest tests/test_full_size_parity.py tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_render_fused.py -m gpu -q --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
{
timeout 600 python tools/round4/fuzz_dibr.py 150 9000 2>&1 | tail -3
timeout 600 python tools/round4/fuzz_soft_mask.py 150 9000 2>&1 | tail -3
timeout 600 python tools/round4/fuzz_dibr_nonfinite.py 60 9000 2>&1 | tail -3
timeout 200 python bench.py --quick --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C4 with 64 slots: ms_per_step', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
} | grep -v amdgpu.ids > $out/fuzz.txt; cat $out/fuzz.txt
