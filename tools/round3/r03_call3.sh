#!/bin/bash
# round 3, call 3: GPU suite after the same-address-atomic fixes (sharded flat list, hierarchical ticket, row work summed per workgroup),
# DIB-R A/B lines (row order on / off, centred / off-centre), chamfer query variants
set -u
out=gpurun_out/r03c; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q --durations=5 --timeout 280 > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.log | head -20
L=$(pwd)/kaolin_amd
{
bash tools/round3/ab.sh base
bash tools/round3/ab.sh row_order_off KAMD_ROW_ORDER=2
bash tools/round3/ab.sh top_of_image_row_order_on -- --look-at 0 -0.62 0
bash tools/round3/ab.sh top_of_image_row_order_off KAMD_ROW_ORDER=2 -- --look-at 0 -0.62 0
bash tools/round3/ab.sh base_again
for n in 8 24; do bash tools/round3/ab.sh bwd_per_cu_$n KAMD_SOFT_BWD_PER_CU=$n; done
bash tools/round3/ab.sh eval_per_cu_16 KAMD_SOFT_EVAL_PER_CU=16
bash tools/round3/ab.sh select_per_cu_24 KAMD_SOFT_SELECT_PER_CU=24
for v in "" _sdgold _sdgr1 _sdgb2; do AB_LABEL="chamfer${v:-_default(r0=1,batch=3)}" KAMD_LIB_PATH=$L/libkaolin_amd$v.so timeout 120 python tools/round3/chamfer_ab.py 2>&1 | tail -1; done
} 2>&1 | tee $out/ab.txt
