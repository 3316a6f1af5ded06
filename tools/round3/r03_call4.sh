#!/bin/bash
# round 3, call 4: after removing the per-workgroup L2 write-back (__threadfence) and giving every append counter its own line
set -u
out=gpurun_out/r03d; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q --durations=5 --timeout 280 > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.log | head -20
L=$(pwd)/kaolin_amd
{
bash tools/round3/ab.sh base
bash tools/round3/ab.sh row_order_off KAMD_ROW_ORDER=2
bash tools/round3/ab.sh tile_counters_one_per_line KAMD_LIB_PATH=$L/libkaolin_amd_cs32.so
bash tools/round3/ab.sh top_of_image_row_order_on -- --look-at 0 -0.62 0
bash tools/round3/ab.sh top_of_image_row_order_off KAMD_ROW_ORDER=2 -- --look-at 0 -0.62 0
bash tools/round3/ab.sh base_again
bash tools/round3/ab.sh tile_counters_one_per_line_again KAMD_LIB_PATH=$L/libkaolin_amd_cs32.so
AB_LABEL="chamfer_default" timeout 120 python tools/round3/chamfer_ab.py 2>&1 | tail -1
} 2>&1 | tee $out/ab.txt
