#!/bin/bash
# round 3, call 6: covered-row span published at the end of the binning kernel (one copy, read before update); raster_tile's tail reordered
set -u
out=gpurun_out/r03f; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q --durations=5 --timeout 280 > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.log | head -20
L=$(pwd)/kaolin_amd
{
bash tools/round3/ab.sh base
bash tools/round3/ab.sh row_centre_off KAMD_ROW_ORDER=2
bash tools/round3/ab.sh fixed_middle_no_lookup KAMD_LIB_PATH=$L/libkaolin_amd_order3.so
bash tools/round3/ab.sh old_tail_order KAMD_LIB_PATH=$L/libkaolin_amd_tail0.so
bash tools/round3/ab.sh top_of_image -- --look-at 0 -0.62 0
bash tools/round3/ab.sh top_of_image_row_centre_off KAMD_ROW_ORDER=2 -- --look-at 0 -0.62 0
bash tools/round3/ab.sh base_again
bash tools/round3/ab.sh fixed_middle_no_lookup_again KAMD_LIB_PATH=$L/libkaolin_amd_order3.so
bash tools/round3/ab.sh old_tail_order_again KAMD_LIB_PATH=$L/libkaolin_amd_tail0.so
} 2>&1 | tee $out/ab.txt
