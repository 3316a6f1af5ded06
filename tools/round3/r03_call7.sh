#!/bin/bash
# round 3, call 7: sampled row-span report + cyclic shift of the tile-row order; raster_tile tail order A/B; fp64 grid tests
set -u
out=gpurun_out/r03g; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q --durations=5 --timeout 280 > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.log | head -20
L=$(pwd)/kaolin_amd
{
bash tools/round3/ab.sh base_shift_tail1
bash tools/round3/ab.sh shift_tail0 KAMD_LIB_PATH=$L/libkaolin_amd_tail0.so
bash tools/round3/ab.sh fixed_middle_tail1 KAMD_LIB_PATH=$L/libkaolin_amd_order3.so
bash tools/round3/ab.sh fixed_middle_tail0 KAMD_LIB_PATH=$L/libkaolin_amd_order3tail0.so
bash tools/round3/ab.sh span_off_env KAMD_ROW_ORDER=2
bash tools/round3/ab.sh top_of_image -- --look-at 0 -0.62 0
bash tools/round3/ab.sh top_of_image_tail0 KAMD_LIB_PATH=$L/libkaolin_amd_tail0.so -- --look-at 0 -0.62 0
bash tools/round3/ab.sh top_of_image_fixed_middle_tail0 KAMD_LIB_PATH=$L/libkaolin_amd_order3tail0.so -- --look-at 0 -0.62 0
bash tools/round3/ab.sh base_shift_tail1_again
bash tools/round3/ab.sh shift_tail0_again KAMD_LIB_PATH=$L/libkaolin_amd_tail0.so
bash tools/round3/ab.sh fixed_middle_tail0_again KAMD_LIB_PATH=$L/libkaolin_amd_order3tail0.so
} 2>&1 | tee $out/ab.txt
