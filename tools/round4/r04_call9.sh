#!/bin/bash
# round 4, call 9: the medium-face threshold of the binning kernel on the knot scene (and that the sphere does not care); the row-order knob
set -u
out=gpurun_out/r04c9; mkdir -p $out
L=$(pwd)/kaolin_amd
for v in 4 8; do
bash tools/round3/ab.sh knot_medium_tiles_$v KAMD_LIB_PATH=$L/libkaolin_amd_med$v.so -- --scene knot
bash tools/round3/ab.sh sphere_medium_tiles_$v KAMD_LIB_PATH=$L/libkaolin_amd_med$v.so
done 2>&1 | tee $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh sphere_row_order_image_middle KAMD_ROW_ORDER=2 KAMD_LIB_PATH=$L/libkaolin_amd_med8.so 2>&1 | tee -a $out/ab.txt | cut -c1-360
KAMD_LIB_PATH=$L/libkaolin_amd_med8.so timeout 300 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py -m gpu -q -x --timeout 280 2>&1 | tail -2
