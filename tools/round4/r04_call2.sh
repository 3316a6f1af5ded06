#!/bin/bash
# round 4, call 2: soft_prep_kernel + soft_select_listed_kernel -- the DIB-R tests, then the step against round 3's library
set -u
out=gpurun_out/r04c2; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_graph_capture.py tests/test_distributed.py -m gpu -q --timeout 400 -x -rf > $out/pytest_gpu.log 2>&1; tail -30 $out/pytest_gpu.log | cut -c1-300
for i in 1 2; do
bash tools/round3/ab.sh r03_lib KAMD_LIB_PATH=$L/libkaolin_amd_r03.so
bash tools/round3/ab.sh r04_prep
done 2>&1 | tee $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh r04_prep_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh r03_knot KAMD_LIB_PATH=$L/libkaolin_amd_r03.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-360
