#!/bin/bash
# round 4, call 3: soft_prep_kernel v2 (512 threads, one counting pass), LDS-free transpose in select_chunk
set -u
out=gpurun_out/r04c3; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_graph_capture.py tests/test_distributed.py -m gpu -q --timeout 400 -rf > $out/pytest_gpu.log 2>&1; tail -15 $out/pytest_gpu.log | cut -c1-400
for i in 1 2; do
bash tools/round3/ab.sh r03_lib KAMD_LIB_PATH=$L/libkaolin_amd_r03.so
bash tools/round3/ab.sh r04_prep2
done 2>&1 | tee $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh r04_prep2_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-360
