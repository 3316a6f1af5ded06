#!/bin/bash
# round 4, call 11: the binning kernel's paired atomics really in flight together (the result's first use moved behind both issues)
set -u
out=gpurun_out/r04c11; mkdir -p $out
L=$(pwd)/kaolin_amd
for rep in 1 2; do
bash tools/round3/ab.sh head_sphere KAMD_LIB_PATH=$L/libkaolin_amd_head.so 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh new_sphere 2>&1 | tee -a $out/ab.txt | cut -c1-220
done
bash tools/round3/ab.sh head_knot KAMD_LIB_PATH=$L/libkaolin_amd_head.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh new_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-220
timeout 300 python -m pytest tests/test_dibr_gpu.py tests/test_tile_order.py -m gpu -x -q 2>&1 | tail -3 | tee $out/pytest.txt
