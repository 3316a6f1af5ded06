#!/bin/bash
# round 4, call 32: the eval kernel skips the items the search found nothing for
set -u
out=gpurun_out/r04c32; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py -m gpu -x -q 2>&1 | tail -2 | tee $out/pytest.txt
for rep in 1 2; do
bash tools/round3/ab.sh head_sphere KAMD_LIB_PATH=$L/libkaolin_amd_head.so 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh new_sphere 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
bash tools/round3/ab.sh head_knot KAMD_LIB_PATH=$L/libkaolin_amd_head.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh new_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
