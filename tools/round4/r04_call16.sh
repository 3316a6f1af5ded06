#!/bin/bash
# round 4, call 16: how many counter atomics a medium face keeps in flight (4 / 8 / 12 / 16: 68 / 68 / 77 / 85 registers)
set -u
out=gpurun_out/r04c16; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_tile_order.py tests/test_full_size_parity.py -m gpu -x -q 2>&1 | tail -2 | tee $out/pytest.txt
for v in mb4 "" mb12 mb16; do
  lib=$L/libkaolin_amd${v:+_$v}.so
  bash tools/round3/ab.sh "sphere_${v:-mb8}" KAMD_LIB_PATH=$lib 2>&1 | tee -a $out/ab.txt | cut -c1-200
  bash tools/round3/ab.sh "knot_${v:-mb8}" KAMD_LIB_PATH=$lib -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
