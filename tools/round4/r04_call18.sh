#!/bin/bash
# round 4, call 18: rasterizer pass -- faces of more than 16 / 32 / 64 tiles on the big list instead of in per-tile entries
set -u
out=gpurun_out/r04c18; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_tile_order.py tests/test_full_size_parity.py -m gpu -x -q 2>&1 | tail -2 | tee $out/pytest.txt
for v in "" bigr32 bigr64; do
  lib=$L/libkaolin_amd${v:+_$v}.so
  bash tools/round3/ab.sh "sphere_${v:-bigr16}" KAMD_LIB_PATH=$lib 2>&1 | tee -a $out/ab.txt | cut -c1-200
  bash tools/round3/ab.sh "knot_${v:-bigr16}" KAMD_LIB_PATH=$lib -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
timeout 300 python tools/round4/knot_parts.py 2>&1 | tee $out/knot_parts.txt | cut -c1-260
