#!/bin/bash
# round 4, call 30: grids of the other persistent kernels on both scenes (eval, soft backward, rasterizer backward)
set -u
out=gpurun_out/r04c30; mkdir -p $out
L=$(pwd)/kaolin_amd
for sc in sphere knot; do
  fl=""; [ $sc = knot ] && fl="--scene knot"
  bash tools/round3/ab.sh ${sc}_default KAMD_LIB_PATH=$L/libkaolin_amd_exp.so -- $fl 2>&1 | tee -a $out/ab.txt | cut -c1-220
  for cfg in KAMD_SOFT_EVAL_PER_CU=16 KAMD_SOFT_EVAL_PER_CU=128 KAMD_SOFT_BWD_PER_CU=8 KAMD_SOFT_BWD_PER_CU=32 KAMD_RBWD_PER_CU=8 KAMD_RBWD_PER_CU=32; do
    bash tools/round3/ab.sh ${sc}_$cfg KAMD_LIB_PATH=$L/libkaolin_amd_exp.so $cfg -- $fl 2>&1 | tee -a $out/ab.txt | cut -c1-220
  done
done
