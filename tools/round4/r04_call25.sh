#!/bin/bash
# round 4, call 25: threads per work item in the soft mask's eval kernel (256 / 128 / 64) on both scenes
set -u
out=gpurun_out/r04c25; mkdir -p $out
L=$(pwd)/kaolin_amd
for v in "" evalt128 evalt64; do
  lib=$L/libkaolin_amd${v:+_$v}.so
  for pc in 32 64; do
  bash tools/round3/ab.sh "sphere_${v:-evalt256}_percu$pc" KAMD_LIB_PATH=$lib KAMD_SOFT_EVAL_PER_CU=$pc 2>&1 | tee -a $out/ab.txt | cut -c1-200
  bash tools/round3/ab.sh "knot_${v:-evalt256}_percu$pc" KAMD_LIB_PATH=$lib KAMD_SOFT_EVAL_PER_CU=$pc -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
  done
done
