#!/bin/bash
# round 4, call 28: the select kernel's grid (workgroups per CU) on the sphere
set -u
out=gpurun_out/r04c28; mkdir -p $out
L=$(pwd)/kaolin_amd
for rep in 1 2; do
for pc in 32 48 64 128; do
bash tools/round3/ab.sh sphere_select_per_cu_$pc KAMD_LIB_PATH=$L/libkaolin_amd_exp.so KAMD_SOFT_SELECT_PER_CU=$pc 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
done
