#!/bin/bash
# round 4, call 23: the two backward kernels side by side again (soft-mask backward on the library's side stream), now that one is VALU-bound
set -u
out=gpurun_out/r04c23; mkdir -p $out
L=$(pwd)/kaolin_amd
for rep in 1 2; do
bash tools/round3/ab.sh one_stream KAMD_LIB_PATH=$L/libkaolin_amd_exp.so 2>&1 | tee -a $out/ab.txt | cut -c1-120
bash tools/round3/ab.sh side_stream KAMD_LIB_PATH=$L/libkaolin_amd_exp.so KAMD_BWD_SIDE_STREAM=1 2>&1 | tee -a $out/ab.txt | cut -c1-120
done
