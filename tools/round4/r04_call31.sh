#!/bin/bash
# round 4, call 31: the eval kernel compiled for 7 / 8 waves per SIMD (it sits at 80 VGPRs = 6) on both scenes
set -u
out=gpurun_out/r04c31; mkdir -p $out
L=$(pwd)/kaolin_amd
for rep in 1 2; do
for v in exp evw7 evw8; do
  bash tools/round3/ab.sh sphere_$v KAMD_LIB_PATH=$L/libkaolin_amd_$v.so 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
done
for v in exp evw7 evw8; do
  bash tools/round3/ab.sh knot_$v KAMD_LIB_PATH=$L/libkaolin_amd_$v.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
