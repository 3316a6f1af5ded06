#!/bin/bash
# round 4, call 14: what a medium-face step of the binning kernel waits for (timing-only builds that stop after the binning launch)
set -u
out=gpurun_out/r04c14; mkdir -p $out
L=$(pwd)/kaolin_amd
for a in 16 1 2 4 8 15; do
  echo -n "abl $a: " | tee -a $out/bin_only.txt
  KAMD_LIB_PATH=$L/libkaolin_amd_tabl$a.so timeout 120 python tools/round4/bin_only.py 2>&1 | tail -1 | tee -a $out/bin_only.txt
done
