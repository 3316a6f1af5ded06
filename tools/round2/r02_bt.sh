#!/bin/bash
set -u
out=gpurun_out/r02y; mkdir -p $out
for v in "" bt1024 bt256; do
  for w in 64 128 256; do
  echo "== variant ${v:-default} KAMD_SDG_WGS=$w"
  if [ -n "$v" ]; then export KAMD_LIB_PATH=$(pwd)/kaolin_amd/libkaolin_amd_$v.so; else unset KAMD_LIB_PATH; fi
  KAMD_SDG_WGS=$w timeout 180 python tools/check_chamfer.py 2>&1 | grep "step\|rror"
  done
done | tee $out/build_threads.txt
