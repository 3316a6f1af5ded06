#!/bin/bash
set -u
out=gpurun_out/r02y; mkdir -p $out
for v in "" sdg4 sdg2; do
  echo "== variant ${v:-default}"
  if [ -n "$v" ]; then export KAMD_LIB_PATH=$(pwd)/kaolin_amd/libkaolin_amd_$v.so; else unset KAMD_LIB_PATH; fi
  KAMD_CHECK_SPLIT=1 timeout 180 python tools/check_chamfer.py 2>&1 | grep "forward, value\|plain\|step\|OK\|rror"
  timeout 300 python -m pytest tests/test_sided_distance.py tests/test_full_size_parity.py -q -x -m gpu -k "sided or chamfer or pair or grid or c3" --timeout 300 2>&1 | tail -1
done | tee $out/sdg.txt
