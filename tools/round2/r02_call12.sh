#!/bin/bash
set -u
out=gpurun_out/r02q; mkdir -p $out
timeout 300 python -m pytest tests/test_triangle_distance.py -q -x -m gpu --timeout 300 2>&1 | tail -1
for nt in 256 128 64; do
  echo "== KAMD_TS_THREADS=$nt"; KAMD_TS_THREADS=$nt timeout 300 python tools/time_tridist.py 2>&1 | grep "point_to_mesh\|td_"
done | tee $out/ts_threads.txt
KAMD_TS_THREADS=128 timeout 300 python -m pytest tests/test_triangle_distance.py -q -x -m gpu --timeout 300 2>&1 | tail -1
