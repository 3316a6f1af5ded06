#!/bin/bash
set -u
tag=r03x; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
{ for thr in 6 8 12 16 24 32; do
  echo "== wave per hard query, threshold $thr"; KAMD_TS_HARD_WAVE=1 KAMD_TS_HARD_THRESHOLD=$thr python tools/time_tridist.py 65536 200000 400000 700000 2>&1 | grep point_to_mesh
done; } > $out/ts_hard_threshold_by_n.txt 2>&1; cut -c1-120 $out/ts_hard_threshold_by_n.txt
