#!/bin/bash
# Round 2, GPU call 5: chamfer (two-level scan, sorted-order gradient arrays) and point_to_mesh A/B: bounds on/off, hard
# threshold sweep, the sweep's counters.  Output -> gpurun_out/r02i/.
set -u
out=gpurun_out/r02i; mkdir -p $out
timeout 180 python tools/check_chamfer.py > $out/check_chamfer.txt 2>&1; echo "check_chamfer rc=$?"; tail -3 $out/check_chamfer.txt
timeout 600 python -m pytest tests/test_sided_distance.py tests/test_graph_capture.py tests/test_full_size_parity.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -2 $out/pytest.log
{
for m in 0 1 2 3; do
  echo "== KAMD_TS_MODE=$m"
  KAMD_TS_MODE=$m timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "point_to_mesh\|td_"
done
for t in 16 32 48 96; do
  echo "== KAMD_TS_HARD_THRESHOLD=$t (mode 0)"
  KAMD_TS_HARD_THRESHOLD=$t timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "point_to_mesh\|td_"
done
for m in 0 3; do
  echo "== stats, KAMD_TS_MODE=$m"
  KAMD_TS_MODE=$m KAMD_TS_STATS=1 timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "ts stats" | tail -1
done
echo "== stats, mode 0, threshold 32"
KAMD_TS_HARD_THRESHOLD=32 KAMD_TS_STATS=1 timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "ts stats" | tail -1
} > $out/ts_modes.txt 2>&1
cat $out/ts_modes.txt
