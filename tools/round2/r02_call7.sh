#!/bin/bash
# Round 2, GPU call 7: point_to_mesh with the workgroup candidate list (tests, knob sweeps, counters); chamfer with the
# resident-set query launch.  Output -> gpurun_out/r02k/.
set -u
out=gpurun_out/r02k; mkdir -p $out
timeout 600 python -m pytest tests/test_triangle_distance.py tests/test_full_size_parity.py tests/test_sided_distance.py tests/test_graph_capture.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -2 $out/pytest.log
{
for few in 12 24 40 64; do for t in 32 48 64; do
  echo "== KAMD_TS_FEW=$few KAMD_TS_HARD_THRESHOLD=$t"
  KAMD_TS_FEW=$few KAMD_TS_HARD_THRESHOLD=$t timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "point_to_mesh\|td_"
done; done
echo "== default knobs, both sizes"
timeout 300 python tools/time_tridist.py 2>&1 | grep "point_to_mesh\|td_"
echo "== stats default"
KAMD_TS_STATS=1 timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "ts stats" | tail -1
} > $out/ts.txt 2>&1
cat $out/ts.txt
KAMD_CHECK_SPLIT=1 timeout 180 python tools/check_chamfer.py 2>&1 | grep "forward\|plain\|step\|OK" | tee $out/chamfer.txt
