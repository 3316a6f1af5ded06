#!/bin/bash
# Round 2, GPU call 8: point_to_mesh hard kernel seeded with the sweep's bound (tests, threshold sweep); chamfer query grid.
set -u
out=gpurun_out/r02l; mkdir -p $out
timeout 600 python -m pytest tests/test_triangle_distance.py tests/test_full_size_parity.py tests/test_sided_distance.py tests/test_graph_capture.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -2 $out/pytest.log
{
for few in 32; do for t in 12 16 24 32 48; do
  echo "== KAMD_TS_FEW=$few KAMD_TS_HARD_THRESHOLD=$t"
  KAMD_TS_FEW=$few KAMD_TS_HARD_THRESHOLD=$t timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "point_to_mesh\|td_"
done; done
echo "== default knobs, both sizes"
timeout 300 python tools/time_tridist.py 2>&1 | grep "point_to_mesh\|td_"
} > $out/ts.txt 2>&1
cat $out/ts.txt
for q in 4 8 16 32 64; do echo "== KAMD_SDG_QUERY_PER_CU=$q"; KAMD_SDG_QUERY_PER_CU=$q KAMD_CHECK_SPLIT=1 timeout 180 python tools/check_chamfer.py 2>&1 | grep "forward, value\|step"; done | tee $out/chamfer.txt
