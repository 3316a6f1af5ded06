#!/bin/bash
# Round 2, GPU call 6: point_to_mesh with the lane = face mode for sparsely wanted tiles (tests + knob sweeps), chamfer
# barrier knobs and the cost split of the fused query.  Output -> gpurun_out/r02j/.
set -u
out=gpurun_out/r02j; mkdir -p $out
timeout 600 python -m pytest tests/test_triangle_distance.py tests/test_full_size_parity.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -2 $out/pytest.log
{
for few in 1 3 6 12 24; do for t in 48 195; do
  echo "== KAMD_TS_FEW=$few KAMD_TS_HARD_THRESHOLD=$t"
  KAMD_TS_FEW=$few KAMD_TS_HARD_THRESHOLD=$t timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "point_to_mesh\|td_"
done; done
echo "== default knobs, both sizes"
timeout 300 python tools/time_tridist.py 2>&1 | grep "point_to_mesh\|td_"
echo "== stats default"
KAMD_TS_STATS=1 timeout 300 python tools/time_tridist.py 1000000 2>&1 | grep "ts stats" | tail -1
} > $out/ts.txt 2>&1
cat $out/ts.txt
{
KAMD_CHECK_SPLIT=1 timeout 180 python tools/check_chamfer.py 2>&1 | grep "forward\|plain\|step"
for w in 64 128 256; do for n in 1 4 16; do
  echo "== KAMD_SDG_WGS=$w KAMD_SDG_NAPS=$n"; KAMD_SDG_WGS=$w KAMD_SDG_NAPS=$n timeout 180 python tools/check_chamfer.py 2>&1 | grep "step"
done; done
} > $out/chamfer.txt 2>&1
cat $out/chamfer.txt
