#!/bin/bash
# point_to_mesh: a wavefront per hard query vs a workgroup per hard query, hard threshold sweep
set -u
tag=r03x; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 500 python -m pytest tests/test_triangle_distance.py -m gpu -q -x --timeout 280 2>&1 | tail -2
{ for env in "KAMD_TS_HARD_WAVE=2" "KAMD_TS_HARD_WAVE=1" "KAMD_TS_HARD_WAVE=1 KAMD_TS_HARD_THRESHOLD=24" "KAMD_TS_HARD_WAVE=1 KAMD_TS_HARD_THRESHOLD=16" "KAMD_TS_HARD_WAVE=1 KAMD_TS_HARD_THRESHOLD=12" "KAMD_TS_HARD_WAVE=1 KAMD_TS_HARD_THRESHOLD=8" "KAMD_TS_HARD_WAVE=1 KAMD_TS_HARD_THRESHOLD=16 KAMD_TS_HARD_PER_CU=8" "KAMD_TS_HARD_WAVE=1 KAMD_TS_HARD_THRESHOLD=16 KAMD_TS_HARD_PER_CU=32" "KAMD_TS_HARD_WAVE=2 KAMD_TS_HARD_THRESHOLD=16"; do
  echo "== $env"; env $env python tools/time_tridist.py 100000 1000000 2>&1 | grep -v amdgpu
done; } > $out/ts_hard_wave.txt 2>&1; cat $out/ts_hard_wave.txt | cut -c1-220
