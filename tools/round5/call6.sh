#!/bin/bash
# round 5, call 6: where the rasterizer's tile kernel spends its wave time (phase ticks), all tiles / the heavy tiles alone;
# the sweep's counters at 1M x 50k; host profile of the DIB-R step
set -u
repo=$(pwd); out=$repo/gpurun_out/r05f; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
{
echo "== phase ticks, all tiles"; KAMD_LIB_PATH=${L}_prof.so timeout 100 python tools/phase_prof.py
echo "== phase ticks, tiles of more than 48 faces only (WRONG RESULTS)"; KAMD_LIB_PATH=${L}_profh.so timeout 100 python tools/phase_prof.py
} 2>&1 | grep -v amdgpu.ids > $out/raster_phases.txt
cat $out/raster_phases.txt
KAMD_LIB_PATH=${L}_exp.so KAMD_TS_STATS=1 timeout 100 python tools/time_tridist.py 1000000 2>&1 | grep -v amdgpu.ids > $out/ts_stats.txt; cat $out/ts_stats.txt
timeout 200 python tools/host_profile_dibr.py 2>&1 | grep -v amdgpu.ids | head -60 > $out/host_profile_dibr.txt; head -45 $out/host_profile_dibr.txt
