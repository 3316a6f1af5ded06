#!/bin/bash
# round 6, call 10: the decomposition of raster_tile after the prologue diet (timing-only builds, WRONG RESULTS) + phase ticks
set -u
repo=$(pwd); out=$repo/gpurun_out/r06j; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2; do
f KAMD_X=product
f KAMD_LIB_PATH=${L}_diag7.so
f KAMD_LIB_PATH=${L}_diag6.so
f KAMD_LIB_PATH=${L}_diag5.so
done
} > $out/raster_diag567.txt 2>&1
cat $out/raster_diag567.txt
KAMD_LIB_PATH=${L}_prof.so timeout 300 python tools/phase_prof.py > $out/raster_phases.txt 2>&1; grep -A14 "^raster_tile" $out/raster_phases.txt; grep "kernel us" $out/raster_phases.txt
