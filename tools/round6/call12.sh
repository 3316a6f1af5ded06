#!/bin/bash
# round 6, call 12: raster_tile: fewer faces staged per round (LDS per workgroup: more workgroups resident now that background
# workgroups keep one wavefront) and no row-span load in front of the counters (A/B)
set -u
repo=$(pwd); out=$repo/gpurun_out/r06l; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2; do
f KAMD_X=product
f KAMD_LIB_PATH=${L}_cap128.so
f KAMD_LIB_PATH=${L}_cap96.so
f KAMD_LIB_PATH=${L}_norow.so
done
SCENE=knot f KAMD_X=product
SCENE=knot f KAMD_LIB_PATH=${L}_cap128.so
SCENE=knot f KAMD_LIB_PATH=${L}_cap96.so
SCENE=knot f KAMD_LIB_PATH=${L}_norow.so
} > $out/raster_cap_norow_ab.txt 2>&1
cat $out/raster_cap_norow_ab.txt
