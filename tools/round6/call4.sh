#!/bin/bash
# round 6, call 4: tiles with candidates alone / background alone (timing-only builds, WRONG RESULTS)
set -u
repo=$(pwd); out=$repo/gpurun_out/r06d; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2; do
f KAMD_X=product
f KAMD_LIB_PATH=${L}_diag5.so
f KAMD_LIB_PATH=${L}_diag6.so
done
SCENE=knot f KAMD_X=product
SCENE=knot f KAMD_LIB_PATH=${L}_diag5.so
SCENE=knot f KAMD_LIB_PATH=${L}_diag6.so
} > $out/raster_diag56.txt 2>&1
cat $out/raster_diag56.txt
