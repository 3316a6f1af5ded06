#!/bin/bash
# round 6, call 3: where raster_tile's time goes with per-tile records: phase ticks + timing-only builds (WRONG RESULTS variants)
set -u
repo=$(pwd); out=$repo/gpurun_out/r06c; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
KAMD_LIB_PATH=${L}_prof.so timeout 300 python tools/phase_prof.py > $out/raster_phases.txt 2>&1; cat $out/raster_phases.txt
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2; do
f KAMD_X=product
f KAMD_LIB_PATH=${L}_diag1.so
f KAMD_LIB_PATH=${L}_diag2.so
f KAMD_LIB_PATH=${L}_diag3.so
f KAMD_LIB_PATH=${L}_diag4.so
done
} > $out/raster_diag.txt 2>&1
cat $out/raster_diag.txt
