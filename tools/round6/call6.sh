#!/bin/bash
# round 6, call 6: what the prologue of 32 768 workgroups costs (product: every workgroup leaves once its count is known; ubench form F)
set -u
repo=$(pwd); out=$repo/gpurun_out/r06f; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 300 tools/ubench/raster_ceiling 200 > $out/raster_ceiling.txt 2>&1; cat $out/raster_ceiling.txt
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2; do
f KAMD_X=product
f KAMD_LIB_PATH=${L}_diag7.so
done
} > $out/raster_diag7.txt 2>&1
cat $out/raster_diag7.txt
