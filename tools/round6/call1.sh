#!/bin/bash
# round 6, call 1: the access-pattern ceiling of raster_tile (tools/ubench/raster_ceiling.hip); background tiles written by wavefront 0 alone (A/B)
set -u
repo=$(pwd); out=$repo/gpurun_out/r06a; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 300 tools/ubench/raster_ceiling 200 > $out/raster_ceiling.txt 2>&1; cat $out/raster_ceiling.txt
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2 3; do
f KAMD_X=product_bg_one_wave
f KAMD_LIB_PATH=${L}_bg4.so
done
SCENE=knot f KAMD_X=product_bg_one_wave
SCENE=knot f KAMD_LIB_PATH=${L}_bg4.so
} > $out/raster_bg_one_wave_ab.txt 2>&1
cat $out/raster_bg_one_wave_ab.txt
