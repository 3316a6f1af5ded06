#!/bin/bash
# round 6, call 13: raster_tile with TWO tiles per workgroup (one prologue, background pairs side by side): parity on the variant + A/B
set -u
repo=$(pwd); out=$repo/gpurun_out/r06m; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
KAMD_LIB_PATH=${L}_pair.so timeout 1200 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_tile_order.py -m gpu -q -x --timeout 600 > $out/pytest_dibr_pair.log 2>&1; tail -4 $out/pytest_dibr_pair.log
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2 3; do
f KAMD_LIB_PATH=${L}_base.so
f KAMD_LIB_PATH=${L}_pair.so
done
SCENE=knot f KAMD_LIB_PATH=${L}_base.so
SCENE=knot f KAMD_LIB_PATH=${L}_pair.so
} > $out/raster_pair_ab.txt 2>&1
cat $out/raster_pair_ab.txt
