#!/bin/bash
# round 6, call 8: a single-round tile's NaN redo is wavefront-local (one workgroup barrier less per tile), the first round's barrier dropped: parity + A/B
set -u
repo=$(pwd); out=$repo/gpurun_out/r06h; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 1200 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_tile_order.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -4 $out/pytest_dibr.log
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2 3; do
f KAMD_X=product_nan_redo_wave_local
f KAMD_LIB_PATH=${L}_base.so
done
SCENE=knot f KAMD_X=product_nan_redo_wave_local
SCENE=knot f KAMD_LIB_PATH=${L}_base.so
} > $out/raster_nan_local_ab.txt 2>&1
cat $out/raster_nan_local_ab.txt
