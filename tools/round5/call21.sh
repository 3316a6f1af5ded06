#!/bin/bash
# round 5, call 21: the tile kernel's instruction count -- background stores as scalar base + one lane offset; 94 SGPRs at 8 waves/SIMD
set -u
repo=$(pwd); out=$repo/gpurun_out/r05x; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2 3; do
f KAMD_X=product_new_bg_stores
f KAMD_LIB_PATH=${L}_oldbg.so
f KAMD_LIB_PATH=${L}_sgpr96.so
done
SCENE=knot f KAMD_X=product_new_bg_stores
SCENE=knot f KAMD_LIB_PATH=${L}_oldbg.so
SCENE=knot f KAMD_LIB_PATH=${L}_sgpr96.so
} > $out/raster_bg_stores_ab.txt 2>&1
cat $out/raster_bg_stores_ab.txt
