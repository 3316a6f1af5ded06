#!/bin/bash
# round 6, call 2: per-tile candidate records (binning writes them, raster_tile streams them): parity + timing
set -u
repo=$(pwd); out=$repo/gpurun_out/r06b; mkdir -p $out
timeout 1200 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_tile_order.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -15 $out/pytest_dibr.log
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2 3; do
f KAMD_X=tile_records
done
SCENE=knot f KAMD_X=tile_records
} > $out/raster_tile_records.txt 2>&1
cat $out/raster_tile_records.txt
