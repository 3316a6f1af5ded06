#!/bin/bash
# round 6, call 7: raster_tile's prologue diet (head arguments preloaded, counters' loads together) + wide feature gather: parity, A/B, phases
set -u
repo=$(pwd); out=$repo/gpurun_out/r06g; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 1200 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_tile_order.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -4 $out/pytest_dibr.log
f() { echo "== $*"; env "$@" timeout 200 python tools/round5/raster_fwd.py 300 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2 3; do
f KAMD_X=product_prologue_diet
f KAMD_LIB_PATH=${L}_base.so
done
SCENE=knot f KAMD_X=product_prologue_diet
SCENE=knot f KAMD_LIB_PATH=${L}_base.so
} > $out/raster_prologue_ab.txt 2>&1
cat $out/raster_prologue_ab.txt
KAMD_LIB_PATH=${L}_prof.so timeout 300 python tools/phase_prof.py > $out/raster_phases.txt 2>&1; grep -A14 "^raster_tile" $out/raster_phases.txt
