#!/bin/bash
# round 5, call 18: raster_tile's row order -- the covered rows from their two ends inwards (heavy silhouette rows first) vs from the middle outwards
set -u
repo=$(pwd); out=$repo/gpurun_out/r05t; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 300 python -m pytest tests/test_dibr_gpu.py -m gpu -q -x --timeout 300 -k "fused or dibr_rasterization" 2>&1 | tail -2
{
for i in 1 2 3; do
for lib in "" _outin; do
  for sc in sphere knot; do
    echo "== lib${lib:-_product(middle-out)} $sc"; if [ -z "$lib" ]; then timeout 100 python tools/round5/raster_fwd.py 30 $sc; else KAMD_LIB_PATH=${L}${lib}.so timeout 100 python tools/round5/raster_fwd.py 30 $sc; fi
  done
done
done
} 2>&1 | grep -v amdgpu.ids > $out/raster_row_order_ab.txt
grep -A1 "^==" $out/raster_row_order_ab.txt | grep -v "^--" | paste - - | sed -E "s/.*== (lib[^ ]* [a-z]+).*'raster_tile_kernel': ([0-9.]+).*/\1 \2/"
