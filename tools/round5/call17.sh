#!/bin/bash
# round 5, call 17: faces staged per round / occupancy of raster_tile after the expansion change
set -u
repo=$(pwd); out=$repo/gpurun_out/r05s; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
{
for i in 1 2; do
for lib in "" _cap96 _cap128 _cap256 _w7; do
  for sc in sphere knot; do
    echo "== lib${lib:-_product(192, 8 waves)} $sc"; if [ -z "$lib" ]; then timeout 100 python tools/round5/raster_fwd.py 30 $sc; else KAMD_LIB_PATH=${L}${lib}.so timeout 100 python tools/round5/raster_fwd.py 30 $sc; fi
  done
done
done
} 2>&1 | grep -v amdgpu.ids > $out/raster_cap_ab.txt
cat $out/raster_cap_ab.txt
