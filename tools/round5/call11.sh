#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r05m; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
{
for lib in "" _prev _ep128 _ep256 _ep1024 _ep4096; do
  for sc in sphere knot; do
    echo "== lib${lib:-_product(512)} $sc"; if [ -z "$lib" ]; then timeout 100 python tools/round5/raster_fwd.py 30 $sc; else KAMD_LIB_PATH=${L}${lib}.so timeout 100 python tools/round5/raster_fwd.py 30 $sc; fi
  done
done
} 2>&1 | grep -v amdgpu.ids > $out/eval_pair_max_ab.txt
cat $out/eval_pair_max_ab.txt
