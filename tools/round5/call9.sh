#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r05j; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 600 python -m pytest tests/test_triangle_distance.py tests/test_triangle_distance_fuzz.py -m gpu -q -x --timeout 300 2>&1 | tail -3
{
for i in 1 2; do
echo "== product (prefetch)"; timeout 100 python tools/time_tridist.py 300000 1000000
echo "== previous"; KAMD_LIB_PATH=${L}_prev.so timeout 100 python tools/time_tridist.py 300000 1000000
done
} 2>&1 | grep -v amdgpu.ids > $out/ts_prefetch_ab.txt; cat $out/ts_prefetch_ab.txt
