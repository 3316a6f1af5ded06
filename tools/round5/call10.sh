#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r05l; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_full_size_parity.py tests/test_render_fused.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
{
for i in 1 2; do
echo "== product (paired light items)"; timeout 100 python tools/round5/raster_fwd.py 30
echo "== previous build"; KAMD_LIB_PATH=${L}_prev.so timeout 100 python tools/round5/raster_fwd.py 30
echo "== knot product"; timeout 100 python tools/round5/raster_fwd.py 30 knot
echo "== knot previous"; KAMD_LIB_PATH=${L}_prev.so timeout 100 python tools/round5/raster_fwd.py 30 knot
done
} 2>&1 | grep -v amdgpu.ids > $out/eval_pairs_ab.txt
cat $out/eval_pairs_ab.txt
