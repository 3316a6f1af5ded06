#!/bin/bash
# round 5, call 7: the id expansion across the wavefronts (readlane + mbcnt, DPP scan) and the tail's two atomics issued together
set -u
repo=$(pwd); out=$repo/gpurun_out/r05h; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_full_size_parity.py tests/test_render_fused.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
{
for i in 1 2; do
echo "== product (new tail)"; timeout 100 python tools/round5/raster_fwd.py 30
echo "== previous build"; KAMD_LIB_PATH=${L}_exp.so timeout 100 python tools/round5/raster_fwd.py 30
done
echo "== knot product"; timeout 100 python tools/round5/raster_fwd.py 30 knot
echo "== knot previous"; KAMD_LIB_PATH=${L}_prev.so timeout 100 python tools/round5/raster_fwd.py 30 knot
} 2>&1 | grep -v amdgpu.ids > $out/tail_ab.txt
cat $out/tail_ab.txt
