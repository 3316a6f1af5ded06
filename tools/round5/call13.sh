#!/bin/bash
# round 5, call 13: soft_select writes its pairs from the pixels' side (second transpose + DPP scan) instead of a loop over the faces
set -u
repo=$(pwd); out=$repo/gpurun_out/r05o; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_full_size_parity.py tests/test_render_fused.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
{
for i in 1 2; do
for sc in sphere knot; do
echo "== product (pairs by pixel) $sc"; timeout 100 python tools/round5/raster_fwd.py 30 $sc
echo "== loop over faces $sc"; KAMD_LIB_PATH=${L}_byface.so timeout 100 python tools/round5/raster_fwd.py 30 $sc
done
done
} 2>&1 | grep -v amdgpu.ids > $out/select_pairs_ab.txt
cat $out/select_pairs_ab.txt
