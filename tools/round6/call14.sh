#!/bin/bash
# round 6, call 14: raster_backward with wide gathers (A/B inside the full step) + parity
set -u
repo=$(pwd); out=$repo/gpurun_out/r06n; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
timeout 1200 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
f() { echo "== $*"; env "$@" timeout 300 python tools/round6/step_kernels.py 200 ${SCENE:-sphere} 2>/dev/null | tail -1; }
{
for i in 1 2 3; do
f KAMD_X=product
f KAMD_LIB_PATH=${L}_base.so
done
SCENE=knot f KAMD_X=product
SCENE=knot f KAMD_LIB_PATH=${L}_base.so
} > $out/step_rbwd_wide_ab.txt 2>&1
cat $out/step_rbwd_wide_ab.txt
