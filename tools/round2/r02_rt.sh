#!/bin/bash
# raster_tile experiment: raster / DIB-R parity tests with the default build, then the bench's DIB-R section per build variant
set -u
out=gpurun_out/${1:-r02rt}; mkdir -p $out; shift
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_oracle.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_graph_capture.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -1 $out/pytest.log
bash tools/round2/r02_variants.sh "$@" 2>&1 | sed 's/^/  /'
