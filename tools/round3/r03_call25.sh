#!/bin/bash
set -u
tag=r03u; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py -m gpu -q -x --timeout 280 2>&1 | tail -2
L=$repo/kaolin_amd
for i in 1 2 3; do
bash tools/round3/ab.sh stream_masks
bash tools/round3/ab.sh chunk_gather KAMD_LIB_PATH=$L/libkaolin_amd_nomasks.so
done | cut -c1-200
