#!/bin/bash
set -u
L=$(pwd)/kaolin_amd
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py -m gpu -q -x --timeout 280 2>&1 | tail -2
for i in 1 2; do
bash tools/round3/ab.sh table_vertical
bash tools/round3/ab.sh table_rows_only KAMD_LIB_PATH=$L/libkaolin_amd_novert.so
bash tools/round3/ab.sh staged_vertical KAMD_LIB_PATH=$L/libkaolin_amd_vertdirect.so
done | cut -c1-140
