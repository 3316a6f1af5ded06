#!/bin/bash
set -u
L=$(pwd)/kaolin_amd
KAMD_LIB_PATH=$L/libkaolin_amd_vertdirect.so timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 280 2>&1 | tail -2
for i in 1 2; do
bash tools/round3/ab.sh staged_vertical KAMD_LIB_PATH=$L/libkaolin_amd_vertdirect.so
bash tools/round3/ab.sh staged_vertical_per_cu4 KAMD_RBWD_PER_CU=4 KAMD_LIB_PATH=$L/libkaolin_amd_vertdirect.so
bash tools/round3/ab.sh staged_vertical_per_cu16 KAMD_RBWD_PER_CU=16 KAMD_LIB_PATH=$L/libkaolin_amd_vertdirect.so
bash tools/round3/ab.sh table_vertical
done | cut -c1-140
KAMD_LIB_PATH=$L/libkaolin_amd_vertdirect.so timeout 200 python bench.py --quick --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.load(sys.stdin); print('staged feature-grad n/a in quick; step', j['ms_per_step'])"
