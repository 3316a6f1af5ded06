#!/bin/bash
set -u
tag=r03r; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py tests/test_edge_filter_math.py -m gpu -q -x --timeout 280 > $out/pytest_dibr.log 2>&1; tail -2 $out/pytest_dibr.log
L=$repo/kaolin_amd
{ for i in 1 2 3; do
  bash tools/round3/ab.sh min3
  bash tools/round3/ab.sh nomin3 KAMD_LIB_PATH=$L/libkaolin_amd_nomin3.so
done; } > $out/ab.txt 2>&1; cut -c1-130 $out/ab.txt
