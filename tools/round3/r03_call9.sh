#!/bin/bash
# raster_tile: runs of TPW tiles per workgroup, build variants
set -u
tag=r03j; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 280 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
L=$repo/kaolin_amd
{ bash tools/round3/ab.sh tpw4_w8
  for v in tpw1 tpw2 tpw2w7 tpw4w7 tpw4w6 tpw8w7; do bash tools/round3/ab.sh $v KAMD_LIB_PATH=$L/libkaolin_amd_$v.so; done
  bash tools/round3/ab.sh tpw4_w8_again
  bash tools/round3/ab.sh tpw4_w8_top -- --look-at 0 -0.62 0
  bash tools/round3/ab.sh tpw1_top KAMD_LIB_PATH=$L/libkaolin_amd_tpw1.so -- --look-at 0 -0.62 0
} > $out/ab.txt 2>&1; cat $out/ab.txt
