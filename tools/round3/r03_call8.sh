#!/bin/bash
# 3D grid + barrier-free background tiles in raster_tile: DIB-R tests, ablation ladder with instruction counts, step timing
set -u
tag=r03i; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 280 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
bash tools/round3/r03_raster_insts.sh $tag > /dev/null 2>&1; cat $out/raster_modes_time.txt $out/raster_modes_pmc.txt
{ bash tools/round3/ab.sh base
  bash tools/round3/ab.sh base_again
} > $out/ab.txt 2>&1; cat $out/ab.txt
