#!/bin/bash
# raster_tile: background rows written by fill planes beside the tiles with faces
set -u
tag=r03s; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 280 > $out/pytest_dibr.log 2>&1; tail -2 $out/pytest_dibr.log
{ for i in 1 2; do
  bash tools/round3/ab.sh fill
  bash tools/round3/ab.sh nofill KAMD_RASTER_FILL=2
done
bash tools/round3/ab.sh fill_top -- --look-at 0 -0.62 0
bash tools/round3/ab.sh nofill_top KAMD_RASTER_FILL=2 -- --look-at 0 -0.62 0
} > $out/ab.txt 2>&1; cut -c1-150 $out/ab.txt
