#!/bin/bash
# raster_tile: speculative inline entries, 7 vs 8 waves/SIMD, ablation branches compiled out
set -u
tag=r03k; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 280 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
L=$repo/kaolin_amd
{ bash tools/round3/ab.sh w8spec
  for v in w8nospec w7spec w7nospec; do bash tools/round3/ab.sh $v KAMD_LIB_PATH=$L/libkaolin_amd_$v.so; done
  bash tools/round3/ab.sh w8spec_again
  for v in w8nospec w7spec w7nospec; do bash tools/round3/ab.sh ${v}_again KAMD_LIB_PATH=$L/libkaolin_amd_$v.so; done
  bash tools/round3/ab.sh w8spec_top -- --look-at 0 -0.62 0
  bash tools/round3/ab.sh w7spec_top KAMD_LIB_PATH=$L/libkaolin_amd_w7spec.so -- --look-at 0 -0.62 0
} > $out/ab.txt 2>&1; cat $out/ab.txt
