#!/bin/bash
# soft-mask search: items addressed by place (first item requested with the shard counts), pair count in the item record, hit counts requested early
set -u
tag=r03q; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 280 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
{ bash tools/round3/ab.sh base
  bash tools/round3/ab.sh base_again
  bash tools/round3/ab.sh top -- --look-at 0 -0.62 0
} > $out/ab.txt 2>&1; cat $out/ab.txt
