#!/bin/bash
# raster_backward: run sums straight to global atomics (no LDS table) vs the LDS table
set -u
tag=r03o; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_tile_order.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 280 > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
L=$repo/kaolin_amd
{ bash tools/round3/ab.sh direct
  bash tools/round3/ab.sh table KAMD_LIB_PATH=$L/libkaolin_amd_table.so
  bash tools/round3/ab.sh direct_per_tile KAMD_BWD_COV_LIST=2
  bash tools/round3/ab.sh direct_per_cu_4 KAMD_RBWD_PER_CU=4
  bash tools/round3/ab.sh direct_per_cu_16 KAMD_RBWD_PER_CU=16
  bash tools/round3/ab.sh direct_ungrouped KAMD_RBWD_GROUPED=2
  bash tools/round3/ab.sh direct_again
  bash tools/round3/ab.sh direct_top -- --look-at 0 -0.62 0
} > $out/ab.txt 2>&1; cat $out/ab.txt
timeout 200 python bench.py --no-cpu-baseline --no-contract-ops 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['frac'], j.get('feature_grad'), j.get('tutorial_objective'))"
