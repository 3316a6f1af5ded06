#!/bin/bash
# round 3, call 2: GPU suite on the flat hit list / ballot pair writes / hoisted bin loads / data-driven row order / K7 fma pin,
# the K7 test statistics on the new pin, then A/B lines (build variants through KAMD_LIB_PATH, grid knobs, row order on / off)
set -u
out=gpurun_out/r03b; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q --durations=8 --timeout 280 > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.log | head -20
timeout 200 python tools/k7_contraction_ab.py --seeds 200 > $out/k7_fma_pinned.json 2> $out/k7.err; tail -c 700 $out/k7_fma_pinned.json; echo
L=$(pwd)/kaolin_amd
{
bash tools/round3/ab.sh base
bash tools/round3/ab.sh base_again
bash tools/round3/ab.sh select_pairs_lds KAMD_LIB_PATH=$L/libkaolin_amd_selpairs0.so
bash tools/round3/ab.sh select_min_waves_6 KAMD_LIB_PATH=$L/libkaolin_amd_selw6.so
bash tools/round3/ab.sh eval_waves_8 KAMD_LIB_PATH=$L/libkaolin_amd_evalw8.so
for n in 8 12 24 32; do bash tools/round3/ab.sh bwd_per_cu_$n KAMD_SOFT_BWD_PER_CU=$n; done
bash tools/round3/ab.sh row_order_off KAMD_ROW_ORDER=0
bash tools/round3/ab.sh offcentre_row_order_on -- --look-at 0.45 -0.4 0
bash tools/round3/ab.sh offcentre_row_order_off KAMD_ROW_ORDER=0 -- --look-at 0.45 -0.4 0
bash tools/round3/ab.sh select_per_cu_24 KAMD_SOFT_SELECT_PER_CU=24
bash tools/round3/ab.sh eval_per_cu_16 KAMD_SOFT_EVAL_PER_CU=16
} 2>&1 | tee $out/ab.txt
