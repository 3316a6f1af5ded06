#!/bin/bash
# round 3, call 5: covered-row centre instead of the ticketed sort; both backward passes in one launch
set -u
out=gpurun_out/r03e; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q --durations=5 --timeout 280 > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
grep -E "^(FAILED|ERROR)" $out/pytest_gpu.log | head -20
{
bash tools/round3/ab.sh base
bash tools/round3/ab.sh two_backward_launches KAMD_BWD_FUSED=2
bash tools/round3/ab.sh row_centre_off KAMD_ROW_ORDER=2
bash tools/round3/ab.sh top_of_image -- --look-at 0 -0.62 0
bash tools/round3/ab.sh top_of_image_row_centre_off KAMD_ROW_ORDER=2 -- --look-at 0 -0.62 0
bash tools/round3/ab.sh base_again
bash tools/round3/ab.sh two_backward_launches_again KAMD_BWD_FUSED=2
for n in 8 12 24; do bash tools/round3/ab.sh fused_bwd_per_cu_$n KAMD_SOFT_BWD_PER_CU=$n; done
} 2>&1 | tee $out/ab.txt
