#!/bin/bash
# round 4, call 6: medium faces binned a lane per tile; ablation / variant branches removed; env knobs compiled out
set -u
out=gpurun_out/r04c6; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -rf --durations=5 > $out/pytest_gpu.log 2>&1; tail -10 $out/pytest_gpu.log | cut -c1-300
for i in 1 2; do bash tools/round3/ab.sh main_clean; done 2>&1 | tee $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh main_clean_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-360
