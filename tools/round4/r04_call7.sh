#!/bin/bash
# round 4, call 7: fp16 grid search; medium faces two per append step; full GPU suite
set -u
out=gpurun_out/r04c7; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -rf --durations=5 > $out/pytest_gpu.log 2>&1; tail -12 $out/pytest_gpu.log | cut -c1-300
python tools/round4/time_sd_f16.py 2>&1 | tail -1 | tee $out/sd_f16.txt
for i in 1 2; do bash tools/round3/ab.sh main; done 2>&1 | tee $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh main_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-360
