#!/bin/bash
# round 4, call 5: main = round-3 select with the LDS-free transpose; full GPU suite, A/B, knot scene
set -u
out=gpurun_out/r04c5; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -rf --durations=8 > $out/pytest_gpu.log 2>&1; tail -14 $out/pytest_gpu.log | cut -c1-300
for i in 1 2 3; do bash tools/round3/ab.sh main_transpose; done 2>&1 | tee $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh main_transpose_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-360
