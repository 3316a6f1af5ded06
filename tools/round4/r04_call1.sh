#!/bin/bash
# round 4, call 1: the full GPU suite on the new build (element-wise tolerances, knot scene, K8 at C5, fp64 batched, contraction
# variants, signature fallback), then A/B of the soft-mask backward / eval instruction diet against round 3's library
set -u
out=gpurun_out/r04c1; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -rf --durations=12 > $out/pytest_gpu.log 2>&1; tail -40 $out/pytest_gpu.log | cut -c1-400
for i in 1 2; do
bash tools/round3/ab.sh r03_lib KAMD_LIB_PATH=$L/libkaolin_amd_r03.so
bash tools/round3/ab.sh r04_diet
done 2>&1 | tee $out/ab.txt | cut -c1-330
for p in 8 12 24; do bash tools/round3/ab.sh bwd_per_cu_$p KAMD_SOFT_BWD_PER_CU=$p; done 2>&1 | tee -a $out/ab.txt | cut -c1-330
