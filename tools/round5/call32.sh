#!/bin/bash
# round 5, call 32: the reference's own test files (unchanged; both ways) and the randomised sweeps of every operator on the final library
set -u
repo=$(pwd); out=$repo/gpurun_out/r05final_extra; mkdir -p $out
bash tools/run_reference_tests.sh r05final_extra > /dev/null 2>&1; cp $out/reference_tests.log $out/reference_tests_alias.log
KAMD_REF_LAYER=1 bash tools/run_reference_tests.sh r05final_extra > /dev/null 2>&1; cp $out/reference_tests.log $out/reference_tests_ref_layer.log
tail -2 $out/reference_tests_alias.log $out/reference_tests_ref_layer.log
{
for f in fuzz_dibr.py fuzz_soft_mask.py fuzz_rasterize_ops.py fuzz_dibr_nonfinite.py fuzz_sided.py fuzz_tridist.py fuzz_others.py fuzz_metrics_grad.py; do
  echo "== tools/round4/$f 120 cases from seed 11000"; timeout 600 python tools/round4/$f 120 11000 2>&1 | grep -v amdgpu.ids | tail -2
done
} > $out/fuzz.txt 2>&1
cat $out/fuzz.txt
