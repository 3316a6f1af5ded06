#!/bin/bash
# round 4, call 34: the randomised sweeps once more from other seeds, on the final library
set -u
out=gpurun_out/r04c34; mkdir -p $out
for t in "fuzz_dibr.py 300 1000" "fuzz_soft_mask.py 150 1000" "fuzz_rasterize_ops.py 150 1000" "fuzz_tridist.py 150 1000" "fuzz_dibr_nonfinite.py 150 1000"; do
  echo "== $t" | tee -a $out/fuzz.txt
  timeout 70 python tools/round4/$t 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $out/fuzz.txt
done
