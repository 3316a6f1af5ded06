#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r05q; mkdir -p $out
{
timeout 600 python tools/round4/fuzz_soft_mask.py 100 700 2>&1 | tail -4
timeout 600 python tools/round4/fuzz_rasterize_ops.py 100 700 2>&1 | tail -4
timeout 600 python tools/round4/fuzz_metrics_grad.py 60 700 2>&1 | tail -4
} | grep -v amdgpu.ids > $out/fuzz_round5_ops.txt
cat $out/fuzz_round5_ops.txt
