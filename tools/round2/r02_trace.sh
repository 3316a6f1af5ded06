#!/bin/bash
# kernel trace (start / end of every dispatch) of a few bench steps -> gpurun_out/r02t/kernel_trace.csv
set -u
repo=$(pwd); out=$repo/gpurun_out/r02t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python $repo/tools/pmc_traffic.py > $out/trace.log 2>&1
find $out/kt -name '*kernel_trace.csv' -exec cp {} $out/kernel_trace.csv \;
rm -rf $out/kt; ls -la $out
