#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats over bench.py on the headline scene only (the default run also times the knot scene, whose
# launches of the same kernels would be averaged in)
set -u
repo=$(pwd); out=$repo/gpurun_out/r04final; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $repo/bench.py --no-cpu-baseline --no-contract-ops --no-scene-variants 2> $out/prof.err | tail -1 > $out/bench_under_rocprof.json
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/prof
head -8 $out/kernel_stats.csv | cut -c1-200
