#!/bin/bash
# one step of the DIB-R bench as a timeline (kernel, start, duration, gap) from a rocprofv3 kernel trace -> gpurun_out/<tag>/timeline.txt
set -u
tag=${1:-r03w}; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python $repo/bench.py --quick --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
f=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $repo/tools/trace_timeline.py $f > $out/timeline.txt 2>&1
rm -rf $out/tr
cat $out/timeline.txt
