#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r03p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in b123 b11; do
KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd_$v.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -- python $repo/bench.py --quick --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find $out/prof_$v -name '*kernel_stats.csv' | head -1)
echo "== $v"; python - <<P
import csv
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if any(k in n for k in ('bin_faces','zero3','raster_tile','fill16','pv_forward')): print(n[:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
P
rm -rf $out/prof_$v
done
