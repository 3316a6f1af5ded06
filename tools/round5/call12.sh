#!/bin/bash
# round 5, call 12: the id expansion A/B inside the FULL step (bench --quick), not only in the forward-only loop
set -u
repo=$(pwd); out=$repo/gpurun_out/r05n; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
q() { echo "== $*"; env "$@" timeout 200 python bench.py --quick --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'median', d['median_ms_per_step'], 'raster timed', d['roofline']['avg_launch_us'], {k: v['avg_us'] for k, v in d['kernels'].items()})"; }
{
for i in 1 2 3; do
q KAMD_X=product
q KAMD_LIB_PATH=${L}_oldexp.so
done
} > $out/expansion_in_step_ab.txt 2>&1
cat $out/expansion_in_step_ab.txt
