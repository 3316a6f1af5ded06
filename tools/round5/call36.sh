#!/bin/bash
# round 5, call 36: the rasterizer's backward taking its covered-tile list backwards (A/B build) -- a sizing measurement for the next round
set -u
repo=$(pwd); out=$repo/gpurun_out/r05al; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
q() { echo "== $* ${EXTRA:-}"; env "$@" timeout 200 python bench.py --quick --steps 40 ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'median', d['median_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"; }
{
for i in 1 2; do
for sc in "" "--scene knot"; do
EXTRA="$sc" q KAMD_X=product
EXTRA="$sc" q KAMD_LIB_PATH=${L}_rbwdrev.so
done
done
} > $out/rbwd_order_ab.txt 2>&1
grep -o "ms_per_step [0-9.]*\|'raster_backward_kernel': [0-9.]*\|^== .*" $out/rbwd_order_ab.txt
