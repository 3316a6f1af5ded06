#!/bin/bash
# round 5, call 4: a wavefront per tile (raster4.inc) against a workgroup per tile; occupancy variants of both
set -u
repo=$(pwd); out=$repo/gpurun_out/r05d; mkdir -p $out
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_graph_capture.py tests/test_tile_order.py -m gpu -q -x --timeout 600 > $out/pytest_dibr.log 2>&1; tail -5 $out/pytest_dibr.log
q() { echo "== $* ${EXTRA:-}"; env "$@" timeout 200 python bench.py --quick --steps 50 ${EXTRA:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'median', d['median_ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"; }
L=$repo/kaolin_amd/libkaolin_amd
{
q KAMD_X=product
q KAMD_LIB_PATH=${L}_exp.so KAMD_RASTER_GEN=2
q KAMD_LIB_PATH=${L}_exp.so KAMD_RASTER_GEN=4
q KAMD_LIB_PATH=${L}_r4w6.so
q KAMD_LIB_PATH=${L}_r4w5.so
q KAMD_LIB_PATH=${L}_r2w7.so KAMD_RASTER_GEN=2
q KAMD_LIB_PATH=${L}_r2w6.so KAMD_RASTER_GEN=2
EXTRA='--scene knot' q KAMD_X=product
EXTRA='--scene knot' q KAMD_LIB_PATH=${L}_exp.so KAMD_RASTER_GEN=2
} > $out/raster4_ab.txt 2>&1
cat $out/raster4_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $repo/bench.py --quick --steps 50 > /dev/null 2> $out/prof.err
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/prof
grep -E "raster|bin_faces" $out/kernel_stats.csv | cut -c1-60,200-400
