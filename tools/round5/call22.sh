#!/bin/bash
# round 5, call 22: the DIB-R kernels timed by their own begin / end timestamps (hipExtLaunchKernelGGL events) -- do they agree with the trace?
set -u
repo=$(pwd); out=$repo/gpurun_out/r05y; mkdir -p $out
timeout 300 python -m pytest tests/test_graph_capture.py tests/test_render_fused.py tests/test_dibr_gpu.py -m gpu -q -x --timeout 300 > $out/pytest.log 2>&1; tail -2 $out/pytest.log
timeout 320 python bench.py --no-cpu-baseline --no-contract-ops --no-scene-variants 2> $out/bench.err | tail -1 > $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $repo/bench.py --no-cpu-baseline --no-contract-ops --no-scene-variants 2> $out/prof.err | tail -1 > $out/bench_under_rocprof.json
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/prof
cd $repo
python - <<'P'
import json, csv
for f in ('bench.json', 'bench_under_rocprof.json'):
    d = json.load(open('gpurun_out/r05y/' + f))
    print(f, 'ms_per_step', d['ms_per_step'], 'median', d.get('median_ms_per_step'), 'graph', d.get('graph_replay_ms_per_step'), 'roofline us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])
    print('   ', {k: v['avg_us'] for k, v in d['kernels'].items()})
rows = list(csv.DictReader(open('gpurun_out/r05y/kernel_stats.csv')))
for r in rows[:14]:
    print('%-70s %6s calls  %8.2f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
P
