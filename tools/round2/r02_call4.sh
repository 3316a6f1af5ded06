#!/bin/bash
# Round 2, GPU call 4: fence-free chamfer build / query timings, DIB-R with the background prefill on the side stream
# (tests + A/B), point_to_mesh A/B of the slab / plane bounds with the sweep's counters.  Output -> gpurun_out/r02h/.
set -u
out=gpurun_out/r02h; mkdir -p $out
timeout 180 python tools/check_chamfer.py > $out/check_chamfer.txt 2>&1; echo "check_chamfer rc=$?"; tail -3 $out/check_chamfer.txt
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_graph_capture.py tests/test_sided_distance.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for m in 1 2; do
  echo "KAMD_DIBR_PREFILL=$m"
  KAMD_DIBR_PREFILL=$m timeout 300 python bench.py --no-cpu-baseline --no-chamfer --no-c5 2>> $out/bench.err | tail -1 > $out/bench_prefill$m.json
  python - $out/bench_prefill$m.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print('  ms/step', j['ms_per_step'], j['per_step_ms'], 'fg', j['feature_grad_variant']['per_step_ms']['median'])
    print('  ', {k.replace('_kernel', ''): v['avg_us'] for k, v in j['kernels'].items()})
except Exception as e:
    print('bench failed', e)
PY
done
tail -5 $out/bench.err
for m in 0 1 2 3; do
  echo "KAMD_TS_MODE=$m"
  KAMD_TS_MODE=$m KAMD_TS_STATS=1 timeout 300 python tools/time_tridist.py 2>&1 | grep -v Warn | grep "ts stats\|point_to_mesh\|td_" | tail -4
done > $out/ts_modes.txt 2>&1
cat $out/ts_modes.txt
