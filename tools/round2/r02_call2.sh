set -u
out=gpurun_out/r02b; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py -q -x -m gpu > $out/pytest_dibr.log 2>&1; tail -3 $out/pytest_dibr.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-chamfer --no-c5 2> $out/bench.err | tail -1 > $out/bench.json
python - <<'PY'
import json
try:
    j = json.load(open('gpurun_out/r02b/bench.json'))
    print('ms_per_step', j['ms_per_step'], {k.replace('_kernel',''): v['avg_us'] for k, v in j['kernels'].items()})
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/r02b/bench.err').read()[-2000:])
PY
