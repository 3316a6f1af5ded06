# usage: bash tools/r02_sweep.sh VAR v1 v2 ...   -- quick bench per value of an environment knob
var=$1; shift
for v in "$@"; do
  env $var=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-chamfer --no-c5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$var=$v', j['ms_per_step'], {k.replace('_kernel',''): v['avg_us'] for k, v in j['kernels'].items()})"
done
