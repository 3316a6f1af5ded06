#!/bin/bash
# round 6, call 9: the whole GPU suite (incl. the staged reference tests, the sparse voxel grid, C3 batch8) + the bench line
set -u
repo=$(pwd); out=$repo/gpurun_out/r06i; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q -x --timeout 1800 > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06i/bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','roofline','cpu_baseline')})
print(d['chamfer'].get('batch8'), d['chamfer']['ms_per_step'], d['chamfer']['operator_only'])
print({k:v for k,v in d['c5'].items() if 'valu' in k or 'work' in k or k=='voxelgrid_256_us'})
print(d['kernels'])
PY
