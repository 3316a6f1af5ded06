#!/bin/bash
# tile visiting orders (build variants): DIB-R tests on the default build, then A/B through the bench's DIB-R section
set -u
out=gpurun_out/r02order; mkdir -p $out
D=$PWD/kaolin_amd
timeout 200 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py -q -x -m gpu --timeout 120 > $out/pytest.log 2>&1; tail -1 $out/pytest.log
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-chamfer --no-c5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.load(sys.stdin); print(j['ms_per_step'], j['per_step_ms']['median'], j['per_step_ms']['min'], 'fg', j['feature_grad_variant']['per_step_ms']['median'], {k.replace('_kernel',''): v['avg_us'] for k, v in j['kernels'].items()})"; }
{
echo "default (rows centre-out): $(run)"
for v in "$@"; do echo "$v: $(KAMD_LIB_PATH=$D/libkaolin_amd_$v.so bash -c "$(declare -f run); run")"; done
echo "default (rows centre-out): $(run)"
} | tee $out/ab2.txt
