#!/bin/bash
# round 3, call 1: the new parity tests + RCCL one-rank tests, the reference's tests through the alias AND through the reference's own
# Python layer over kaolin_amd._C, the K7 contraction A/B, the bench line (contract operators, graph replay), host profile
set -u
out=gpurun_out/r03a; mkdir -p $out
timeout 420 python -m pytest tests -m gpu -q -x --durations=12 --timeout 280 > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
bash tools/run_reference_tests.sh r03a_alias > /dev/null 2>&1; tail -2 gpurun_out/r03a_alias/reference_tests.log
KAMD_REF_LAYER=1 bash tools/run_reference_tests.sh r03a_reflayer > /dev/null 2>&1; tail -2 gpurun_out/r03a_reflayer/reference_tests.log
timeout 200 python tools/k7_contraction_ab.py --seeds 200 > $out/k7_off.json 2> $out/k7_off.err; tail -c 900 $out/k7_off.json
KAMD_LIB_PATH=$(pwd)/kaolin_amd/libkaolin_amd_k7fma.so timeout 200 python tools/k7_contraction_ab.py --seeds 200 > $out/k7_fma.json 2> $out/k7_fma.err; tail -c 900 $out/k7_fma.json
timeout 280 python bench.py 2> $out/bench.err | tail -1 > $out/bench.json
python - $out/bench.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print('  ms/step', j['ms_per_step'], j['per_step_ms'], 'roofline', j['roofline'])
    print('  ', {k.replace('_kernel', ''): v['avg_us'] for k, v in j['kernels'].items()})
    print('  host', j['host_enqueue_ms_per_step'], 'graph', j['graph_replay'], 'dist', j['distributed'])
    print('  contract', json.dumps(j['contract_operators']))
    print('  chamfer', j['chamfer']['ms_per_step'], j['chamfer']['operator_only'], 'c5', j['c5']['voxelgrid_256_us'], j['c5']['point_to_mesh_1Mx50k_ms'])
    print('  cpu', json.dumps(j['cpu_baseline'])[:1500])
except Exception as e:
    print('bench failed', e)
PY
tail -5 $out/bench.err
timeout 120 python tools/host_profile_dibr.py > $out/host_profile_dibr.txt 2>&1; head -4 $out/host_profile_dibr.txt
