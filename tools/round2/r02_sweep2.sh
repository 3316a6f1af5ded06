#!/bin/bash
# grid sweeps of the persistent soft-mask kernels after this round's changes -> gpurun_out/r02w/sweep.txt
set -u
out=gpurun_out/r02w; mkdir -p $out
python - <<'PY' > $out/sweep.txt 2>&1
import json, os, subprocess, sys
def run(env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-chamfer', '--no-c5'],
                       capture_output=True, text=True, env=e, timeout=300)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        k = j['kernels']
        return j['per_step_ms']['median'], {n.replace('_kernel', ''): k[n]['avg_us'] for n in ('soft_select_kernel', 'soft_eval_kernel', 'soft_mask_backward_list_kernel')}
    except Exception as ex:
        return None, r.stderr[-300:]
print('default', *run({}), flush=True)
for p in (5, 6, 7, 12, 14, 16, 21, 24):
    print('SOFT_BWD_PER_CU', p, *run({'KAMD_SOFT_BWD_PER_CU': str(p)}), flush=True)
for p in (16, 24, 40, 48, 64):
    print('SOFT_SELECT_PER_CU', p, *run({'KAMD_SOFT_SELECT_PER_CU': str(p)}), flush=True)
for p in (8, 16, 24, 48):
    print('SOFT_EVAL_PER_CU', p, *run({'KAMD_SOFT_EVAL_PER_CU': str(p)}), flush=True)
PY
cat $out/sweep.txt
