#!/bin/bash
# Round 2, GPU call 1: full-size oracle parity tests, the baseline bench of the round-1 kernels on this box, and the grid
# sweeps that settle commit 6fc71de (occupancy-query grids of the persistent kernels).  Output -> gpurun_out/r02a/.
set -u
repo=$(pwd); out=$repo/gpurun_out/r02a; mkdir -p $out
timeout 900 python -m pytest tests/test_full_size_parity.py -q -x > $out/pytest_full_size.log 2>&1; tail -3 $out/pytest_full_size.log
KAMD_VERBOSE=1 timeout 600 python bench.py 2> $out/bench.err | tail -1 > $out/bench.json
grep kamd $out/bench.err | sort | uniq -c
python - <<'PY' > $out/sweep.txt 2>&1
import json, os, subprocess, sys
def run(env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-chamfer', '--no-c5'],
                       capture_output=True, text=True, env=e, timeout=300)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        k = j['kernels']
        return j['ms_per_step'], {n.replace('_kernel', ''): k[n]['avg_us'] for n in k}
    except Exception as ex:
        return None, r.stderr[-300:]
for p in (8, 12, 16, 24, 32):
    print('SOFT_SEARCH_PER_CU', p, *run({'KAMD_SOFT_SEARCH_PER_CU': str(p)}), flush=True)
for p in (6, 10, 16):
    print('SOFT_BWD_PER_CU', p, *run({'KAMD_SOFT_BWD_PER_CU': str(p)}), flush=True)
PY
cat $out/sweep.txt
for p in 4 6 7 8 12; do echo "TS_HARD_PER_CU $p"; KAMD_TS_HARD_PER_CU=$p timeout 120 python tools/time_tridist.py 2>&1 | grep -v Warn | tail -2; done > $out/ts_hard_sweep.txt 2>&1
cat $out/ts_hard_sweep.txt
cut -c1-600 $out/bench.json
