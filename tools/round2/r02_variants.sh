#!/bin/bash
# build-knob variants of the library (make -C kaolin_amd/csrc variant NAME=.. DEFS=..) through the bench -> gpurun_out/r02x/
set -u
out=gpurun_out/r02x; mkdir -p $out
python - "$@" <<'PY' > $out/variants.txt 2>&1
import json, os, subprocess, sys
def run(lib, extra):
    e = dict(os.environ)
    if lib: e['KAMD_LIB_PATH'] = os.path.abspath(f'kaolin_amd/libkaolin_amd_{lib}.so')
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-c5'] + extra,
                       capture_output=True, text=True, env=e, timeout=300)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        k = j['kernels']
        ch = j.get('chamfer') or {}
        return j['per_step_ms']['median'], {n.replace('_kernel', ''): k[n]['avg_us'] for n in k}, ch.get('ms_per_step'), ch.get('kernels_avg_us')
    except Exception as ex:
        return None, r.stderr[-400:]
for lib in [''] + sys.argv[1:]:
    extra = [] if lib.startswith('sdg') or lib == '' else ['--no-chamfer']
    print(lib or 'default', *run(lib, extra), flush=True)
PY
cat $out/variants.txt
