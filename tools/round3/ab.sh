#!/bin/bash
# usage: bash tools/round3/ab.sh LABEL [VAR=VALUE ...] [-- bench flags]   -- one quick bench line (headline step + kernel table) under an environment
label=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
env "${envs[@]}" timeout 150 python bench.py --quick --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read())
    k={n.replace('_kernel',''): v['avg_us'] for n, v in j['kernels'].items()}
    print('%-34s ms/step %.4f median %.4f  host %.3f | ' % ('$label', j['ms_per_step'], j['per_step_ms']['median'], j['host_enqueue_ms_per_step']) + ' '.join('%s %.1f' % (n, v) for n, v in k.items()))
except Exception as e:
    print('$label FAILED', e)
"
