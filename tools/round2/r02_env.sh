#!/bin/bash
# A/B of one environment knob through the bench's DIB-R section: r02_env.sh NAME VALUE [VALUE...]
set -u
name=$1; shift
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-chamfer --no-c5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.load(sys.stdin); print(j['ms_per_step'], j['per_step_ms']['median'], j['per_step_ms']['min'])"; }
echo "default: $(run)"
for v in "$@"; do echo "$name=$v: $(env $name=$v bash -c "$(declare -f run); run")"; done
echo "default: $(run)"
