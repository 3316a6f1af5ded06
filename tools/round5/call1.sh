#!/bin/bash
# round 5, call 1: the re-seed rule (K5 / K7), the own counting sort + XCD mapping of the sweep, bench --gpus self-launch
set -u
repo=$(pwd); out=$repo/gpurun_out/r05a; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x --durations=12 --timeout 600 > $out/pytest_gpu.log 2>&1; tail -30 $out/pytest_gpu.log
timeout 200 python tools/time_tridist.py 100000 1000000 > $out/time_tridist.txt 2>&1; cat $out/time_tridist.txt
timeout 200 python tools/time_chamfer.py > $out/time_chamfer.txt 2>&1; cat $out/time_chamfer.txt
timeout 300 python bench.py --no-cpu-baseline --no-contract-ops --no-scene-variants 2> $out/bench.err | tail -1 > $out/bench.json; cut -c1-1500 $out/bench.json
