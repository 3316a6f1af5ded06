#!/bin/bash
# round 5, call 14: the randomised parity sweeps on the round's library, from seeds the suite does not use
set -u
repo=$(pwd); out=$repo/gpurun_out/r05p; mkdir -p $out
{
timeout 600 python tools/round4/fuzz_tridist.py 240 1000 2>&1 | tail -5
timeout 600 python tools/round4/fuzz_sided.py 200 500 2>&1 | tail -5
timeout 600 python tools/round4/fuzz_dibr.py 100 900 2>&1 | tail -4
timeout 400 python tools/round4/fuzz_others.py 40 300 2>&1 | tail -6
} | grep -v amdgpu.ids > $out/fuzz_round5.txt
cat $out/fuzz_round5.txt
