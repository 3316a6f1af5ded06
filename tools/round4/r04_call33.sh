#!/bin/bash
# round 4, call 33: the final library's quick bench lines on both scenes (the evidence call before landed on a box whose clocks were ~30 % lower)
set -u
out=gpurun_out/r04c33; mkdir -p $out
bash tools/round3/ab.sh final_sphere 2>&1 | tee -a $out/ab.txt | cut -c1-230
bash tools/round3/ab.sh final_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-230
