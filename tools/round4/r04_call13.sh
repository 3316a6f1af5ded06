#!/bin/bash
# round 4, call 13: the knot scene in parts, again (bowl alone, faces shuffled, bowl subdivided) -- which geometry costs which kernel
set -u
out=gpurun_out/r04c13; mkdir -p $out
timeout 300 python tools/round4/knot_parts.py 2>&1 | tee $out/knot_parts.txt | cut -c1-260
