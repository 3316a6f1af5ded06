#!/bin/bash
# round 6, call 5: phase ticks again with the profiler's own atomics spread over 2048 rows
set -u
repo=$(pwd); out=$repo/gpurun_out/r06e; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
KAMD_LIB_PATH=${L}_prof.so timeout 300 python tools/phase_prof.py > $out/raster_phases.txt 2>&1; cat $out/raster_phases.txt
KAMD_PROF_SCENE=knot KAMD_LIB_PATH=${L}_prof.so timeout 300 python tools/phase_prof.py > $out/raster_phases_knot.txt 2>&1; cat $out/raster_phases_knot.txt
