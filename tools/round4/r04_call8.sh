#!/bin/bash
# round 4, call 8: where bin_faces spends 146 us on the knot scene (phase ticks of the prof build), and the row-order knob on raster_tile
set -u
out=gpurun_out/r04c8; mkdir -p $out
L=$(pwd)/kaolin_amd
KAMD_LIB_PATH=$L/libkaolin_amd_prof.so KAMD_PROF_SCENE=knot timeout 300 python tools/phase_prof.py 2>&1 | tail -30 | tee $out/phase_knot.txt | cut -c1-400
KAMD_LIB_PATH=$L/libkaolin_amd_prof.so timeout 300 python tools/phase_prof.py 2>&1 | head -3 | tee $out/phase_sphere.txt | cut -c1-400
