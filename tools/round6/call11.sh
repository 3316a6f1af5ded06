#!/bin/bash
# round 6, call 11: the timeline of one raster_tile launch (per-tile begin / end stamps, profiling build)
set -u
repo=$(pwd); out=$repo/gpurun_out/r06k; mkdir -p $out
KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd_prof.so timeout 300 python tools/round6/tile_timeline.py sphere > $out/tile_timeline_sphere.txt 2>&1; cat $out/tile_timeline_sphere.txt
