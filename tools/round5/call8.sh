#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r05i; mkdir -p $out
KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd_prof.so timeout 200 python tools/round5/ts_phases.py 2>&1 | grep -v amdgpu.ids > $out/ts_phases.txt; cat $out/ts_phases.txt
