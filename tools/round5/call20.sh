#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r05w; mkdir -p $out
KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd_exp.so KAMD_TS_STATS=1 timeout 300 python tools/round5/degenerate_repro.py 2>&1 | grep -v amdgpu.ids > $out/degenerate_repro.txt; cat $out/degenerate_repro.txt | cut -c1-420
