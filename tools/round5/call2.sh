#!/bin/bash
# round 5, call 2: the sweep's block order over the XCDs and the face test's form (point_to_mesh 1M x 50k)
set -u
repo=$(pwd); out=$repo/gpurun_out/r05b; mkdir -p $out
run() { echo "== $*"; env "$@" timeout 120 python tools/time_tridist.py 100000 1000000 2>&1 | grep -v amdgpu.ids; }
{
run KAMD_X=product
run KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd_branchy.so
run KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd_branchy.so KAMD_TS_XCD_CHUNK=9999
for c in 1 2 5 9 17 33 65 129 9999; do run KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd_exp.so KAMD_TS_XCD_CHUNK=$c; done
} > $out/ts_xcd_ab.txt 2>&1
cat $out/ts_xcd_ab.txt
