#!/bin/bash
# round 5, call 37: point_to_mesh's sweep taking the query blocks from the end of the curve (a throw-away build: -DKAMD_TS_REVERSE at the
# slot computation of ts_sweep_kernel) -- the sizing measurement DESIGN section 8 asks for
set -u
repo=$(pwd); out=$repo/gpurun_out/r05am; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
{
for i in 1 2; do
echo "== product"; timeout 120 python tools/time_tridist.py 1000000 2>/dev/null | tail -2
echo "== reversed"; KAMD_LIB_PATH=${L}_tsrev.so timeout 120 python tools/time_tridist.py 1000000 2>/dev/null | tail -2
done
} > $out/ts_order_ab.txt 2>&1
cat $out/ts_order_ab.txt
