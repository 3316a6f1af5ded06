#!/bin/bash
# round 4, call 27: the select kernel on the knot scene -- grid size and occupancy (is an item's latency or the number of resident items the limit?)
set -u
out=gpurun_out/r04c27; mkdir -p $out
L=$(pwd)/kaolin_amd
for pc in 16 32 64 128; do
bash tools/round3/ab.sh knot_select_per_cu_$pc KAMD_LIB_PATH=$L/libkaolin_amd_exp.so KAMD_SOFT_SELECT_PER_CU=$pc -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
for v in selw4 selw6; do
bash tools/round3/ab.sh knot_$v KAMD_LIB_PATH=$L/libkaolin_amd_$v.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh sphere_$v KAMD_LIB_PATH=$L/libkaolin_amd_$v.so 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
