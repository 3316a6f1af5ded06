#!/bin/bash
set -u
L=$(pwd)/kaolin_amd
bash tools/round3/ab.sh base
for v in abl4 abl8 abl11nf; do bash tools/round3/ab.sh $v KAMD_LIB_PATH=$L/libkaolin_amd_$v.so; done
bash tools/round3/ab.sh abl4_per_tile KAMD_BWD_COV_LIST=2 KAMD_LIB_PATH=$L/libkaolin_amd_abl4.so
bash tools/round3/ab.sh abl4_per_cu2 KAMD_RBWD_PER_CU=2 KAMD_LIB_PATH=$L/libkaolin_amd_abl4.so
