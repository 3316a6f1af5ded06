#!/bin/bash
set -u
L=$(pwd)/kaolin_amd
for i in 1 2; do
bash tools/round3/ab.sh new_occ6
bash tools/round3/ab.sh new_evalw7 KAMD_LIB_PATH=$L/libkaolin_amd_evalw7.so
bash tools/round3/ab.sh committed KAMD_LIB_PATH=$L/libkaolin_amd_committed.so
done
