#!/bin/bash
set -u
L=$(pwd)/kaolin_amd
for i in 1 2; do
bash tools/round3/ab.sh w8
bash tools/round3/ab.sh w7 KAMD_LIB_PATH=$L/libkaolin_amd_w7.so
bash tools/round3/ab.sh w7tail1 KAMD_LIB_PATH=$L/libkaolin_amd_w7tail1.so
done | cut -c1-160
