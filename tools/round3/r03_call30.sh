#!/bin/bash
set -u
L=$(pwd)/kaolin_amd
bash tools/round3/ab.sh base
for v in ev1 ev2 ev4 ev5; do bash tools/round3/ab.sh $v KAMD_LIB_PATH=$L/libkaolin_amd_$v.so; done
