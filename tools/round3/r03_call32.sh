#!/bin/bash
# the two backward kernels side by side again: both are persistent grids now and neither needs all the wave slots
set -u
for i in 1 2; do
bash tools/round3/ab.sh one_stream
bash tools/round3/ab.sh side_stream KAMD_BWD_SIDE_STREAM=1
bash tools/round3/ab.sh side_rb4_fb8 KAMD_BWD_SIDE_STREAM=1 KAMD_RBWD_PER_CU=4 KAMD_SOFT_BWD_PER_CU=8
bash tools/round3/ab.sh side_rb4_fb12 KAMD_BWD_SIDE_STREAM=1 KAMD_RBWD_PER_CU=4 KAMD_SOFT_BWD_PER_CU=12
bash tools/round3/ab.sh side_rb3_fb8 KAMD_BWD_SIDE_STREAM=1 KAMD_RBWD_PER_CU=3 KAMD_SOFT_BWD_PER_CU=8
bash tools/round3/ab.sh side_rb2_fb16 KAMD_BWD_SIDE_STREAM=1 KAMD_RBWD_PER_CU=2 KAMD_SOFT_BWD_PER_CU=16
done | cut -c1-110
