#!/bin/bash
# round 4, call 29: select grid 128 per CU as the default, both scenes, against the experiment build at 32 on the same box
set -u
out=gpurun_out/r04c29; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_graph_capture.py -m gpu -x -q 2>&1 | tail -2 | tee $out/pytest.txt
for rep in 1 2; do
bash tools/round3/ab.sh sphere_32 KAMD_LIB_PATH=$L/libkaolin_amd_exp.so KAMD_SOFT_SELECT_PER_CU=32 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh sphere_128_default 2>&1 | tee -a $out/ab.txt | cut -c1-200
done
bash tools/round3/ab.sh knot_32 KAMD_LIB_PATH=$L/libkaolin_amd_exp.so KAMD_SOFT_SELECT_PER_CU=32 -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh knot_128_default -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
