#!/bin/bash
# round 4, call 4: where soft_prep_kernel's 30 us go (timing-only ablation builds) + the fixed tests
set -u
out=gpurun_out/r04c4; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 600 python -m pytest tests/test_tile_order.py tests/test_distributed.py tests/test_dibr_gpu.py -m gpu -q --timeout 400 -rf > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log | cut -c1-300
bash tools/round3/ab.sh full 2>&1 | tee $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh abl1_exit_after_check KAMD_LIB_PATH=$L/libkaolin_amd_prepabl1.so 2>&1 | tee -a $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh abl2_exit_after_ordering KAMD_LIB_PATH=$L/libkaolin_amd_prepabl2.so 2>&1 | tee -a $out/ab.txt | cut -c1-360
bash tools/round3/ab.sh abl4_no_emit KAMD_LIB_PATH=$L/libkaolin_amd_prepabl4.so 2>&1 | tee -a $out/ab.txt | cut -c1-360
