#!/bin/bash
# round 4, call 24: big faces mark the tiles of their rectangles (a bit per tile, a word per 64 tiles of a row): the other tiles of the view
# keep their background fast path
set -u
out=gpurun_out/r04c24; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_tile_order.py tests/test_full_size_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $out/pytest.txt
bash tools/round3/ab.sh head_sphere KAMD_LIB_PATH=$L/libkaolin_amd_head.so 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh new_sphere 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh head_knot KAMD_LIB_PATH=$L/libkaolin_amd_head.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh new_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh head_sphere KAMD_LIB_PATH=$L/libkaolin_amd_head.so 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh new_sphere 2>&1 | tee -a $out/ab.txt | cut -c1-200
