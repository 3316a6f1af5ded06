#!/bin/bash
# round 4, call 15: medium faces binned a lane per FACE (each lane walks its own rectangle, 4 counter atomics in flight), pool x8
set -u
out=gpurun_out/r04c15; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_tile_order.py tests/test_full_size_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $out/pytest.txt
bash tools/round3/ab.sh head_sphere KAMD_LIB_PATH=$L/libkaolin_amd_head.so 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh new_sphere 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh new_sphere_med9 KAMD_LIB_PATH=$L/libkaolin_amd_med9.so 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh new_sphere_med16 KAMD_LIB_PATH=$L/libkaolin_amd_med16.so 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh head_knot KAMD_LIB_PATH=$L/libkaolin_amd_head.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh new_knot -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh new_knot_med9 KAMD_LIB_PATH=$L/libkaolin_amd_med9.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-220
bash tools/round3/ab.sh new_knot_med16 KAMD_LIB_PATH=$L/libkaolin_amd_med16.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-220
timeout 300 python tools/round4/knot_parts.py 2>&1 | tee $out/knot_parts.txt | cut -c1-260
