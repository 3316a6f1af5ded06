#!/bin/bash
# round 4, call 22: selection + evaluation of the soft mask in ONE launch (the selecting wavefront evaluates its item's pairs)
set -u
out=gpurun_out/r04c22; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $out/pytest.txt
bash tools/round3/ab.sh sphere_unfused KAMD_LIB_PATH=$L/libkaolin_amd_fuse5.so KAMD_SOFT_FUSE=2 2>&1 | tee -a $out/ab.txt | cut -c1-230
bash tools/round3/ab.sh sphere_fused_w4 2>&1 | tee -a $out/ab.txt | cut -c1-230
bash tools/round3/ab.sh sphere_fused_w5 KAMD_LIB_PATH=$L/libkaolin_amd_fuse5.so 2>&1 | tee -a $out/ab.txt | cut -c1-230
bash tools/round3/ab.sh sphere_fused_w3 KAMD_LIB_PATH=$L/libkaolin_amd_fuse3.so 2>&1 | tee -a $out/ab.txt | cut -c1-230
bash tools/round3/ab.sh knot_unfused KAMD_LIB_PATH=$L/libkaolin_amd_fuse5.so KAMD_SOFT_FUSE=2 -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-230
bash tools/round3/ab.sh knot_fused_w4 -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-230
bash tools/round3/ab.sh knot_fused_w5 KAMD_LIB_PATH=$L/libkaolin_amd_fuse5.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-230
