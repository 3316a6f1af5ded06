#!/bin/bash
# round 4, call 21: the single-launch voxelizer's knobs (fill workgroups per CU, per-slab gates vs one wait for the whole clear)
set -u
out=gpurun_out/r04c21; mkdir -p $out
L=$(pwd)/kaolin_amd
timeout 300 python -m pytest tests/test_voxelgrid.py -m gpu -x -q 2>&1 | tail -2 | tee $out/pytest.txt
for cfg in "KAMD_VOX_FUSED=2" "KAMD_VOX_FUSED=1" "KAMD_VOX_NFILL=4" "KAMD_VOX_NFILL=1" "KAMD_VOX_GATE=2" "KAMD_VOX_GATE=2 KAMD_VOX_NFILL=4" "KAMD_VOX_GATE=2 KAMD_VOX_NFILL=8"; do
  echo "== $cfg" | tee -a $out/time_vox.txt
  env $cfg KAMD_LIB_PATH=$L/libkaolin_amd_voxexp.so timeout 120 python tools/time_vox.py 2>&1 | grep -v amdgpu.ids | head -2 | tee -a $out/time_vox.txt
done
