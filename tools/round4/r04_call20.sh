#!/bin/bash
# round 4, call 20: the voxelizer as ONE launch (fill and marking roles, a flag per cleared slab): stress test, then timing both forms
set -u
out=gpurun_out/r04c20; mkdir -p $out
timeout 300 python -m pytest tests/test_voxelgrid.py -m gpu -x -q 2>&1 | tail -3 | tee $out/pytest.txt
for f in 1 2; do
  echo "KAMD_VOX_FUSED=$f" | tee -a $out/time_vox.txt
  KAMD_VOX_FUSED=$f timeout 120 python tools/time_vox.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $out/time_vox.txt
done
