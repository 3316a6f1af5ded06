#!/bin/bash
set -u
L=$(pwd)/kaolin_amd
timeout 500 python -m pytest tests/test_sided_distance.py tests/test_full_size_parity.py -m gpu -q -x --timeout 280 2>&1 | tail -2
for i in 1 2; do
AB_LABEL=new python tools/round3/chamfer_ab.py 2>&1 | grep -v amdgpu | tail -1
AB_LABEL=committed KAMD_LIB_PATH=$L/libkaolin_amd_committed.so python tools/round3/chamfer_ab.py 2>&1 | grep -v amdgpu | tail -1
done
