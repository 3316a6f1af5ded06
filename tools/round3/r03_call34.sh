#!/bin/bash
set -u
timeout 500 python -m pytest tests/test_deftet.py -m gpu -q -x --timeout 280 2>&1 | tail -2
echo "== staged"; python tools/time_deftet.py 2>&1 | grep -v amdgpu | head -8 | cut -c1-260
echo "== lane per value"; KAMD_DEFTET_BWD_STAGED=2 python tools/time_deftet.py 2>&1 | grep -v amdgpu | head -8 | cut -c1-260
