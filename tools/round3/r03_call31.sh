#!/bin/bash
set -u
timeout 300 python -m pytest tests/test_dibr_gpu.py -m gpu -q -x --timeout 280 2>&1 | tail -1
bash tools/round3/ab.sh prefetch_per_cu_32
for v in 6 7 8 12 16 24; do bash tools/round3/ab.sh prefetch_per_cu_$v KAMD_SOFT_EVAL_PER_CU=$v; done
bash tools/round3/ab.sh prefetch_per_cu_32_again
