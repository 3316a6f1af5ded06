#!/bin/bash
set -u
bash tools/round3/ab.sh base
for v in 24 28 48 64; do bash tools/round3/ab.sh eval_per_cu_$v KAMD_SOFT_EVAL_PER_CU=$v; done
for v in 24 28 48 64; do bash tools/round3/ab.sh select_per_cu_$v KAMD_SOFT_SELECT_PER_CU=$v; done
bash tools/round3/ab.sh base_again
