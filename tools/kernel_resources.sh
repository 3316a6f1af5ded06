#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [name filter]  -- registers / LDS / scratch per kernel (hipcc -Rpass-analysis)
cd "$(dirname "$0")/../kaolin_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-math-errno \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | grep -E "Function Name|SGPRs:|VGPRs:|Occupancy|LDS Size|ScratchSize" \
  | sed 's/.*remark: [^ ]* //; s/ \[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - - - | sed 's/Function Name: //; s/  */ /g' | grep -E "${2:-.}"
