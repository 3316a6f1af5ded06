#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [name filter]  -- registers / LDS / scratch per kernel (hipcc -Rpass-analysis)
cd "$(dirname "$0")/../kaolin_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-math-errno \
  -Rpass-analysis=kernel-resource-usage ${KR_DEFS:-} -c "$1" -o /dev/null 2>&1 | python3 -c "
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)', line)
    if m: cur={'name':m.group(1)}; continue
    for k in ('TotalSGPRs','VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)', line)
        if m and cur is not None:
            cur[k.split(' ')[0]]=m.group(1)
            if k.startswith('LDS'):
                print('%-60s sgpr %3s vgpr %3s scratch %3s occ %s lds %s' % (cur['name'][:60], cur.get('TotalSGPRs'), cur.get('VGPRs'), cur.get('ScratchSize'), cur.get('Occupancy'), cur.get('LDS')))
" | grep -E "${2:-.}"
