#!/bin/bash
set -u
for t in 14 16 18 20 22; do echo "== KAMD_VOX_LOG2_THREADS=$t"; KAMD_VOX_LOG2_THREADS=$t timeout 120 python tools/time_vox.py 2>&1 | grep -v Warn | tail -6; done | tee gpurun_out/r02_vox.txt
