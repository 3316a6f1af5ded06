#!/bin/bash
# raster_tile ablations (KAMD_RASTER_MODE bits) with the current kernel: plain rasterize (front faces) and the fused operator
set -u
out=gpurun_out/r02abl; mkdir -p $out
for m in 0 32 16 8 1 3 7; do
  echo "mode $m: $(KAMD_RASTER_MODE=$m timeout 120 python tools/exp_raster.py 2>/dev/null | grep 'front faces\|dibr_rast' | sed 's/  */ /g' | tr '\n' ' ' | cut -c1-400)"
done | tee $out/abl.txt
