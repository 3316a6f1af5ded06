#!/bin/bash
set -u
export RI_MODES=0
for env in "KAMD_RASTER_FILL=2" "KAMD_RASTER_FILL_PACE=512" "KAMD_RASTER_FILL_PACE=514" "KAMD_RASTER_FILL_PACE=516" "KAMD_RASTER_FILL_PACE=520"; do
  echo "== $env"; env $env python tools/round3/raster_insts.py time 2>&1 | grep "mode 0"
done
