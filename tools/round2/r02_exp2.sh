for m in 0 1 3 4 7; do echo "KAMD_RASTER_MODE=$m"; KAMD_RASTER_MODE=$m timeout 200 python tools/exp_raster.py 2>&1 | grep -E "front faces|dibr_raster"; done
