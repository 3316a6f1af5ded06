#!/bin/bash
set -u
tag=r03v; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_dibr_gpu.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -q -x --timeout 280 2>&1 | tail -2
L=$repo/kaolin_amd
export RI_MODES=0
echo "== new"; python tools/round3/raster_insts.py time 2>&1 | grep mode
echo "== committed"; KAMD_LIB_PATH=$L/libkaolin_amd_committed.so python tools/round3/raster_insts.py time 2>&1 | grep mode
for i in 1 2; do
bash tools/round3/ab.sh new
bash tools/round3/ab.sh committed KAMD_LIB_PATH=$L/libkaolin_amd_committed.so
done | cut -c1-170
RI_MODES=0 bash tools/round3/r03_raster_insts.sh r03v > /dev/null 2>&1; cat $out/raster_modes_pmc.txt
