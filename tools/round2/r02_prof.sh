#!/bin/bash
# rocprofv3 kernel trace + SQ counter pass of tools/prof_dibr.py -> gpurun_out/$1/
set -u
tag=${1:-r02p}; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python $repo/tools/prof_dibr.py 5 > $out/prof_dibr.txt 2>&1
find $out/kt -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/kt
if [ "${2:-}" = "pmc" ]; then
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc1 -- python $repo/tools/prof_dibr.py 2 > /dev/null 2>&1
  find $out/pmc1 -name '*counter_collection.csv' -exec cp {} $out/pmc_sq1.csv \;
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $out/pmc2 -- python $repo/tools/prof_dibr.py 2 > /dev/null 2>&1
  find $out/pmc2 -name '*counter_collection.csv' -exec cp {} $out/pmc_sq2.csv \;
  rm -rf $out/pmc1 $out/pmc2
fi
cd $repo
tail -2 $out/prof_dibr.txt
python tools/summarize_prof.py $out/kernel_stats.csv $out/kernel_stats.txt "$tag" | head -40
if [ -f $out/pmc_sq1.csv ]; then python tools/pmc_summary.py $out/pmc_sq1.csv raster_tile raster_backward soft_search soft_mask_backward_list bin_faces bin_scan; python tools/pmc_summary.py $out/pmc_sq2.csv raster_tile raster_backward soft_search soft_mask_backward_list bin_faces bin_scan; fi
