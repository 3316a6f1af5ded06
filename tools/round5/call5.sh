#!/bin/bash
# round 5, call 5: what bounds the rasterizer's tile kernel in its three forms (SQ counters), and what the single-round / the
# heavy tiles cost alone (timing-only diagnostic builds)
set -u
repo=$(pwd); out=$repo/gpurun_out/r05e; mkdir -p $out
L=$repo/kaolin_amd/libkaolin_amd
{
echo "== product (a workgroup per tile)"; timeout 100 python tools/round5/raster_fwd.py 20
for g in 3 4; do echo "== KAMD_RASTER_GEN=$g"; KAMD_LIB_PATH=${L}_exp.so KAMD_RASTER_GEN=$g timeout 100 python tools/round5/raster_fwd.py 20; done
echo "== diagnostic: wavefront per tile, tiles of more than one round leave at once (WRONG RESULTS)"; KAMD_LIB_PATH=${L}_r4light.so KAMD_RASTER_GEN=4 timeout 100 python tools/round5/raster_fwd.py 20
echo "== diagnostic: workgroup per tile, tiles of at most 48 faces leave after their scan (WRONG RESULTS)"; KAMD_LIB_PATH=${L}_r2heavy.so KAMD_RASTER_GEN=2 timeout 100 python tools/round5/raster_fwd.py 20
echo "== knot"; timeout 100 python tools/round5/raster_fwd.py 20 knot
echo "== knot diag light"; KAMD_LIB_PATH=${L}_r4light.so KAMD_RASTER_GEN=4 timeout 100 python tools/round5/raster_fwd.py 20 knot
echo "== knot diag heavy"; KAMD_LIB_PATH=${L}_r2heavy.so KAMD_RASTER_GEN=2 timeout 100 python tools/round5/raster_fwd.py 20 knot
} 2>&1 | grep -v amdgpu.ids > $out/raster_forms_time.txt
cat $out/raster_forms_time.txt
cd /tmp && export TMPDIR=/tmp
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
C2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
for g in 2 3 4; do
  for c in 1 2; do
    if [ $c = 1 ]; then CS=$C1; else CS=$C2; fi
    KAMD_LIB_PATH=${L}_exp.so KAMD_RASTER_GEN=$g timeout 120 rocprofv3 --pmc $CS --output-format csv -d $out/pmc -- python $repo/tools/round5/raster_fwd.py 3 > /dev/null 2>&1
    find $out/pmc -name '*counter_collection.csv' -exec cp {} $out/pmc_g${g}_c$c.csv \;
    rm -rf $out/pmc
    python $repo/tools/pmc_table.py $out/pmc_g${g}_c$c.csv $out/pmc_gen${g}_set$c.txt "KAMD_RASTER_GEN=$g rocprofv3 --pmc $CS -- python tools/round5/raster_fwd.py 3" > /dev/null 2>&1
    rm -f $out/pmc_g${g}_c$c.csv
  done
done
grep -h "raster\|columns" $out/pmc_gen*_set*.txt | cut -c1-400
