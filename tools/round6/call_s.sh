set -x
O=$PWD/gpurun_out/r06s
mkdir -p $O
repo=$PWD
python -m pytest tests/test_sided_distance.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "c3 or chamfer or batch8" >> $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for lib in base quadseq "" q33 q42 q52; do
  echo "== ${lib:-product(quad ring, 3 batched passes, 4 waves)}" >> $O/chamfer.txt
  KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd${lib:+_$lib}.so python tools/round6/chamfer_kernels.py >> $O/chamfer.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for lib in base ""; do
  export KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd${lib:+_$lib}.so
  tag=${lib:-product}
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --output-format csv -d $O/p1 -- python $repo/tools/round6/chamfer_kernels.py 10 > /dev/null 2>&1
  find $O/p1 -name '*counter_collection.csv' -exec cp {} $O/pmc_sq_$tag.csv \; ; rm -rf $O/p1
  timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/p2 -- python $repo/tools/round6/chamfer_kernels.py 10 > /dev/null 2>&1
  find $O/p2 -name '*counter_collection.csv' -exec cp {} $O/pmc_ta_$tag.csv \; ; rm -rf $O/p2
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_ATOMIC_sum TCC_ATOMIC_sum TCC_EA_WRREQ_sum --output-format csv -d $O/p3 -- python $repo/tools/round6/chamfer_kernels.py 10 > /dev/null 2>&1
  find $O/p3 -name '*counter_collection.csv' -exec cp {} $O/pmc_tcc_$tag.csv \; ; rm -rf $O/p3
  for k in sq ta tcc; do [ -f $O/pmc_${k}_$tag.csv ] && python $repo/tools/pmc_table.py $O/pmc_${k}_$tag.csv $O/pmc_${k}_$tag.txt "rocprofv3 --pmc ... -- python tools/round6/chamfer_kernels.py 10 ($tag)" > /dev/null 2>&1; rm -f $O/pmc_${k}_$tag.csv; done
done
cd $repo
grep -E "passed|failed|rc" $O/pytest.log; grep -v amdgpu.ids $O/chamfer.txt; ls $O
