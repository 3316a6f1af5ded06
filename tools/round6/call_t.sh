set -x
O=$PWD/gpurun_out/r06t
mkdir -p $O
repo=$PWD
python -m pytest tests/test_sided_distance.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "c3 or chamfer or batch8" >> $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for lib in base x0 ""; do
  echo "== ${lib:-product(quad ring, XCD-contiguous chunks)}" >> $O/chamfer.txt
  KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd${lib:+_$lib}.so python tools/round6/chamfer_kernels.py >> $O/chamfer.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for lib in x0 ""; do
  export KAMD_LIB_PATH=$repo/kaolin_amd/libkaolin_amd${lib:+_$lib}.so
  tag=${lib:-product}
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --output-format csv -d $O/p3 -- python $repo/tools/round6/chamfer_kernels.py 10 > /dev/null 2>&1
  find $O/p3 -name '*counter_collection.csv' -exec cp {} $O/pmc_tcc_$tag.csv \; ; rm -rf $O/p3
  python $repo/tools/pmc_table.py $O/pmc_tcc_$tag.csv $O/pmc_tcc_$tag.txt "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum -- python tools/round6/chamfer_kernels.py 10 ($tag)" > /dev/null 2>&1; rm -f $O/pmc_tcc_$tag.csv
done
cd $repo
grep -E "passed|failed|rc" $O/pytest.log; grep -v amdgpu.ids $O/chamfer.txt; grep "sdg_query" $O/pmc_tcc_*.txt | cut -c1-200
