set -x
O=$PWD/gpurun_out/r06u
mkdir -p $O
python -m pytest tests/test_sided_distance.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "c3 or chamfer or batch8" >> $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for lib in base ""; do
  echo "== ${lib:-product(quad ring macro form, XCD-contiguous chunks)}" >> $O/chamfer.txt
  KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd${lib:+_$lib}.so python tools/round6/chamfer_kernels.py >> $O/chamfer.txt 2>&1
done
KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_pt.so python tools/round6/chamfer_kernels.py 4 > $O/phases.txt 2>&1
grep -E "passed|failed|rc" $O/pytest.log; grep -v amdgpu.ids $O/chamfer.txt; grep sdg_build $O/phases.txt | sort | uniq -c | sort -rn | head -5; grep "sdg_build wg" $O/phases.txt | sed -n '8,14p;30,36p'
