set -x
O=gpurun_out/r06r
mkdir -p $O
python -m pytest tests/test_sided_distance.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "c3 or chamfer or batch8" >> $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for lib in base bu1 qu4 bu8 ""; do
  echo "== ${lib:-product(quad ring, build unroll 4, box words on own lines, two-level barrier)}" >> $O/chamfer.txt
  KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd${lib:+_$lib}.so python tools/round6/chamfer_kernels.py >> $O/chamfer.txt 2>&1
done
grep -E "passed|failed|rc" $O/pytest.log; grep -v amdgpu.ids $O/chamfer.txt
