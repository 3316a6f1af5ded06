set -x
O=gpurun_out/r06q
mkdir -p $O
python -m pytest tests/test_mesh_to_spc.py tests/test_sided_distance.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "c3 or chamfer or batch8" >> $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for lib in base exp expA expB expC; do
  for sc in sphere knot; do
    echo "== $lib $sc" >> $O/step.txt
    KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_$lib.so python tools/round6/step_kernels.py 60 $sc >> $O/step.txt 2>&1
  done
done
for sc in sphere knot; do
  echo "== expA per_cu=8 $sc" >> $O/step.txt
  KAMD_SOFT_BWD_PER_CU=8 KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_expA.so python tools/round6/step_kernels.py 60 $sc >> $O/step.txt 2>&1
done
for lib in base sdgU1 exp sdgU8; do
  echo "== $lib" >> $O/chamfer.txt
  KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_$lib.so python tools/round6/chamfer_kernels.py >> $O/chamfer.txt 2>&1
done
for lib in base exp; do
  echo "== $lib" >> $O/spc.txt
  KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_$lib.so python tools/time_spc.py >> $O/spc.txt 2>&1
done
grep -E "passed|failed|rc" $O/pytest.log; grep -v amdgpu.ids $O/step.txt $O/chamfer.txt $O/spc.txt
