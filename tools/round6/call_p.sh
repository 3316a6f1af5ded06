set -x
mkdir -p gpurun_out/r06p
O=gpurun_out/r06p
python -m pytest tests/test_sided_distance.py tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_full_size_parity.py tests/test_render_fused.py tests/test_graph_capture.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for lib in base exp; do
  for sc in sphere knot knot_shuffled; do
    echo "== $lib $sc" >> $O/step.txt
    KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_$lib.so python tools/round6/step_kernels.py 60 $sc >> $O/step.txt 2>&1
  done
  echo "== $lib" >> $O/chamfer.txt
  KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_$lib.so python tools/round6/chamfer_kernels.py >> $O/chamfer.txt 2>&1
done
for pc in 8 4; do
  for sc in sphere knot; do
    echo "== exp per_cu=$pc $sc" >> $O/step.txt
    KAMD_SOFT_BWD_PER_CU=$pc KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_exp.so python tools/round6/step_kernels.py 60 $sc >> $O/step.txt 2>&1
  done
done
tail -3 $O/pytest.log; cat $O/step.txt $O/chamfer.txt
