O=$PWD/gpurun_out/r06pf; mkdir -p $O
timeout 900 python -m pytest tests/test_dibr_gpu.py tests/test_dibr_fuzz.py tests/test_render_fused.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python -m pytest tests/test_full_size_parity.py -m gpu -x -q -k "c4 or knot or c2 or soft or k_buffers" >> $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for rep in 1 2; do for sc in sphere knot knot_shuffled; do
for v in base pf pf4 pf2; do
echo "== $sc $v" >> $O/step.txt
KAMD_LIB_PATH=$PWD/kaolin_amd/libkaolin_amd_$v.so timeout 300 python tools/round6/step_kernels.py 100 $sc >> $O/step.txt 2>&1
done; done; done
grep -E "passed|failed|rc" $O/pytest.log | head; grep -v amdgpu.ids $O/step.txt | sed -e 's/bin_faces.*soft_select_kernel/soft_select_kernel/' | cut -c1-150
