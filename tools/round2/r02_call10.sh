#!/bin/bash
# Round 2, GPU call 10: chamfer with the value launch (tests + timing); phase profile incl. the binning kernel's wave durations
set -u
out=gpurun_out/r02n; mkdir -p $out
timeout 600 python -m pytest tests/test_sided_distance.py tests/test_graph_capture.py tests/test_full_size_parity.py -q -x -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -2 $out/pytest.log
KAMD_CHECK_SPLIT=1 timeout 180 python tools/check_chamfer.py 2>&1 | grep "forward\|plain\|step\|OK\|rror" | tee $out/chamfer.txt
KAMD_LIB_PATH=kaolin_amd/libkaolin_amd_prof.so timeout 300 python tools/phase_prof.py 2>&1 | grep -v Warn | tee $out/phase.txt
