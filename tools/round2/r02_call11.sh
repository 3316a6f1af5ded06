#!/bin/bash
set -u
out=gpurun_out/r02o; mkdir -p $out
KAMD_CHECK_SPLIT=1 timeout 180 python tools/check_chamfer.py 2>&1 | grep "forward\|plain\|step\|OK\|rror" | tee $out/chamfer.txt
KAMD_LIB_PATH=kaolin_amd/libkaolin_amd_prof.so timeout 300 python tools/phase_prof.py 2>&1 | grep "bin_faces" | tee $out/phase.txt
timeout 300 python -m pytest tests/test_sided_distance.py tests/test_graph_capture.py -q -x -m gpu --timeout 300 2>&1 | tail -1
