#!/bin/bash
# round 4: the reference's own test files for the path, unchanged, on the final build -- through install_as_kaolin() and through the
# reference's own Python layer over kaolin_amd._C -- and its test_triangle_distance over 200 seeds
set -u
out=gpurun_out/r04_reftests; mkdir -p $out
bash tools/run_reference_tests.sh r04_alias > /dev/null 2>&1; tail -3 gpurun_out/r04_alias/reference_tests.log | tee $out/summary.txt
KAMD_REF_LAYER=1 bash tools/run_reference_tests.sh r04_reflayer > /dev/null 2>&1; tail -3 gpurun_out/r04_reflayer/reference_tests.log | tee -a $out/summary.txt
timeout 400 python tools/k7_contraction_ab.py --seeds 200 > $out/k7_200_seeds.json 2> $out/k7.err; tail -c 900 $out/k7_200_seeds.json | tee -a $out/summary.txt
