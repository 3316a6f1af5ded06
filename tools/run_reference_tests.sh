#!/bin/bash
# Runs the staged reference tests (tools/stage_reference_tests.sh) unchanged against kaolin_amd; log -> gpurun_out/<tag>/
# KAMD_REF_LAYER=1 in the environment: the reference's own Python layer (its autograd Functions) over kaolin_amd._C
tag=${1:-r02_reftests}; out=$(pwd)/gpurun_out/$tag; mkdir -p $out
cd _ref_tests || exit 1
timeout 1500 python -m pytest tests/python/kaolin -q -p no:cacheprovider --import-mode=importlib ${2:-} 2>&1 > $out/reference_tests_full.log; (grep -E "^(FAILED|ERROR)" $out/reference_tests_full.log | sed "s/\[.*//" | sort | uniq -c | sort -rn | head -40; tail -3 $out/reference_tests_full.log) > $out/reference_tests.log
tail -40 $out/reference_tests.log
