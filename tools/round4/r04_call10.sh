#!/bin/bash
# round 4, call 10: which part of round 3's binning kernel is slow on the knot scene (its timing-only ablation builds), and a full bench line
set -u
out=gpurun_out/r04c10; mkdir -p $out
L=$(pwd)/kaolin_amd
bash tools/round3/ab.sh r03_knot KAMD_LIB_PATH=$L/libkaolin_amd_r03.so -- --scene knot 2>&1 | tee $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh r03_knot_no_raster_lists KAMD_LIB_PATH=$L/libkaolin_amd_r03abl1.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh r03_knot_no_soft_lists KAMD_LIB_PATH=$L/libkaolin_amd_r03abl2.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh r03_knot_no_lists KAMD_LIB_PATH=$L/libkaolin_amd_r03abl3.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
bash tools/round3/ab.sh r03_knot_no_lists_no_records KAMD_LIB_PATH=$L/libkaolin_amd_r03abl11.so -- --scene knot 2>&1 | tee -a $out/ab.txt | cut -c1-200
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json; tail -3 $out/bench.err
