#!/bin/bash
# One call on the GPU box that regenerates everything kept under profiles/ for a round:
#   bash tools/round_profile.sh <tag> [pmc] [commit]   (commit: `git rev-parse --short HEAD` of the library, recorded in traffic.json's _source --
#   the GPU box has no .git)
# writes gpurun_out/<tag>/{pytest_gpu.log, bench.json, bench_under_rocprof.json, kernel_stats.csv,
# pmc_fetch.csv, pmc_write.csv}.  PMC passes run on their own (never combined with a trace domain).
set -u
tag=${1:-r01x}; pmc=${2:-}; commit=${3:-unknown}
repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=15 --timeout 1500 > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
if [ -n "$pmc" ]; then
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python $repo/tools/pmc_traffic.py > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python $repo/tools/pmc_traffic.py > /dev/null 2>&1
  find $out/pmc_f -name '*counter_collection.csv' -exec cp {} $out/pmc_fetch.csv \;
  find $out/pmc_w -name '*counter_collection.csv' -exec cp {} $out/pmc_write.csv \;
  rm -rf $out/pmc_f $out/pmc_w
  # SQ counters quoted in DESIGN (wave occupancy, VALU / SALU / LDS instruction and wait counts), two passes of 8 counters
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc1 -- python $repo/tools/pmc_traffic.py > /dev/null 2>&1
  find $out/pmc1 -name '*counter_collection.csv' -exec cp {} $out/pmc_sq1.csv \;
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $out/pmc2 -- python $repo/tools/pmc_traffic.py > /dev/null 2>&1
  find $out/pmc2 -name '*counter_collection.csv' -exec cp {} $out/pmc_sq2.csv \;
  rm -rf $out/pmc1 $out/pmc2
  # the tables kept under profiles/ (traffic.json: calibrated HBM bytes per launch of every kernel, the DIB-R step, the chamfer step /
  # operator and config C5; raw per-kernel tables of the four passes)
  python $repo/tools/parse_traffic.py $out/pmc_fetch.csv $out/pmc_write.csv $out/traffic.json "tools/round_profile.sh $tag pmc (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_traffic.py), library built from commit $commit" > /dev/null 2> $out/parse_traffic.err
  python $repo/tools/pmc_table.py $out/pmc_fetch.csv $out/pmc_FETCH_SIZE.txt "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python tools/pmc_traffic.py" > /dev/null 2>&1
  python $repo/tools/pmc_table.py $out/pmc_write.csv $out/pmc_WRITE_SIZE.txt "rocprofv3 --pmc WRITE_SIZE --output-format csv -- python tools/pmc_traffic.py" > /dev/null 2>&1
  python $repo/tools/pmc_table.py $out/pmc_sq1.csv $out/pmc_SQ_waves_insts.txt "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -- python tools/pmc_traffic.py" > /dev/null 2>&1
  python $repo/tools/pmc_table.py $out/pmc_sq2.csv $out/pmc_SQ_lds_vmem.txt "rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -- python tools/pmc_traffic.py" > /dev/null 2>&1
  python $repo/tools/issue_table.py $out/pmc_SQ_waves_insts.txt $out/pmc_SQ_lds_vmem.txt $out/traffic.json $out/issue_table.txt > /dev/null 2>&1
  rm -f $out/pmc_fetch.csv $out/pmc_write.csv $out/pmc_sq1.csv $out/pmc_sq2.csv   # (tens of MB; the tables above are what is kept)
fi
# the bench line and the kernel trace come AFTER the counters: bench.py's roofline.traffic reads profiles/traffic.json, which must be
# THIS library's (the committed line cited the table of an earlier commit)
if [ -s $out/traffic.json ]; then cp $out/traffic.json $repo/profiles/traffic.json; fi
cd $repo
timeout 600 python bench.py 2> $out/bench.err | tail -1 > $out/bench.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $repo/bench.py --no-cpu-baseline --no-contract-ops --no-scene-variants 2> $out/prof.err | tail -1 > $out/bench_under_rocprof.json
find $out/prof -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/prof
cd $repo; ls -la $out; cat $out/bench.json | cut -c1-400
