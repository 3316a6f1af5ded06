#!/bin/bash
# raster_tile under its ablations: time, then SQ instruction / wait counters per mode (own PMC pass)
set -u
tag=${1:-r03h}; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
timeout 200 python tools/round3/raster_insts.py time 2>&1 | grep -v amdgpu.ids > $out/raster_modes_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc1 -- python $repo/tools/round3/raster_insts.py pmc > /dev/null 2>&1
find $out/pmc1 -name '*counter_collection.csv' -exec cp {} $out/pmc_sq1.csv \;
rm -rf $out/pmc1
python - <<P > $out/raster_modes_pmc.txt
import csv, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open('$out/pmc_sq1.csv')):
    if 'raster_tile_kernel2' in r['Kernel_Name']:
        rows[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
modes = [0, 32, 8, 1, 3, 7, 'nofaces']
ids = sorted(rows)
names = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_BUSY_CYCLES']
print('mode | ' + ' | '.join(names))
for i, d in enumerate(ids):
    if i % 2 == 1:
        print(modes[i // 2] if i // 2 < len(modes) else '?', '|', ' | '.join('%.0f' % rows[d].get(n, -1) for n in names))
P
rm -f $out/pmc_sq1.csv
cat $out/raster_modes_time.txt $out/raster_modes_pmc.txt
