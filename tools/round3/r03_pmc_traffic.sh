#!/bin/bash
# the two HBM-traffic passes of tools/round_profile.sh alone (FETCH_SIZE, WRITE_SIZE over tools/pmc_traffic.py) -> gpurun_out/<tag>/traffic.json
set -u
tag=${1:-r03z2}; repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python $repo/tools/pmc_traffic.py > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python $repo/tools/pmc_traffic.py > /dev/null 2>&1
find $out/pmc_f -name '*counter_collection.csv' -exec cp {} $out/pmc_fetch.csv \;
find $out/pmc_w -name '*counter_collection.csv' -exec cp {} $out/pmc_write.csv \;
rm -rf $out/pmc_f $out/pmc_w
python $repo/tools/parse_traffic.py $out/pmc_fetch.csv $out/pmc_write.csv $out/traffic.json "tools/round3/r03_pmc_traffic.sh $tag (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_traffic.py)" > /dev/null 2> $out/parse_traffic.err
python $repo/tools/pmc_table.py $out/pmc_fetch.csv $out/pmc_FETCH_SIZE.txt "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python tools/pmc_traffic.py" > /dev/null 2>&1
python $repo/tools/pmc_table.py $out/pmc_write.csv $out/pmc_WRITE_SIZE.txt "rocprofv3 --pmc WRITE_SIZE --output-format csv -- python tools/pmc_traffic.py" > /dev/null 2>&1
rm -f $out/pmc_fetch.csv $out/pmc_write.csv
python -c "
import json; t=json.load(open('$out/traffic.json'))
for k in ('_step','_chamfer_step','_chamfer_operator','_voxelgrid_256','_point_to_mesh_1Mx50k'):
    v=t.get(k); print(k, v and {a:(b if not isinstance(b,dict) else {n:round(x['fetch_bytes_per_call']+x['write_bytes_per_call']) for n,x in b.items()}) for a,b in v.items() if a!='other_kernels'})
" | cut -c1-1500
