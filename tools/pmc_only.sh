set -u
repo=$(pwd); out=$repo/gpurun_out/r01i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python $repo/tools/pmc_traffic.py > /dev/null 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python $repo/tools/pmc_traffic.py > /dev/null 2>&1
find $out/pmc_f -name '*counter_collection.csv' -exec cp {} $out/pmc_fetch.csv \;
find $out/pmc_w -name '*counter_collection.csv' -exec cp {} $out/pmc_write.csv \;
rm -rf $out/pmc_f $out/pmc_w
ls -la $out
