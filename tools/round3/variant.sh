#!/bin/bash
# usage: tools/round3/variant.sh NAME "file1.hip file2.hip" "-DKAMD_...": the named files rebuilt with DEFS, linked with the other objects
# of the last `make` -> kaolin_amd/libkaolin_amd_NAME.so (selected with KAMD_LIB_PATH)
cd "$(dirname "$0")/../../kaolin_amd/csrc"
mkdir -p var_obj/$1
objs=""
for f in $2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-math-errno -Wall -Wno-unused-function -DKAMD_EXPERIMENT $3 -c $f -o var_obj/$1/${f%.hip}.o || exit 1
  objs="$objs var_obj/$1/${f%.hip}.o"
done
others=$(for o in *.o; do skip=0; for f in $2; do [ "$o" = "${f%.hip}.o" ] && skip=1; done; [ $skip = 0 ] && echo $o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkaolin_amd_$1.so $objs $others
