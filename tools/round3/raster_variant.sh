#!/bin/bash
# usage: tools/round3/raster_variant.sh NAME "-DKAMD_RASTER_...": rasterize.hip rebuilt with DEFS, linked with the other objects -> kaolin_amd/libkaolin_amd_NAME.so
cd "$(dirname "$0")/../../kaolin_amd/csrc"
mkdir -p var_obj/$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-math-errno -Wall -Wno-unused-function $2 -c rasterize.hip -o var_obj/$1/rasterize.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkaolin_amd_$1.so var_obj/$1/rasterize.o $(ls *.o | grep -v '^rasterize.o$')
