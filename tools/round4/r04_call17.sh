#!/bin/bash
# round 4, call 17: where the binning kernel's wavefronts spend their time on the bowl alone / the knot / the sphere (prof build)
set -u
out=gpurun_out/r04c17; mkdir -p $out
L=$(pwd)/kaolin_amd
for s in bowl knot sphere; do
  KAMD_PROF_SCENE=$s KAMD_LIB_PATH=$L/libkaolin_amd_prof.so timeout 200 python tools/phase_prof.py 2>&1 | grep -v amdgpu.ids | head -2 | tee $out/phase_$s.txt | cut -c1-700
done
