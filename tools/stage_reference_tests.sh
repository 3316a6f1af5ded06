#!/bin/bash
# Copies the reference's own test files for the hot path (+ the sample assets they read) into the untracked scratch
# directory _ref_tests/ so that they travel to the GPU box with gpurun and run UNCHANGED against kaolin_amd through
# install_as_kaolin() (see _ref_tests/conftest.py written below).  Nothing from /root/reference enters the repository:
# _ref_tests/ is git-ignored.  usage: bash tools/stage_reference_tests.sh && gpurun -- 'bash tools/run_reference_tests.sh'
set -e
ref=/root/reference; dst=$(dirname "$0")/../_ref_tests
rm -rf "$dst"; mkdir -p "$dst/tests/python/kaolin" "$dst/tests/samples"
for f in metrics/test_pointcloud.py metrics/test_trianglemesh.py metrics/test_render.py render/mesh/test_rasterization.py \
         render/mesh/test_dibr.py render/mesh/test_deftet.py render/mesh/test_utils.py ops/mesh/test_check_sign.py \
         ops/conversions/test_trianglemesh.py; do
  mkdir -p "$dst/tests/python/kaolin/$(dirname $f)"; cp "$ref/tests/python/kaolin/$f" "$dst/tests/python/kaolin/$f"
done
cp "$ref/tests/samples/model.obj" "$ref/tests/samples/model.mtl" "$ref/tests/samples/tex.png" "$dst/tests/samples/" 2>/dev/null || true
cp -r "$ref/tests/samples/dibr" "$dst/tests/samples/"
for d in rasterization ops render; do [ -d "$ref/tests/samples/$d" ] && cp -r "$ref/tests/samples/$d" "$dst/tests/samples/" || true; done
cat > "$dst/conftest.py" <<'PY'
# scratch harness: the reference's tests import `kaolin`; alias kaolin_amd under that name first
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kaolin_amd
kaolin_amd.install_as_kaolin()
PY
du -sh "$dst"
