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
# The reference's OWN Python layer over the operator boundary of SURVEY 8(b): with KAMD_REF_LAYER=1 the four modules that
# hold its autograd Functions (RasterizeCuda, DibrSoftMaskCuda, _SidedDistanceFunction, _UnbatchedTriangleDistanceCuda) are
# loaded UNTOUCHED from this scratch copy and bound to kaolin._C := kaolin_amd._C -- the composition a maintainer gets by
# replacing kaolin/csrc/bindings.cpp:103-115 and nothing else.
mkdir -p "$dst/ref_layer/render/mesh" "$dst/ref_layer/metrics"
for f in render/mesh/rasterization.py render/mesh/dibr.py render/mesh/nvdiffrast_context.py metrics/pointcloud.py metrics/trianglemesh.py; do
  cp "$ref/kaolin/$f" "$dst/ref_layer/$f"
done
cat > "$dst/conftest.py" <<'PY'
# scratch harness: the reference's tests import `kaolin`; alias kaolin_amd under that name first
import importlib.util, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import kaolin_amd
kal = kaolin_amd.install_as_kaolin()


def _load_reference_module(name, rel):
    """Loads _ref_tests/ref_layer/<rel> as module `name` (its `from kaolin import _C` / relative imports resolve through
    sys.modules, i.e. to kaolin_amd) and re-exports its public names into the parent package, as the reference's
    `from .x import *` does."""
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, 'ref_layer', rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    parent_name, leaf = name.rsplit('.', 1)
    parent = sys.modules[parent_name]
    setattr(parent, leaf, mod)
    for k in getattr(mod, '__all__', []):
        setattr(parent, k, getattr(mod, k))
    return mod


if os.environ.get('KAMD_REF_LAYER') == '1':
    _load_reference_module('kaolin.render.mesh.nvdiffrast_context', 'render/mesh/nvdiffrast_context.py')
    r = _load_reference_module('kaolin.render.mesh.rasterization', 'render/mesh/rasterization.py')
    d = _load_reference_module('kaolin.render.mesh.dibr', 'render/mesh/dibr.py')
    p = _load_reference_module('kaolin.metrics.pointcloud', 'metrics/pointcloud.py')
    t = _load_reference_module('kaolin.metrics.trianglemesh', 'metrics/trianglemesh.py')
    for m in (r, d, p, t):
        assert m.__file__.startswith(os.path.join(HERE, 'ref_layer')), m.__file__
        assert m._C is kaolin_amd._C
    assert kal.render.mesh.rasterize is r.rasterize and kal.render.mesh.dibr_rasterization is d.dibr_rasterization
    assert kal.metrics.pointcloud.chamfer_distance is p.chamfer_distance
    assert kal.metrics.trianglemesh.point_to_mesh_distance is t.point_to_mesh_distance
    print('KAMD_REF_LAYER=1: reference rasterization.py / dibr.py / metrics/pointcloud.py / metrics/trianglemesh.py over kaolin_amd._C',
          file=sys.stderr)
PY
du -sh "$dst"
