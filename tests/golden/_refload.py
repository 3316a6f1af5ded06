"""Loads the reference's pure-PyTorch oracle modules BY PATH from /root/reference (this container
only -- /root/reference does not exist on the GPU box) with a stub ``kaolin._C`` so that golden
vectors can be generated from the reference itself.  Used only by make_golden.py."""
import importlib.util
import os
import sys
import types

REF = '/root/reference'


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    """Returns a dict of reference modules holding the oracles of SURVEY.md section 8(c)."""
    if 'kaolin' in sys.modules and getattr(sys.modules['kaolin'], '_is_ref_stub', False):
        return sys.modules['kaolin']._mods
    import torch
    k = _pkg('kaolin')
    k._is_ref_stub = True
    c = _pkg('kaolin._C')
    c.render = _pkg('kaolin._C.render')
    c.render.mesh = _pkg('kaolin._C.render.mesh')
    c.metrics = _pkg('kaolin._C.metrics')
    c.ops = _pkg('kaolin._C.ops')
    for n in ('kaolin.ops', 'kaolin.ops.mesh', 'kaolin.ops.conversions', 'kaolin.ops.spc', 'kaolin.rep',
              'kaolin.metrics', 'kaolin.render', 'kaolin.render.mesh', 'kaolin.render.camera', 'kaolin.utils'):
        _pkg(n)
    mods = {}
    # stubs for imports the oracle files do not use on our path
    sys.modules['kaolin.ops.spc'].points = types.ModuleType('points')
    sys.modules['kaolin.ops.spc.points'] = sys.modules['kaolin.ops.spc'].points
    for n in ('quantize_points', 'points_to_morton', 'morton_to_points', 'unbatched_points_to_octree'):
        setattr(sys.modules['kaolin.ops.spc.points'], n, None)
    sys.modules['kaolin.ops.batch'] = types.ModuleType('kaolin.ops.batch')
    for n in ('tile_to_packed', 'packed_to_padded', 'get_first_idx'):
        setattr(sys.modules['kaolin.ops.batch'], n, None)
    sys.modules['kaolin.rep.spc'] = types.ModuleType('kaolin.rep.spc')
    sys.modules['kaolin.rep.spc'].Spc = None
    sys.modules['kaolin.rep'].Spc = None
    # metrics/trianglemesh.py imports ..ops.mesh.uniform_laplacian (unused on our path) and its oracle calls
    # torch.cuda.synchronize() (metrics/trianglemesh.py:232), which raises on a GPU-less box -> no-op it here
    sys.modules['kaolin.ops.mesh'].uniform_laplacian = None
    if not torch.cuda.is_available():
        torch.cuda.synchronize = lambda *a, **k: None
    mods['legacy_camera'] = _load('kaolin.render.camera.legacy', 'kaolin/render/camera/legacy.py')
    mods['pointcloud'] = _load('kaolin.metrics.pointcloud', 'kaolin/metrics/pointcloud.py')
    mods['deftet'] = _load('kaolin.render.mesh.deftet', 'kaolin/render/mesh/deftet.py')
    mods['ops_trianglemesh'] = _load('kaolin.ops.mesh.trianglemesh', 'kaolin/ops/mesh/trianglemesh.py')
    mods['conv_pointcloud'] = _load('kaolin.ops.conversions.pointcloud', 'kaolin/ops/conversions/pointcloud.py')
    mods['conv_trianglemesh'] = _load('kaolin.ops.conversions.trianglemesh', 'kaolin/ops/conversions/trianglemesh.py')
    mods['trianglemesh'] = _load('kaolin.metrics.trianglemesh', 'kaolin/metrics/trianglemesh.py')
    k._mods = mods
    return mods
