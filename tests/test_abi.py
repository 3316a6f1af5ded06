"""The C-ABI library loads and exports every symbol include/kaolin_amd.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'kaolin_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(kamd_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_something():
    syms = _header_symbols()
    assert 'kamd_version' in syms and len(syms) >= 25


def test_library_exports_every_declared_symbol():
    from kaolin_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _header_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from kaolin_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()


def test_load_and_version():
    from kaolin_amd import _lib
    lib = _lib.load()
    assert lib.kamd_version().startswith(b'kaolin_amd')


def test_argument_counts_match_header():
    """ctypes argtypes lengths equal the number of parameters in the header prototypes."""
    from kaolin_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'kaolin_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    for name, params in re.findall(r'\b(kamd_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', text):
        params = params.strip()
        n = 0 if params in ('', 'void') else params.count(',') + 1
        assert len(_lib.SIGNATURES[name][1]) == n, name


def test_no_product_import_of_oracle():
    """Nothing under kaolin_amd/ may import the oracle (it is test infrastructure)."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, 'kaolin_amd')):
        for f in files:
            if f.endswith('.py') and re.search(r'^\s*(import|from)\s+oracle\b', open(os.path.join(d, f)).read(), re.M):
                bad.append(f)
    assert not bad


def test_workspace_queries_are_host_only_and_shape_driven():
    """The *_workspace entry points are pure host arithmetic (callable without a GPU): sizes grow with the problem, and
    0 tells the shim that a path does not apply (it then takes the other one)."""
    from kaolin_amd import _lib
    lib = _lib.load()
    sd = lib.kamd_sided_distance_forward_workspace
    pair = lib.kamd_sided_distance_pair_forward_workspace
    assert sd(1, 100000, 100000, 4) > 0 and sd(2, 100000, 100000, 4) > sd(1, 100000, 100000, 4)
    assert sd(1, 100000, 100000, 8) > sd(1, 100000, 100000, 4)    # fp64: the same grid pipeline with double points in its sorted copy
    assert sd(1, 1000, 1000, 8) == 0 and sd(1, 1000, 1000, 2) == 0   # small fp64 / fp16 clouds: the generic kernel needs no scratch
    assert sd(1, 100000, 100000, 2) > sd(1, 100000, 100000, 4)      # at::Half takes the grid search too: float copies + the float pipeline
    assert sd(0, 10, 10, 4) == 0 and sd(1, 0, 10, 4) == 0
    # both directions from one binning pass: both clouds must be large enough for the grid search, fp32 / fp64
    assert pair(1, 100000, 100000, 4) > 0 and pair(1, 100000, 100000, 8) > pair(1, 100000, 100000, 4)
    assert pair(1, 100, 100000, 4) == 0 and pair(1, 100000, 100, 4) == 0 and pair(1, 100000, 100000, 2) == 0
    assert pair(70000, 8192, 8192, 4) == 0                        # batch beyond the launch grid's y extent
    assert pair(1, 8192, 4000000, 4) > pair(1, 8192, 400000, 4)
    assert lib.kamd_rasterize_forward_workspace(8, 1024, 1024, 400000, 4) > \
        lib.kamd_rasterize_forward_workspace(1, 1024, 1024, 50000, 4) > 0
    assert lib.kamd_rasterize_forward_workspace(1, 0, 1024, 50000, 4) == 0
    assert lib.kamd_trianglemeshes_to_voxelgrids_workspace(1, 30000, 4) > 0
    assert lib.kamd_deftet_forward_workspace(1, 50000, 4096, 4) > 0
