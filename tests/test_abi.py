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
