"""Python access to the CPU oracle -- TEST INFRASTRUCTURE ONLY (see kaolin_oracle.c header).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  Nothing under ``kaolin_amd/`` imports it.

Functions take and return CPU ``torch`` tensors and mirror the signatures of the reference's
``kaolin._C`` operators.  ``build()`` compiles ``kaolin_oracle.c`` with gcc (serial + OpenMP).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build(force=False):
    targets = ['libkaolin_oracle.so', 'libkaolin_oracle_omp.so']
    src = os.path.join(_HERE, 'kaolin_oracle.c')
    stale = force or any(
        not os.path.exists(os.path.join(_HERE, t)) or
        os.path.getmtime(os.path.join(_HERE, t)) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.run(['make', '-C', _HERE, '-B'] if force else ['make', '-C', _HERE], check=True,
                       stdout=subprocess.DEVNULL)


def lib(omp=False):
    key = 'omp' if omp else 'serial'
    if key not in _libs:
        name = 'libkaolin_oracle_omp.so' if omp else 'libkaolin_oracle.so'
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _libs[key] = ctypes.CDLL(path)
    return _libs[key]


def num_threads(omp=True):
    f = lib(omp).oracle_num_threads
    f.restype = ctypes.c_int
    return int(f())


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _cpu(t, dtype=None):
    t = t.detach().cpu().contiguous()
    if dtype is not None:
        t = t.to(dtype)
    return t


_SFX = {torch.float32: 'f32', torch.float64: 'f64', torch.float16: 'f16'}


# ---- sided distance -------------------------------------------------------------
def sided_distance_forward(p1, p2, omp=False):
    p1, p2 = _cpu(p1), _cpu(p2)
    B, N, M = p1.shape[0], p1.shape[1], p2.shape[1]
    dist = torch.zeros((B, N), dtype=p1.dtype)
    idx = torch.zeros((B, N), dtype=torch.long)
    f = getattr(lib(omp), f'oracle_sided_distance_forward_{_SFX[p1.dtype]}')
    f(ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), _p(p1), _p(p2), _p(dist), _p(idx))
    return dist, idx


def sided_distance_backward(grad, p1, p2, idx):
    grad, p1, p2, idx = _cpu(grad), _cpu(p1), _cpu(p2), _cpu(idx)
    B, N, M = p1.shape[0], p1.shape[1], p2.shape[1]
    g1, g2 = torch.zeros_like(p1), torch.zeros_like(p2)
    f = getattr(lib(False), f'oracle_sided_distance_backward_{_SFX[p1.dtype]}')
    f(ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), _p(grad), _p(p1), _p(p2), _p(idx), _p(g1), _p(g2))
    return g1, g2
