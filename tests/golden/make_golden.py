"""Generates the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container (where /root/reference is mounted):
    python tests/golden/make_golden.py [section ...]
The fixtures (.npz) are committed; this script is the record of how they were made.  Each section
imports the reference's own pure-PyTorch oracle by path and stores seeded inputs
plus the oracle's outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

SECTIONS = {}


def section(f):
    SECTIONS[f.__name__] = f
    return f


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name + '.npz', {k: v.shape for k, v in out.items()})


@section
def sided_distance():
    """reference oracle: kaolin/metrics/pointcloud.py:186-197 (_sided_distance, values only) +
    an fp64 brute-force argmin for the indices (unique minima on continuous random data)."""
    ref = _refload.load_reference()['pointcloud']
    cases = {}
    for tag, (B, N, M, scale, seed) in {'a': (3, 50, 50, 100., 0), 'b': (2, 257, 1031, 1., 1),
                                       'c': (1, 2000, 2000, 1., 0)}.items():
        torch.manual_seed(seed)
        if tag == 'c':   # BASELINE config C1: uniform [0,1)^3, seed 0, consecutive draws
            p1 = torch.rand(B, N, 3)
            p2 = torch.rand(B, M, 3)
        else:
            p1 = torch.randn(B, N, 3) * scale
            p2 = torch.randn(B, M, 3) * scale
        for dt, dn in ((torch.float32, 'f32'), (torch.float64, 'f64')):
            a, b = p1.to(dt), p2.to(dt)
            d = ref._sided_distance(a, b)
            dd = ((a.double()[:, :, None, :] - b.double()[:, None, :, :]) ** 2).sum(-1)
            idx = dd.argmin(-1)
            cases[f'{tag}_{dn}_dist'] = d
            cases[f'{tag}_{dn}_idx'] = idx
        cases[f'{tag}_p1'] = p1
        cases[f'{tag}_p2'] = p2
    save('sided_distance', **cases)


if __name__ == '__main__':
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        SECTIONS[s]()
