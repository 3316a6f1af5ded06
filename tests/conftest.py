"""pytest configuration: the ``gpu`` marker, repo-root imports, oracle build."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a box without a GPU is skipped (the driver selects with -m).
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _built_oracle():
    import oracle
    oracle.build()
    return oracle


GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
