import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session', autouse=True)
def _torch_gpu_first():
    """On a GPU box: let torch initialise its HIP context before the library's first call (tests
    that hand torch device buffers to the library otherwise depend on the test order)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load
