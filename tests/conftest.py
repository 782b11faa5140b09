import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "ops_small.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def golden_attention():
    path = os.path.join(ROOT, "tests", "golden", "attention_small.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o
    _o.lib()
    return _o
