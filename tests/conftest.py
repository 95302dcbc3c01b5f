import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a `-m gpu` run on a box without a GPU must fail loudly, not skip silently
    pass


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def dev():
    assert torch.cuda.is_available(), "gpu-marked test started on a box without a ROCm device"
    return torch.device("cuda:0")


def rel_max(a, b):
    """max|a-b| / max|b| -- the normwise metric of SURVEY.md hard part 4."""
    a = np.asarray(a)
    b = np.asarray(b)
    denom = float(np.abs(b).max())
    return float(np.abs(a - b).max()) / (denom if denom > 0 else 1.0)
