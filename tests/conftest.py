import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def params_from(npz, prefix):
    """OrderedDict of torch tensors for keys `prefix + <state_dict key>` in file order."""
    import torch
    from collections import OrderedDict
    out = OrderedDict()
    for k in npz.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(npz[k].copy())
    return out


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
