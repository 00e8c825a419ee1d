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


def draw_noise(sd, batch, generator=None):
    """oracle.draw_flipout_noise for tests that INJECT the tensors into the engine.  bayesian-torch draws its signs as
    `uniform_(-1, 1).sign()`, which is exactly 0 with probability 2^-24 per element (u = 0.5 of torch's 24-bit uniform): the perturbation
    of that element is then dropped.  The engine's injection format is +1 / -1 (packed into bits by the fused kernels), so a drawn 0 is
    replaced by +1 here - on both sides of the comparison, since the oracle computes with the returned tensors too."""
    from oracle import ntf_oracle as O
    noise = O.draw_flipout_noise(sd, batch, generator)
    for n in noise:
        for k in ("s_in", "s_out"):
            n[k][n[k] == 0] = 1.0
    return noise
