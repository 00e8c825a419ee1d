"""`Bnn`: the reference's Bayesian (Flipout) variant (src/mdl/bnn.py:12-27): Fnn's whole train/test loop with every
Linear replaced by bayesian-torch's LinearFlipout (prior N(0,1), posterior init mu~N(0,0.1), rho~N(-3,0.1)),
`KL/B` added to the loss (src/mdl/fnn.py:136,149) and `nmc` stochastic forwards averaged at test time
(src/mdl/fnn.py:202-211).  bayesian-torch is not needed: its arithmetic is restated in the HIP kernels
(its restatement is pinned distributionally on the 40 bayesian-torch checkpoints / .pred files the reference's authors committed (tests/test_bnn_committed.py, g12))."""
from __future__ import annotations

from collections import OrderedDict

from .fnn import Fnn, make_fnn
from .ntf import cfg_get

POSTERIOR_MU_INIT, POSTERIOR_RHO_INIT = 0.0, -3.0  # src/mdl/bnn.py:19-24


def make_bnn(fnn_cls):
    class Bnn(fnn_cls):
        def __init__(self, output, device, seed, cgf):
            super().__init__(output, device, seed, cgf)
            self.is_bayesian = True

        def init(self, input_size, output_size):
            """super().init() first (its draws are consumed exactly as dnn_to_bnn(super().init(...)) does, bnn.py:25), then
            LinearFlipout.init_parameters per layer in order: mu_weight, rho_weight, mu_bias, rho_bias."""
            import torch
            super().init(input_size, output_size)
            dims = self._dims
            sd = OrderedDict()
            for i in range(len(dims) - 1):
                o, n = dims[i + 1], dims[i]
                sd[f"layers.{i}.mu_weight"] = torch.empty(o, n).normal_(POSTERIOR_MU_INIT, 0.1)
                sd[f"layers.{i}.rho_weight"] = torch.empty(o, n).normal_(POSTERIOR_RHO_INIT, 0.1)
                sd[f"layers.{i}.mu_bias"] = torch.empty(o).normal_(POSTERIOR_MU_INIT, 0.1)
                sd[f"layers.{i}.rho_bias"] = torch.empty(o).normal_(POSTERIOR_RHO_INIT, 0.1)
            self.model = sd
            self.is_bayesian = True
            return self.model

    Bnn.__qualname__ = "Bnn"
    return Bnn


Bnn = make_bnn(Fnn)
