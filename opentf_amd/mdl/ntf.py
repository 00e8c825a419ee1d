"""Plugin base mirroring the reference's `mdl.ntf.Ntf` (src/mdl/ntf.py:5-31): same constructor, attributes,
output-directory naming and method names.  `evaluate` / `adila` are the reference's post-hoc CPU metrics
(src/mdl/ntf.py:32-134) and are out of this build's scope: when the classes of
this package are mounted inside the reference tree they inherit those two methods from the
reference's own `Ntf`; standalone they raise.
"""
from __future__ import annotations

import os
import random

import numpy as np


def cfg_items(cfg):
    """(key, value) pairs of a config section: omegaconf DictConfig, dict or attribute-dict."""
    try:
        from omegaconf import OmegaConf  # present in the reference's environment
        if OmegaConf.is_config(cfg):
            return list(OmegaConf.to_container(cfg, resolve=True).items())
    except ImportError:
        pass
    return list(dict(cfg).items())


def cfg_get(cfg, key, default=None):
    try:
        v = cfg[key] if hasattr(cfg, "__getitem__") else getattr(cfg, key)
    except (KeyError, AttributeError):
        return default
    return v


def cfg2str(cfg) -> str:
    """'.'.join(f'{k}{v}') of the model's config section (src/pkgmgr.py:95) — part of the on-disk contract."""
    def fmt(v):
        return str(list(v)) if isinstance(v, (list, tuple)) or type(v).__name__ == "ListConfig" else str(v)
    return ".".join(f"{k}{fmt(v)}" for k, v in cfg_items(cfg)) if cfg else ""


def set_seed(seed):
    """src/pkgmgr.py:81-93: python, numpy and torch generators, once per model object (src/mdl/ntf.py:14)."""
    if seed is None:
        return
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class _NullWriter:
    def __init__(self, log_dir=None): pass
    def add_scalar(self, tag, scalar_value, global_step): pass
    def close(self): pass


def summary_writer():
    """tensorboardX.SummaryWriter when importable (src/mdl/ntf.py:13), else a no-op with the same surface."""
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter
    except ImportError:
        return _NullWriter


class Ntf:
    def __init__(self, output, device, seed, cfg):
        self.cfg = cfg
        self.seed = seed
        self.device = device
        self.model = None
        self.is_bayesian = False
        self.writer = summary_writer()
        set_seed(self.seed)
        self.output = output + self.name()
        if not os.path.isdir(self.output):
            os.makedirs(self.output)

    def name(self):
        return f"/{self.__class__.__name__.lower()}.{cfg2str(self.cfg)}"

    def learn(self, teamsvecs, splits, prev_model): pass

    def test(self, teamsvecs, splits, testcfg): pass

    def evaluate(self, teamsvecs, splits, evalcfg):
        raise NotImplementedError("evaluate() is the reference's CPU metric stage (src/mdl/ntf.py:32-92); mount this plugin in the "
                                  "reference tree (INTEGRATION.md) to inherit it")

    def adila(self, teamsvecs, splits, faircfg):
        raise NotImplementedError("adila() is the reference's fairness stage (src/mdl/ntf.py:108-134); see INTEGRATION.md")
