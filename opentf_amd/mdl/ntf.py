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


def dist_rank():
    """rank of this process under torchrun (one process per GPU), 0 when single-process"""
    import torch
    d = torch.distributed
    return d.get_rank() if d.is_available() and d.is_initialized() else 0


def _dist_barrier():
    import torch
    d = torch.distributed
    if d.is_available() and d.is_initialized() and d.get_world_size() > 1:
        d.barrier()


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


class _PredJob:
    __slots__ = ("pred_set", "fold", "path", "final")
    def __init__(self, pred_set, fold, path, final): self.pred_set, self.fold, self.path, self.final = pred_set, fold, path, final


def _pred_jobs(output, splits, on_train, per_epoch):
    """the prediction files of a run directory, set by set and fold by fold; `final` marks a fold's end-of-training file `f{k}.{set}.pred` (the one the set-level
    tables aggregate), the per-epoch files `f{k}.{set}.e{e}.pred` follow it in epoch order"""
    import re
    names = os.listdir(output) if per_epoch else []
    for pred_set in (("test", "train", "valid") if on_train else ("test",)):
        for k in splits["folds"].keys():
            yield _PredJob(pred_set, k, f"{output}/f{k}.{pred_set}.pred", True)
            epoch_files = [(int(m.group(1)), n) for n in names for m in [re.match(rf"f{k}\.{pred_set}\.e(\d+)\.pred$", n)] if m]
            for _, n in sorted(epoch_files):
                yield _PredJob(pred_set, k, f"{output}/{n}", False)


def _read_pred(path):
    """`y_pred` of a prediction file (src/mdl/fnn.py:218) as a numpy array (dense files) or a scipy CSR (the top-K files, sparse COO inside)"""
    import scipy.sparse
    import torch
    y = torch.load(path, map_location="cpu", weights_only=False)["y_pred"]
    if not y.is_sparse: return y.numpy()
    y = y.coalesce(); ij = y.indices().numpy()
    return scipy.sparse.csr_matrix((y.values().numpy(), (ij[0], ij[1])), shape=tuple(y.size()))


class _FoldTable:
    """what the set-level files need of the folds' final files: one column of means per fold, and the running sum of the per-instance tables"""
    def __init__(self): self.means, self.inst_sum = [], None

    def add(self, mean, inst):
        self.means.append(mean["mean"])
        if inst is not None: self.inst_sum = inst.copy() if self.inst_sum is None else self.inst_sum.add(inst, fill_value=0)

    def write(self, stem, n_folds):
        import pandas as pd
        per_fold = pd.concat(self.means, axis=1)
        pd.DataFrame({"mean": per_fold.mean(axis=1), "std": per_fold.std(axis=1)}).to_csv(f"{stem}.mean.csv")
        if self.inst_sum is not None: (self.inst_sum / n_folds).to_csv(f"{stem}.instance_mean.csv", index=False)


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
        os.makedirs(self.output, exist_ok=True)   # one process per GPU: every rank constructs the model

    def name(self):
        return f"/{self.__class__.__name__.lower()}.{cfg2str(self.cfg)}"

    def learn(self, teamsvecs, splits, prev_model): pass

    def test(self, teamsvecs, splits, testcfg): pass

    def evaluate(self, teamsvecs, splits, evalcfg):
        """Eval stage: scores the `.pred` files test() wrote and leaves the files the reference's stage leaves (src/mdl/ntf.py:32-92): per prediction file
        `<file>.eval.mean.csv` (+ `.eval.instance.csv`, `.eval.roc.pkl`), per prediction set `{set}.pred.eval.mean.csv` (mean / std over the folds' final files) and
        `{set}.pred.eval.instance_mean.csv`.  A thin driver: `_pred_jobs` lists the files, `evl.metric.score_predictions` does the arithmetic (rank metrics and skill
        coverage on the device), `_FoldTable` keeps what the set-level files need.  Mounted in the reference tree the plugin inherits the reference's own evaluate()
        instead."""
        from ..evl import metric
        assert os.path.isdir(self.output), f"No folder for {self.output} exist!"
        if dist_rank() != 0:        # torchrun: ONE writer of the csv / pkl files (as in test()); the others wait for it
            _dist_barrier(); return
        spec = metric.EvalSpec.from_cfg(evalcfg)
        tables = {}
        for job in _pred_jobs(self.output, splits, bool(cfg_get(evalcfg, "on_train")), bool(cfg_get(evalcfg, "per_epoch"))):
            rows = splits["test"] if job.pred_set == "test" else splits["folds"][job.fold][job.pred_set]
            inst, mean, roc = metric.score_predictions(teamsvecs, rows, _read_pred(job.path), spec)
            if roc is not None:   # the (fpr, tpr) pair plot_roc consumes (src/mdl/ntf.py:67-69)
                import pickle
                with open(f"{job.path}.eval.roc.pkl", "wb") as f: pickle.dump(roc, f)
            if spec.per_instance: inst.to_csv(f"{job.path}.eval.instance.csv", float_format="%.5f", index=False)
            mean.to_csv(f"{job.path}.eval.mean.csv")
            if job.final: tables.setdefault(job.pred_set, _FoldTable()).add(mean, inst if spec.per_instance else None)
        for pred_set, t in tables.items():
            t.write(f"{self.output}/{pred_set}.pred.eval", len(splits["folds"]))
        _dist_barrier()   # rank 0 wrote: the other ranks (waiting at the top) may go on

    def adila(self, teamsvecs, splits, faircfg):
        raise NotImplementedError("adila() is the reference's fairness stage (src/mdl/ntf.py:108-134); see INTEGRATION.md")
