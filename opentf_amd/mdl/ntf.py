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
        """The reference's eval stage (src/mdl/ntf.py:32-92) with the per-instance metrics on the device (opentf_amd.evl.metric):
        same inputs (the `.pred` files test() wrote), same outputs (`*.pred.eval.mean.csv`, `*.pred.eval.instance.csv`,
        `{set}.pred.eval.mean.csv` with mean/std over folds).  Mounted in the reference tree the plugin inherits the reference's
        own evaluate() instead."""
        import re
        import pandas as pd
        import scipy.sparse
        import torch
        from ..evl import metric
        assert os.path.isdir(self.output), f"No folder for {self.output} exist!"
        if dist_rank() != 0:        # torchrun: ONE writer of the csv / pkl files (as in test()); the others wait for it
            _dist_barrier(); return
        y_test = teamsvecs["member"][splits["test"]]
        trec = list(cfg_get(cfg_get(evalcfg, "metrics"), "trec") or [])
        other = list(cfg_get(cfg_get(evalcfg, "metrics"), "other") or [])
        per_instance = bool(cfg_get(evalcfg, "per_instance"))
        for pred_set in (["test", "train", "valid"] if cfg_get(evalcfg, "on_train") else ["test"]):
            fold_mean, mean_std = pd.DataFrame(), pd.DataFrame()
            fold_inst = pd.DataFrame()
            for foldidx in splits["folds"].keys():
                Y = y_test if pred_set == "test" else teamsvecs["member"][splits["folds"][foldidx][pred_set]]
                predfiles = [f"{self.output}/f{foldidx}.{pred_set}.pred"]
                if cfg_get(evalcfg, "per_epoch"):
                    predfiles += [f"{self.output}/{_}" for _ in os.listdir(self.output) if re.match(rf"f{foldidx}.{pred_set}.e\d+.pred$", _)]
                for i, predfile in enumerate(sorted(sorted(predfiles), key=len)):
                    Y_ = torch.load(predfile, map_location="cpu", weights_only=False)["y_pred"]
                    if Y_.is_sparse:
                        Y_ = Y_.coalesce()
                        ind = Y_.indices().numpy()
                        Y_ = scipy.sparse.csr_matrix((Y_.values().numpy(), (ind[0], ind[1])), shape=tuple(Y_.size()))
                    else:
                        Y_ = Y_.numpy()
                    assert Y.shape == Y_.shape, f"Shape mismatch between truth Y {Y.shape} vs preds Y_ {Y_.shape}!"
                    df, df_mean = pd.DataFrame(), pd.DataFrame()
                    if trec:
                        df, df_mean = metric.calculate_metrics(Y, Y_, cfg_get(evalcfg, "topK"), per_instance, trec)
                    auc = [m for m in other if "aucroc" in m]
                    if auc:
                        aucroc, fpr_tpr = metric.calculate_auc_roc(Y, Y_, curve=(auc[0] == "aucroc+"))
                        if df_mean.empty: df_mean = pd.DataFrame(columns=["mean"])
                        df_mean.loc["aucroc"] = aucroc
                        if fpr_tpr:   # the (fpr, tpr) pair plot_roc consumes (src/mdl/ntf.py:67-69)
                            import pickle
                            with open(f"{predfile}.eval.roc.pkl", "wb") as outfile: pickle.dump(fpr_tpr, outfile)
                    skc = [m for m in other if "skill_coverage" in m]
                    if skc:
                        X = teamsvecs["skill"] if scipy.sparse.issparse(teamsvecs["skill"]) else teamsvecs["original_skill"]
                        X = X[splits["test"]] if pred_set == "test" else X[splits["folds"][foldidx][pred_set]]
                        df_skc, df_mean_skc = metric.calculate_skill_coverage(X, Y_, teamsvecs["skillcoverage"], per_instance,
                                                                               topks=skc[0].replace("skill_coverage_", ""))
                        df = df_skc if (df is None or df.empty) else pd.concat([df.reset_index(drop=True), df_skc.reset_index(drop=True)], axis=1)
                        df_mean = df_mean_skc if df_mean.empty else pd.concat([df_mean, df_mean_skc], axis=0)
                    if per_instance: df.to_csv(f"{predfile}.eval.instance.csv", float_format="%.5f", index=False)
                    df_mean.to_csv(f"{predfile}.eval.mean.csv")
                    if i == 0:
                        fold_mean = pd.concat([fold_mean, df_mean], axis=1)
                        if per_instance: fold_inst = fold_inst.add(df, fill_value=0)
            mean_std["mean"] = fold_mean.mean(axis=1)
            mean_std["std"] = fold_mean.std(axis=1)
            mean_std.to_csv(f"{self.output}/{pred_set}.pred.eval.mean.csv")
            if per_instance:
                fold_inst.truediv(len(splits["folds"].keys())).to_csv(f"{self.output}/{pred_set}.pred.eval.instance_mean.csv", index=False)
        _dist_barrier()   # rank 0 wrote: the other ranks (waiting at the top) may go on

    def adila(self, teamsvecs, splits, faircfg):
        raise NotImplementedError("adila() is the reference's fairness stage (src/mdl/ntf.py:108-134); see INTEGRATION.md")
