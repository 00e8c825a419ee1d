"""`tNtf`: streaming (temporal) training over year intervals — the standalone counterpart of the reference's wrapper
(src/mdl/tntf.py:7-44), needed wherever this package runs WITHOUT the reference tree (the GPU box, `bench.py`, the tests);
mounted inside the reference tree the reference's own `mdl.tntf.tNtf` wraps the plugin unchanged.

Same contract on disk and towards the caller (pinned by tests/golden/g13_tntf_dblp.npz, a run of the reference's class):
one directory per interval `{output}/{year}` holding `splits.pkl` + the inner model's files; interval i is K-folded with
`KFold(tfolds, shuffle=True, random_state=seed)` over `arange(year_idx[i][0], year_idx[i+1][0])` and written INTO the caller's
`splits['folds']` (the reference mutates it, tntf.py:30-31); the last `step_ahead` intervals are never trained on; every
interval warm-starts from the previous one's `f{k}.pt`; a directory that already holds more than one year resumes past the
earliest ones (tntf.py:22-26); `inner.output` is left pointing at the last trained interval, which is where test() reads.

What differs is where the data lives: the inner model's engine — skill / member CSR, embedding table, staging buffers — is
created once and stays resident in HBM across all intervals (`keep_engine`), instead of being rebuilt and re-uploaded per
interval as `learn()` called in a loop would do; only the K x tfolds row-id lists change between intervals.
"""
from __future__ import annotations

import logging
import os
import pickle

import numpy as np

from .ntf import Ntf, cfg_get, dist_rank

log = logging.getLogger(__name__)


def interval_folds(year_idx, i, tfolds, seed):
    """[(train_ids, valid_ids)] * tfolds of interval i (team ids are contiguous per year: year_idx = [(first_team_id, year), ...])"""
    from sklearn.model_selection import KFold
    ids = np.arange(year_idx[i][0], year_idx[i + 1][0])
    return [(ids[tr], ids[va]) for tr, va in KFold(n_splits=int(tfolds), random_state=seed, shuffle=True).split(ids)]


def make_tntf(base):
    class tNtf(base):
        def __init__(self, output, device, seed, cgf, model, year_idx):
            super().__init__(output, device, seed, cgf)
            self.model = model
            self.year_idx = year_idx
            self.output = self.model.output

        def name(self): return ""

        def intervals(self):
            """indices of the intervals that are trained on: all but the last `step_ahead` (those are the test set)"""
            return range(len(self.year_idx) - int(cfg_get(self.cfg, "step_ahead")))

        def learn(self, teamsvecs, splits, prev_model):
            trained = sorted(int(d) for d in os.listdir(self.model.output) if d.isdigit())   # year directories of an earlier run
            can_keep = hasattr(self.model, "release_engine")
            if can_keep:
                self.model.keep_engine = True
            try:
                for i in self.intervals():
                    year = self.year_idx[i][1]
                    if len(trained) > 1:   # resume: skip as many intervals as there are finished years but the last one
                        log.info(f"The model has already been trained on year {trained[0]}")
                        trained.pop(0)
                        continue
                    for k, (tr, va) in enumerate(interval_folds(self.year_idx, i, cfg_get(self.cfg, "tfolds"), self.seed)):
                        splits["folds"][k]["train"], splits["folds"][k]["valid"] = tr, va
                    self.model.output = f"{self.output}/{year}"
                    os.makedirs(self.model.output, exist_ok=True)
                    if dist_rank() == 0:
                        with open(f"{self.model.output}/splits.pkl", "wb") as f:
                            pickle.dump(splits, f)
                    self.model.learn(teamsvecs, splits, prev_model)   # fresh Adam / scheduler / early stopper per interval, weights carried over
                    prev_model = {k: f"{self.model.output}/f{k}.pt" for k in splits["folds"].keys()}
            finally:
                if can_keep:
                    self.model.keep_engine = False
                    self.model.release_engine()

        def test(self, teamsvecs, splits, testcfg): self.model.test(teamsvecs, splits, testcfg)

        def evaluate(self, teamsvecs, splits, evalcfg): self.model.evaluate(teamsvecs, splits, evalcfg)

        def adila(self, teamsvecs, splits, faircfg): self.model.adila(teamsvecs, splits, faircfg)

    tNtf.__qualname__ = "tNtf"
    return tNtf


tNtf = make_tntf(Ntf)
