"""`tNtf`: the reference's streaming trainer (src/mdl/tntf.py:7-47): for each year interval (all but the last
`step_ahead`), K-fold that interval's teams, point the inner model at `{output}/{year}` and fine-tune it from the
previous interval's `f{k}.pt`.  Each interval gets a fresh Adam / scheduler / early stopper (because Fnn.learn
creates them); only weights carry over."""
from __future__ import annotations

import logging
import os
import pickle

import numpy as np

from .ntf import Ntf, cfg_get

log = logging.getLogger(__name__)


def make_tntf(base):
    class tNtf(base):
        def __init__(self, output, device, seed, cgf, model, year_idx):
            super().__init__(output, device, seed, cgf)
            self.model = model
            self.year_idx = year_idx
            self.output = self.model.output

        def name(self): return ""

        def learn(self, teamsvecs, splits, prev_model):
            from sklearn.model_selection import KFold
            done = [int(item) for item in os.listdir(self.model.output) if item.isdigit()]
            step_ahead = int(cfg_get(self.cfg, "step_ahead"))
            for i, v in enumerate(self.year_idx[:-step_ahead]):  # the last intervals are the test set
                if len(done) > 1:  # resume: this year was trained by an earlier run (tntf.py:22-26)
                    log.info(f"The model has already been trained on year {min(done)}")
                    done.remove(min(done))
                    continue
                train = np.arange(self.year_idx[i][0], self.year_idx[i + 1][0])
                skf = KFold(n_splits=int(cfg_get(self.cfg, "tfolds")), random_state=self.seed, shuffle=True)
                for k, (tr, va) in enumerate(skf.split(train)):
                    splits["folds"][k]["train"] = train[tr]
                    splits["folds"][k]["valid"] = train[va]
                self.model.output = f"{self.output}/{self.year_idx[i][1]}"
                if not os.path.isdir(self.model.output):
                    os.makedirs(self.model.output)
                with open(f"{self.model.output}/splits.pkl", "wb") as f:
                    pickle.dump(splits, f)
                self.model.learn(teamsvecs, splits, prev_model)
                prev_model = {k: f"{self.model.output}/f{k}.pt" for k in splits["folds"].keys()}

        def test(self, teamsvecs, splits, testcfg): self.model.test(teamsvecs, splits, testcfg)

        def evaluate(self, teamsvecs, splits, evalcfg): self.model.evaluate(teamsvecs, splits, evalcfg)

        def adila(self, teamsvecs, splits, faircfg): self.model.adila(teamsvecs, splits, faircfg)

    tNtf.__qualname__ = "tNtf"
    return tNtf


tNtf = make_tntf(Ntf)
