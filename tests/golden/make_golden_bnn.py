#!/usr/bin/env python3
"""Fixture g12: the Bnn artefacts the reference's authors COMMITTED (real bayesian-torch outputs).

Runs only in the build container (reads /root/reference/output).  Emits data only: for every toy
dataset's sparse-input `bnn.*` run directory — each checkpoint's state_dict (mu/rho), epoch, t_loss,
v_loss, the matching `.pred` file's dense MC-mean `y_pred` [n_test, M] and its `uncertainty`
{'pred': predictive entropy, 'model': mutual information} — together with the dataset's CSR
matrices and splits.  These are the only values in the reference tree that bayesian-torch 0.5.0
itself produced (src/mdl/bnn.py:15-27, src/mdl/fnn.py:136,149,158-161,202-218), so they pin the
oracle's Flipout / KL / entropy restatement (tests/test_bnn_committed.py).

    python tests/golden/make_golden_bnn.py
"""
import os
import pickle
import re

import numpy as np
import scipy.sparse
import torch

REF = "/root/reference/output"
HERE = os.path.dirname(os.path.abspath(__file__))
RUNS = {"dblp": "dblp/toy.dblp.v12.json", "imdb": "imdb/toy.title.basics.tsv", "gith": "gith/toy.repos.csv", "uspt": "uspt/toy.patent.tsv"}
BNN = "bnn.b1000.e100.ns5.lr0.001.es5.h[128].spe10.lbce.tpw10.tnw1.nsdunigram_b.nmc10"


class _Stub:  # stands in for the omegaconf classes pickled inside the checkpoints' 'cfg'
    def __init__(self, *a, **k): pass
    def __setstate__(self, st): self.__dict__["_st"] = st


class _U(pickle.Unpickler):
    def find_class(self, mod, name):
        if mod.startswith("omegaconf"): return _Stub
        return super().find_class(mod, name)


class _PM:
    __name__ = "stubpickle"
    Unpickler = _U
    load = staticmethod(pickle.load)


def tload(f): return torch.load(f, map_location="cpu", weights_only=False, pickle_module=_PM)


def main():
    arrs = {}
    runs = []
    for ds, path in RUNS.items():
        with open(f"{REF}/{path}/teamsvecs.pkl", "rb") as f: tv = pickle.load(f)
        with open(f"{REF}/{path}/splits.f3.r0.85.pkl", "rb") as f: sp = pickle.load(f)
        s, m = scipy.sparse.csr_matrix(tv["skill"]), scipy.sparse.csr_matrix(tv["member"])
        assert set(np.unique(s.data)) <= {1} and set(np.unique(m.data)) <= {1}
        arrs.update({f"{ds}.skill_indptr": s.indptr, f"{ds}.skill_indices": s.indices, f"{ds}.member_indptr": m.indptr,
                     f"{ds}.member_indices": m.indices, f"{ds}.shape": np.array([s.shape[0], s.shape[1], m.shape[1]]), f"{ds}.test": sp["test"]})
        for k, v in sp["folds"].items():
            arrs[f"{ds}.train{k}"] = v["train"]; arrs[f"{ds}.valid{k}"] = v["valid"]
        d = f"{REF}/{path}/splits.f3.r0.85/{BNN}"
        for fn in sorted(os.listdir(d)):
            mt = re.fullmatch(r"f(\d+)(\.e(\d+))?\.pt", fn)
            if not mt: continue
            fold, ep = int(mt.group(1)), mt.group(3)
            tag = f"{ds}.f{fold}" + (f".e{ep}" if ep is not None else "")
            ck = tload(f"{d}/{fn}")
            assert ck["f"] == fold
            pr = tload(f"{d}/f{fold}.test." + (f"e{ep}." if ep is not None else "") + "pred")
            yp = pr["y_pred"]
            assert not yp.is_sparse and len(pr["uncertainty"]["pred"]) == 1  # toy test sets are a single batch
            for k, v in ck["model_state_dict"].items(): arrs[f"{tag}.p.{k}"] = v.numpy()
            arrs[f"{tag}.e"] = ck["e"]; arrs[f"{tag}.t_loss"] = ck["t_loss"]; arrs[f"{tag}.v_loss"] = ck["v_loss"]
            arrs[f"{tag}.y_pred"] = yp.numpy()
            arrs[f"{tag}.unc_pred"] = pr["uncertainty"]["pred"][0]; arrs[f"{tag}.unc_model"] = pr["uncertainty"]["model"][0]
            runs.append(tag)
    arrs["runs"] = np.array(runs)
    np.savez_compressed(f"{HERE}/g12_bnn_committed.npz", **arrs)
    print(len(runs), "checkpoints:", runs)
    print({k: arrs[f"{k}.shape"] for k in RUNS})


if __name__ == "__main__":
    main()
