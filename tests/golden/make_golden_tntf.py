#!/usr/bin/env python3
"""Fixture g13: the reference's OWN streaming trainer `mdl.tntf.tNtf` (src/mdl/tntf.py:16-38) run here on toy dblp
around its own `mdl.fnn.Fnn` with nsd=None (nothing random but init / loader order, which the product reproduces
from the seed), plus the per-year `splits.pkl` the reference's authors committed for their temporal Bnn run.

Runs only in the build container.  Emits data only: per year interval the K-fold splits the reference wrote, the
per-epoch loss series, the final weights of every fold, and the test predictions of the last interval.

    python tests/golden/make_golden_tntf.py
"""
import json
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, Cfg, install_shims  # noqa: E402  (same shims as the other reference-run fixtures)


def main():
    SummaryWriter, lr_log = install_shims()
    from mdl.fnn import Fnn
    from mdl.tntf import tNtf
    ds = f"{REF}/output/dblp/toy.dblp.v12.json"
    with open(f"{ds}/teamsvecs.pkl", "rb") as f: tv = pickle.load(f)
    with open(f"{ds}/indexes.pkl", "rb") as f: i2y = pickle.load(f)["i2y"]
    tmp = tempfile.mkdtemp(prefix="golden_tntf_")
    seed, nfolds, step_ahead = 0, 3, 1
    # the temporal splits as src/main.py:22-23 makes them (main.py itself needs hydra): the ones the authors committed
    cdir = f"{ds}/splits.f3.r0.85.t1/bnn.b1000.e100.ns5.lr0.001.es5.h[128].spe10.lbce.tpw10.tnw1.nsdunigram_b.nmc10"
    with open(f"{cdir}/2000/splits.pkl", "rb") as f: splits = pickle.load(f)
    assert np.array_equal(splits["test"], np.arange(i2y[-step_ahead][0], tv["skill"].shape[0]))
    cfg = dict(b=3, e=4, ns=3, lr=0.01, es=2, h=[16], spe=0, l="bce", tpw=10, tnw=1, nsd=None)
    inner = Fnn(tmp, "cpu", seed, Cfg(cfg))
    t = tNtf(tmp, "cpu", seed, Cfg(tfolds=nfolds, step_ahead=step_ahead), inner, i2y)
    SummaryWriter.scalars.clear()
    arrs = {"cfg": json.dumps(cfg), "i2y": np.array(i2y), "seed": seed, "tfolds": nfolds, "step_ahead": step_ahead,
            "test": splits["test"], "root_name": inner.name()}
    t.learn(tv, splits, None)
    years = sorted(int(d) for d in os.listdir(t.output) if d.isdigit())
    arrs["years"] = np.array(years)
    for y in years:
        with open(f"{t.output}/{y}/splits.pkl", "rb") as f: sp = pickle.load(f)
        for k in sp["folds"]:
            arrs[f"{y}.train{k}"] = sp["folds"][k]["train"]; arrs[f"{y}.valid{k}"] = sp["folds"][k]["valid"]
            ck = torch.load(f"{t.output}/{y}/f{k}.pt")
            arrs.update({f"{y}.f{k}.{n}": v.numpy() for n, v in ck["model_state_dict"].items()})
            arrs[f"{y}.f{k}.e"] = ck["e"]; arrs[f"{y}.f{k}.t_loss"] = ck["t_loss"]; arrs[f"{y}.f{k}.v_loss"] = ck["v_loss"]
    arrs["scalars"] = json.dumps(SummaryWriter.scalars)
    # tNtf.test -> inner.test on the LAST interval's directory (inner.output is left pointing there, tntf.py:33)
    t.test(tv, splits, Cfg(per_epoch=False, on_train=False, topK=None))
    arrs["last_output_suffix"] = os.path.relpath(inner.output, t.output)
    for k in splits["folds"]:
        arrs[f"test.f{k}.y_pred"] = torch.load(f"{inner.output}/f{k}.test.pred")["y_pred"].numpy()

    # the splits the reference's authors committed for their temporal Bnn run (same seed 0, tfolds 3): pins KFold usage
    for y in sorted(int(d) for d in os.listdir(cdir) if d.isdigit()):
        with open(f"{cdir}/{y}/splits.pkl", "rb") as f: sp = pickle.load(f)
        arrs[f"committed.{y}.test"] = sp["test"]
        for k in sp["folds"]:
            arrs[f"committed.{y}.train{k}"] = sp["folds"][k]["train"]; arrs[f"committed.{y}.valid{k}"] = sp["folds"][k]["valid"]
    np.savez_compressed(f"{HERE}/g13_tntf_dblp.npz", **arrs)
    print("years", years, "last", arrs["last_output_suffix"], "keys", len(arrs))


if __name__ == "__main__":
    main()
