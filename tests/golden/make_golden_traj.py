#!/usr/bin/env python3
"""Fixture g16: WHERE TRAINING LANDS, as distributions over seeds (missing #2).

Runs only in the build container (imports /root/reference through make_golden.py's shims).  Emits data only.

(a) The reference's own `Fnn.learn` + `Fnn.test` (src/mdl/fnn.py:78-219) on toy dblp for nsd in {uniform, unigram, unigram_b} over 20 seeds: per seed and fold
    the early-stop epoch, final t_loss / v_loss (the checkpoint's), and the test-set mean of y_pred over the positives and over the rest.  A seed fixes the initial
    weights and the batch order on both sides (the plugin consumes torch's CPU generator as the reference does); what differs is the stream of sampled negatives -
    torch's CPU `rand_like` / `multinomial` there, the device generators here - so the two sides are two samples of one distribution if the device samplers are
    right ACROSS steps and epochs (a per-step distribution test cannot see a counter that repeats).
(b) What the reference's authors committed for their Bnn runs (bayesian-torch, `bnn.b1000.e100.ns5.lr0.001.es5.h[128].spe10.lbce.tpw10.tnw1.nsdunigram_b.nmc10`) on
    the four toy datasets: `test.pred.eval.mean.csv` (mean / std over the three folds of every metric).  The checkpoints' (e, t_loss, v_loss) are in g12 already.

    python tests/golden/make_golden_traj.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG   # noqa: E402  (install_shims, Cfg)

REF = "/root/reference"
SEEDS = 20
CFG = dict(b=5, e=20, ns=3, lr=0.01, es=3, h=[16], spe=0, l="bce", tpw=10, tnw=1)
RUNS = {"dblp": "dblp/toy.dblp.v12.json", "imdb": "imdb/toy.title.basics.tsv", "gith": "gith/toy.repos.csv", "uspt": "uspt/toy.patent.tsv"}
BNN = "bnn.b1000.e100.ns5.lr0.001.es5.h[128].spe10.lbce.tpw10.tnw1.nsdunigram_b.nmc10"


def main():
    import pickle
    SW, _lr_log = MG.install_shims()
    from mdl.fnn import Fnn
    with open(f"{REF}/output/dblp/toy.dblp.v12.json/teamsvecs.pkl", "rb") as f: dblp = pickle.load(f)
    with open(f"{REF}/output/dblp/toy.dblp.v12.json/splits.f3.r0.85.pkl", "rb") as f: dblp_sp = pickle.load(f)
    y_test = np.asarray(dblp["member"][dblp_sp["test"]].todense()) > 0
    testcfg = MG.Cfg(per_epoch=False, on_train=False, topK=None)
    arrs = {"cfg": json.dumps(CFG), "seeds": np.arange(SEEDS)}
    for nsd in ["uniform", "unigram", "unigram_b"]:
        e = np.zeros((SEEDS, 3), np.int64); tl = np.zeros((SEEDS, 3)); vl = np.zeros((SEEDS, 3)); pp = np.zeros((SEEDS, 3)); pn = np.zeros((SEEDS, 3))
        curves = np.full((2, SEEDS, 3, CFG["e"]), np.nan)          # [t | v][seed][fold][epoch]: the per-epoch series the reference logs (fnn.py:154-155), NaN after the stop
        for seed in range(SEEDS):
            SW.scalars.clear()
            tmp = tempfile.mkdtemp(prefix="traj_")
            m = Fnn(tmp, "cpu", seed, MG.Cfg({**CFG, "nsd": nsd}))
            sp = {"test": dblp_sp["test"], "folds": {k: dict(v) for k, v in dblp_sp["folds"].items()}}
            m.learn(dblp, sp, None)
            for tag, v, step in SW.scalars:
                k, which = tag.split("_", 1)
                curves[0 if which == "t_loss" else 1, seed, int(k), step] = v
            m.test(dblp, sp, testcfg)
            for k in sp["folds"]:
                ck = torch.load(f"{m.output}/f{k}.pt")
                e[seed, k], tl[seed, k], vl[seed, k] = ck["e"], ck["t_loss"], ck["v_loss"]
                yp = torch.load(f"{m.output}/f{k}.test.pred")["y_pred"].numpy()
                pp[seed, k], pn[seed, k] = yp[y_test].mean(), yp[~y_test].mean()
        arrs[f"fnn.{nsd}.curves"] = curves
        arrs.update({f"fnn.{nsd}.e": e, f"fnn.{nsd}.t_loss": tl, f"fnn.{nsd}.v_loss": vl, f"fnn.{nsd}.pred_pos": pp, f"fnn.{nsd}.pred_neg": pn})
        print(nsd, "stop epoch mean %.2f  v_loss %.4f +- %.4f  pred pos %.4f neg %.4f" % (e.mean(), vl.mean(), vl.std(), pp.mean(), pn.mean()))
    import pandas as pd
    for ds, path in RUNS.items():
        df = pd.read_csv(f"{REF}/output/{path}/splits.f3.r0.85/{BNN}/test.pred.eval.mean.csv", index_col=0)
        arrs[f"bnn.{ds}.metrics"] = np.array(list(df.index)); arrs[f"bnn.{ds}.mean"] = df["mean"].to_numpy(np.float64); arrs[f"bnn.{ds}.std"] = df["std"].to_numpy(np.float64)
        print(ds, dict(zip(df.index[:4], df["mean"][:4])))
    np.savez_compressed(f"{HERE}/g16_traj.npz", **arrs)
    print("wrote g16_traj.npz")


if __name__ == "__main__":
    main()
