#!/usr/bin/env python3
"""Golden vectors for the ranking metrics (reference src/evl/metric.py, run through pytrec_eval by the reference's authors):
the COMMITTED prediction files of the reference's toy runs (`output/*/toy.*/splits.f3.r0.85/{fnn,bnn,rnd}*/f0.test.pred`) with the
COMMITTED per-instance results next to them (`f0.test.pred.eval.instance.csv`).  Data only: y_pred, the truth rows, the skill rows,
the skill-coverage matrix and the expected metric table.  Runs only in the build container (/root/reference present).

    python tests/golden/make_golden_metrics.py
"""
import glob
import os
import pickle

import numpy as np
import pandas as pd
import scipy.sparse
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Stub:
    def __init__(self, *a, **k): pass
    def __setstate__(self, st): pass


class _U(pickle.Unpickler):
    def find_class(self, mod, name):
        return _Stub if mod.startswith("omegaconf") else super().find_class(mod, name)


class _PM:
    __name__ = "stubpickle"; Unpickler = _U; load = staticmethod(pickle.load)


def main():
    arrs, names = {}, []
    for ds in sorted(glob.glob(f"{REF}/output/*/toy.*")):
        if not os.path.exists(f"{ds}/teamsvecs.pkl") or not os.path.exists(f"{ds}/splits.f3.r0.85.pkl"): continue
        tv = pickle.load(open(f"{ds}/teamsvecs.pkl", "rb")); sp = pickle.load(open(f"{ds}/splits.f3.r0.85.pkl", "rb"))
        scp = f"{ds}/splits.f3.r0.85/skillcoverage.pkl"
        if not os.path.exists(scp): continue
        sc = scipy.sparse.csr_matrix(pickle.load(open(scp, "rb")))
        member, skill = scipy.sparse.csr_matrix(tv["member"]), scipy.sparse.csr_matrix(tv["skill"])
        for mdl in sorted(glob.glob(f"{ds}/splits.f3.r0.85/*") + glob.glob(f"{ds}/splits.f3.r0.85/*/*")):
            base = os.path.basename(mdl)
            if not base.startswith(("fnn.", "bnn.", "rnd.")) or not os.path.isdir(mdl): continue
            pred, inst = f"{mdl}/f0.test.pred", f"{mdl}/f0.test.pred.eval.instance.csv"
            if not (os.path.exists(pred) and os.path.exists(inst)): continue
            y = torch.load(pred, map_location="cpu", pickle_module=_PM, weights_only=False)["y_pred"]
            y = (y.to_dense() if y.is_sparse else y).numpy().astype(np.float32)
            df = pd.read_csv(inst)
            if len(df) != len(sp["test"]) or y.shape != (len(sp["test"]), member.shape[1]): continue
            parent = os.path.basename(os.path.dirname(mdl))
            name = f"{os.path.basename(os.path.dirname(ds))}.{base.split('.')[0]}" + ("" if parent.startswith("splits") else "." + parent.split(".")[0])
            Y, X = member[sp["test"]], skill[sp["test"]]
            arrs[f"{name}.y_pred"] = y
            arrs[f"{name}.truth_indptr"], arrs[f"{name}.truth_indices"] = Y.indptr, Y.indices
            arrs[f"{name}.skill_indptr"], arrs[f"{name}.skill_indices"] = X.indptr, X.indices
            arrs[f"{name}.cov_indptr"], arrs[f"{name}.cov_indices"] = sc.indptr, sc.indices
            arrs[f"{name}.shape"] = np.array([len(sp["test"]), member.shape[1], skill.shape[1]])
            arrs[f"{name}.expected"] = df.values.astype(np.float64)
            arrs[f"{name}.columns"] = np.array(list(df.columns))
            names.append(name)
    arrs["names"] = np.array(names)
    np.savez_compressed(f"{HERE}/g10_metrics.npz", **arrs)
    print("cases:", names)


if __name__ == "__main__":
    main()
