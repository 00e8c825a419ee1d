"""Mirror of the reference's `evl.metric` (src/evl/metric.py) with the per-instance work on the MI355X.

Same function names, arguments and returned DataFrames as the reference (`calculate_metrics` 5-35, `calculate_skill_coverage`
44-73); the reference builds python dicts for pytrec_eval row by row, here the ranked top-K lists go through
`ntf_rank_metrics` / `ntf_skill_coverage` of libopentf_amd.so.  `calculate_auc_roc` stays sklearn on the host, as in the reference.
Ties in the scores are ranked by ascending expert id (trec_eval: descending document name) — irrelevant for real-valued model
outputs, stated here because it is the one place the two can differ.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

TREC = ("P", "recall", "ndcg_cut", "map_cut", "success")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ranked_topk(Y_, k):
    """[n, k] expert ids by decreasing score (stable: ascending id among equal scores), from a dense array or a scipy CSR.  Sparse input (the
    top-K `.pred` files): ONE lexsort over all stored entries, no per-row Python work; only rows that store fewer than k entries (never the case
    for files test() wrote with topK >= k) are completed one by one with the lowest-id unstored experts, which all score 0."""
    n = Y_.shape[0]
    if sp.issparse(Y_):
        Y_ = sp.csr_matrix(Y_)
        cnt = np.diff(Y_.indptr)
        rows = np.repeat(np.arange(n, dtype=np.int64), cnt)
        order = np.lexsort((Y_.indices, -Y_.data, rows))                 # by row, then score descending, then expert id
        pos = np.arange(len(order), dtype=np.int64) - np.repeat(Y_.indptr[:-1].astype(np.int64), cnt)   # rank inside the row
        keep = pos < k
        out = np.zeros((n, k), dtype=np.int32)
        out[rows[keep], pos[keep]] = Y_.indices[order][keep]
        for i in np.nonzero(cnt < k)[0]:
            cols = Y_.indices[Y_.indptr[i]:Y_.indptr[i + 1]]
            out[i, cnt[i]:] = np.setdiff1d(np.arange(Y_.shape[1], dtype=np.int64), cols)[: k - cnt[i]]
        return out
    Y_ = np.asarray(Y_)
    idx = np.argsort(-Y_, axis=1, kind="stable")[:, :k]
    return np.ascontiguousarray(idx, dtype=np.int32)


def _cutoffs(metrics, family):
    for m in metrics:
        if m.startswith(family + "_"):
            return [int(x) for x in m[len(family) + 1:].split(",")]
    return []


def calculate_metrics(Y, Y_, topK=None, per_instance=False, metrics=("P_2,5", "recall_2,5", "ndcg_cut_2,5"), device=0, ranked=None):
    """`ranked` [n, K] int32 (optional): the ranked expert ids themselves, e.g. `Engine.forward_topk`'s indices straight from the device; Y_ may
    then be None - no prediction matrix, sparse or dense, is formed or sorted on the host."""
    import pandas as pd
    from .. import libntf
    assert ranked is not None or Y.shape == Y_.shape, f"Shape mismatch between truth Y {Y.shape} vs preds Y_ {Y_.shape}!"
    Y = sp.csr_matrix(Y); Y.sort_indices()
    n, M = Y.shape
    fams = [f for f in TREC if _cutoffs(metrics, f)]
    cuts = sorted({k for f in fams for k in _cutoffs(metrics, f)})
    kmax = min(max(cuts), min(topK, M) if topK else M)
    if ranked is not None:
        top = np.ascontiguousarray(np.asarray(ranked)[:, :kmax], dtype=np.int32)
        assert top.shape == (n, kmax), f"ranked ids {np.asarray(ranked).shape} cover fewer than the {kmax} ranks the cutoffs need"
    else:
        top = _ranked_topk(Y_, kmax)
    ip, ix = np.ascontiguousarray(Y.indptr, dtype=np.int64), np.ascontiguousarray(Y.indices, dtype=np.int32)
    cu = np.ascontiguousarray(cuts, dtype=np.int32)
    out = np.zeros((n, 5 * len(cuts)), dtype=np.float32)
    rc = libntf.lib().ntf_rank_metrics(int(device), _ptr(top), n, top.shape[1], _ptr(ip), _ptr(ix), n, None, _ptr(cu), len(cuts), _ptr(out))
    if rc != 0:
        raise libntf.NtfError(f"ntf_rank_metrics failed ({rc})")
    cols, data = [], []
    for f in fams:  # the reference's column order: family by family, each over its cutoffs
        for k in _cutoffs(metrics, f):
            cols.append(f"{f}_{k}"); data.append(out[:, TREC.index(f) * len(cuts) + cuts.index(k)].astype(np.float64))
    df = pd.DataFrame(np.stack(data, axis=1), columns=cols, index=[f"q{i}" for i in range(n)])
    df_mean = df.mean().to_frame("mean").rename_axis("metrics")
    return (df if per_instance else None), df_mean


def micro_auc_sparse(Y, Y_):
    """Micro-averaged ROC AUC of a SPARSE score matrix against sparse 0/1 truth without densifying either: the Mann-Whitney statistic
    with mid-ranks, U = sum over positives of (#negatives scored lower + 0.5 #negatives scored equal), AUC = U / (P * N_neg).  All the
    entries a top-K prediction does not store are one tie group at score 0.  Equals sklearn's `roc_auc_score(Y.toarray(),
    Y_.toarray(), average='micro')` (what src/evl/metric.py:36-41 computes) — which needs the dense [n_test, M] pair."""
    Y = sp.csr_matrix(Y); Y_ = sp.csr_matrix(Y_)
    n, M = Y_.shape
    total = n * M
    P = int((Y.data != 0).sum())
    if P == 0 or P == total:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    Yb = Y.copy(); Yb.data = (Yb.data != 0).astype(np.int8); Yb.eliminate_zeros(); Yb.sort_indices()
    S = Y_.copy(); S.sum_duplicates(); S.sort_indices()
    # label of every stored score: flat keys row * M + col looked up among the positives' keys
    key_s = (np.repeat(np.arange(n, dtype=np.int64), np.diff(S.indptr)) * M + S.indices.astype(np.int64))
    key_p = (np.repeat(np.arange(n, dtype=np.int64), np.diff(Yb.indptr)) * M + Yb.indices.astype(np.int64))
    is_pos = np.isin(key_s, key_p, assume_unique=True)
    scores = S.data.astype(np.float64)
    pos_stored = int(is_pos.sum())
    # tie groups over the stored scores plus the implicit zeros
    vals, inv = np.unique(scores, return_inverse=True)
    p_g = np.bincount(inv, weights=is_pos.astype(np.float64), minlength=len(vals))
    n_g = np.bincount(inv, minlength=len(vals)).astype(np.float64) - p_g
    imp_total = total - len(scores)
    imp_pos = P - pos_stored
    z = np.searchsorted(vals, 0.0)
    if z < len(vals) and vals[z] == 0.0:
        p_g[z] += imp_pos; n_g[z] += imp_total - imp_pos
    else:
        vals = np.insert(vals, z, 0.0); p_g = np.insert(p_g, z, imp_pos); n_g = np.insert(n_g, z, imp_total - imp_pos)
    neg_below = np.concatenate([[0.0], np.cumsum(n_g)[:-1]])
    U = float(np.sum(p_g * (neg_below + 0.5 * n_g)))
    return U / (float(P) * float(total - P))


def calculate_auc_roc(Y, Y_, curve=False):
    """src/evl/metric.py:36-41.  Sparse predictions (the top-K `.pred` files) go through `micro_auc_sparse`; dense ones, and the curve
    itself, through sklearn as in the reference."""
    assert Y.shape == Y_.shape
    if sp.issparse(Y_) and not curve:
        return micro_auc_sparse(Y, Y_), None
    from sklearn import metrics as skm
    dense = Y_.toarray() if sp.issparse(Y_) else np.asarray(Y_)
    auc = skm.roc_auc_score(Y.toarray(), dense, average="micro", multi_class="ovr")
    if curve:
        fpr, tpr, _ = skm.roc_curve(Y.toarray().ravel(), dense.ravel())
        return auc, (fpr, tpr)
    return auc, None


def calculate_skill_coverage(X, Y_, expertskillvecs, per_instance=False, topks="2,5,10", device=0):
    import pandas as pd
    from .. import libntf
    assert X.shape[0] == Y_.shape[0]
    X = sp.csr_matrix(X); X.sort_indices()
    cov = sp.csr_matrix(expertskillvecs); cov.sort_indices()
    cuts = [int(k) for k in topks.split(",")]
    n, E = Y_.shape
    top = _ranked_topk(Y_, min(max(cuts), E))
    out = np.zeros((n, len(cuts)), dtype=np.float32)
    cu = np.ascontiguousarray(cuts, dtype=np.int32)
    xs, xi = np.ascontiguousarray(X.indptr, dtype=np.int64), np.ascontiguousarray(X.indices, dtype=np.int32)
    cs, ci = np.ascontiguousarray(cov.indptr, dtype=np.int64), np.ascontiguousarray(cov.indices, dtype=np.int32)
    rc = libntf.lib().ntf_skill_coverage(int(device), _ptr(top), n, top.shape[1], _ptr(xs), _ptr(xi), n, None, _ptr(cs), _ptr(ci), E, _ptr(cu), len(cuts), _ptr(out))
    if rc != 0:
        raise libntf.NtfError(f"ntf_skill_coverage failed ({rc})")
    df = pd.DataFrame(out.astype(np.float64), columns=[f"skill_coverage_{k}" for k in cuts])
    return df, df.mean().to_frame("mean").rename_axis("metrics")


class EvalSpec:
    """what an eval config asks for (src/__config__.yaml eval section), parsed once"""
    def __init__(self, topK, per_instance, trec, other):
        self.topK, self.per_instance, self.trec = topK, bool(per_instance), list(trec or [])
        other = list(other or [])
        self.auc = next((m for m in other if "aucroc" in m), None)               # 'aucroc' or 'aucroc+' (+ = keep the curve)
        self.skc = next((m for m in other if "skill_coverage" in m), None)       # 'skill_coverage_2,5,10'

    @classmethod
    def from_cfg(cls, evalcfg):
        from ..mdl.ntf import cfg_get
        m = cfg_get(evalcfg, "metrics")
        return cls(cfg_get(evalcfg, "topK"), cfg_get(evalcfg, "per_instance"), cfg_get(m, "trec"), cfg_get(m, "other"))


def score_predictions(teamsvecs, rows, Y_, spec, device=0):
    """One prediction matrix against the truth rows `rows` of teamsvecs: (per-instance table, mean table, roc curve or None).  Row order of the mean table as the
    reference writes it: trec metrics, aucroc, skill coverage (src/mdl/ntf.py:57-84)."""
    import pandas as pd
    Y = teamsvecs["member"][rows]
    assert Y.shape == Y_.shape, f"Shape mismatch between truth Y {Y.shape} vs preds Y_ {Y_.shape}!"
    inst_parts, mean_parts, roc = [], [], None
    if spec.trec:
        df, df_mean = calculate_metrics(Y, Y_, spec.topK, spec.per_instance, spec.trec, device=device)
        inst_parts.append(df); mean_parts.append(df_mean)
    if spec.auc:
        auc, roc = calculate_auc_roc(Y, Y_, curve=(spec.auc == "aucroc+"))
        mean_parts.append(pd.DataFrame({"mean": [auc]}, index=pd.Index(["aucroc"], name="metrics")))
    if spec.skc:
        X = teamsvecs["skill"] if sp.issparse(teamsvecs["skill"]) else teamsvecs["original_skill"]
        df, df_mean = calculate_skill_coverage(X[rows], Y_, teamsvecs["skillcoverage"], spec.per_instance, topks=spec.skc.replace("skill_coverage_", ""), device=device)
        inst_parts.append(df); mean_parts.append(df_mean)
    inst_parts = [d.reset_index(drop=True) for d in inst_parts if d is not None and not d.empty]
    inst = pd.concat(inst_parts, axis=1) if inst_parts else pd.DataFrame()
    mean = pd.concat(mean_parts, axis=0) if mean_parts else pd.DataFrame(columns=["mean"])
    mean.index.name = "metrics"
    return inst, mean, roc
