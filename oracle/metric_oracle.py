"""CPU oracle for the ranking metrics of the reference's eval stage (src/evl/metric.py).  TEST INFRASTRUCTURE ONLY
(same rules as ntf_oracle.py: only tests/, smoke() and bench's cpu_baseline may import it).

The reference delegates to pytrec-eval-terrier==0.5.2 (trec_eval), which is neither vendored nor installable here.  The
functions below restate trec_eval's published definitions of P_k, recall_k, ndcg_cut_k, map_cut_k, success_k and the
reference's own `calculate_skill_coverage` (src/evl/metric.py:44-73), and are PINNED against the reference's committed
per-instance results: `output/*/toy.*/.../f0.test.pred` + `f0.test.pred.eval.instance.csv` (tests/golden/g10_metrics.npz,
produced by tests/golden/make_golden_metrics.py).
"""
from __future__ import annotations

import math

import numpy as np


def ranked_list(scores_row: np.ndarray, k: int) -> np.ndarray:
    """The run trec_eval ranks for one query: src/evl/metric.py:25-31 keeps the top-k columns as documents 'd<col>' with their
    scores; trec_eval then orders by score descending and breaks ties by document name DESCENDING (string order)."""
    k = min(k, len(scores_row))
    idx = np.argpartition(-scores_row, k - 1)[:k] if k < len(scores_row) else np.arange(len(scores_row))
    docs = sorted(idx.tolist(), key=lambda c: (-float(scores_row[c]), _neg_str("d" + str(c))))
    return np.array(docs, dtype=np.int64)


class _neg_str:
    """sort key: reverse lexicographic order of a string"""
    __slots__ = ("s",)
    def __init__(self, s): self.s = s
    def __lt__(self, o): return self.s > o.s
    def __eq__(self, o): return self.s == o.s


def trec_metrics(ranked: np.ndarray, relevant: set, cutoffs) -> dict:
    """P_k, recall_k, ndcg_cut_k, map_cut_k, success_k for binary relevance (gain 1, log2 discount)."""
    R = len(relevant)
    rel = np.array([1.0 if int(c) in relevant else 0.0 for c in ranked])
    out = {}
    for k in cutoffs:
        top = rel[:k]
        hits = float(top.sum())
        out[f"P_{k}"] = hits / k
        out[f"recall_{k}"] = hits / R if R else 0.0
        dcg = sum(g / math.log2(i + 2) for i, g in enumerate(top))
        idcg = sum(1.0 / math.log2(i + 2) for i in range(min(R, k)))
        out[f"ndcg_cut_{k}"] = dcg / idcg if idcg > 0 else 0.0
        ap, seen = 0.0, 0
        for i, g in enumerate(top):
            if g:
                seen += 1
                ap += seen / (i + 1)
        out[f"map_cut_{k}"] = ap / R if R else 0.0
        out[f"success_{k}"] = 1.0 if hits > 0 else 0.0
    return out


def skill_coverage(scores_row: np.ndarray, required_skills: np.ndarray, cov_indptr, cov_indices, cutoffs) -> dict:
    """src/evl/metric.py:63-69: experts ranked by np.argsort(row)[::-1]; coverage_k = |skills of the top-k experts ∩ required| / |required|."""
    order = np.argsort(scores_row)[::-1]
    req = set(int(s) for s in required_skills)
    out = {}
    for k in cutoffs:
        have = set()
        for e in order[:k]:
            have.update(int(s) for s in cov_indices[cov_indptr[e]:cov_indptr[e + 1]])
        out[f"skill_coverage_{k}"] = len(have & req) / len(req)
    return out


def instance_table(y_pred, truth_indptr, truth_indices, skill_indptr, skill_indices, cov_indptr, cov_indices, cutoffs=(2, 5, 10), topK=None):
    """Rows = test instances, columns in the reference's order: the five trec families (each over the cutoffs), then skill coverage."""
    n, M = y_pred.shape
    k_first = min(topK, M) if topK else M
    cols = [f"{m}_{k}" for m in ("P", "recall", "ndcg_cut", "map_cut", "success") for k in cutoffs] + [f"skill_coverage_{k}" for k in cutoffs]
    table = np.zeros((n, len(cols)))
    for i in range(n):
        rel = set(int(c) for c in truth_indices[truth_indptr[i]:truth_indptr[i + 1]])
        d = trec_metrics(ranked_list(y_pred[i], k_first), rel, cutoffs)
        d.update(skill_coverage(y_pred[i], skill_indices[skill_indptr[i]:skill_indptr[i + 1]], cov_indptr, cov_indices, cutoffs))
        table[i] = [d[c] for c in cols]
    return cols, table
