"""Synthetic teams with the shape statistics of the reference's datasets.

The real teamsvecs.pkl files are not in the reference tree and cannot be fetched, so benchmarks run on
CSR matrices generated directly: per-row nnz = 1 + Poisson(mean - 1), column ids from a Zipf(1.3)-like
popularity (heavy tail, as in the reference's log-log stats plots), duplicates within a row dropped, rows
sorted.  Deterministic in `seed`.
"""
from __future__ import annotations

import numpy as np

# name -> (teams N, skills S, experts M, mean skills/team, mean experts/team); output/*/prep.teamsvecs.log, data/*/readme.md
SHAPES = {
    "dblp": (1_995_708, 90_671, 233_629, 8.57, 3.06),        # dblp.v12.json.mt10.ts2
    "dblp_full": (4_877_383, 132_334, 5_022_955, 8.57, 3.06),
    "uspt_full": (7_068_508, 241_961, 3_508_807, 6.29, 2.51),    # output/uspt/patent.tsv/prep.teamsvecs.log:34
    "uspt": (2_596_322, 213_317, 394_187, 6.29, 2.51),       # patent.tsv.mt10.ts2
    "gith": (612_119, 486, 1_369_895, 1.37, 5.53),           # repos.csv (unfiltered)
    "imdb": (186_385, 27, 44_774, 1.54, 1.88),
}


def zipf_csr(n_rows: int, n_cols: int, mean_nnz: float, seed: int, alpha: float = 1.3, shift: float = 30.0):
    """CSR (int64 indptr, int32 sorted indices) with per-row nnz = min(1 + Poisson(mean-1), n_cols) and column
    popularity ~ 1/(rank + shift)^alpha (Zipf-Mandelbrot: Zipf(1.3) tail with a flattened head, so that the most
    popular expert sits in ~1% of the teams rather than in half of them).  Columns are distinct within a row:
    duplicates are dropped and the row is topped up with fresh draws."""
    rng = np.random.default_rng(seed)
    target = np.minimum(1 + rng.poisson(max(mean_nnz - 1.0, 0.0), n_rows), n_cols).astype(np.int64)
    w = 1.0 / (np.arange(1, n_cols + 1, dtype=np.float64) + shift) ** alpha
    cdf = np.cumsum(w); cdf /= cdf[-1]
    perm = rng.permutation(n_cols)  # popularity rank -> column id
    key = np.empty(0, dtype=np.int64)
    have = np.zeros(n_rows, dtype=np.int64)
    for _ in range(12):
        need = target - have
        total = int(need.sum())
        if total == 0:
            break
        cols = perm[np.searchsorted(cdf, rng.random(total), side="right").clip(0, n_cols - 1)].astype(np.int64)
        row_of = np.repeat(np.arange(n_rows, dtype=np.int64), need)
        key = np.unique(np.concatenate([key, row_of * n_cols + cols]))  # sorted by (row, col), in-row duplicates dropped
        have = np.bincount(key // n_cols, minlength=n_rows)
    indptr = np.concatenate([[0], np.cumsum(have)]).astype(np.int64)
    return indptr, (key % n_cols).astype(np.int32)


def make_dataset(name: str = "dblp", d: int = 128, seed: int = 0, n_rows: int | None = None, n_experts: int | None = None):
    N, S, M, ms, mm = SHAPES[name]
    if n_rows:
        N = int(n_rows)
    if n_experts:
        M = int(n_experts)
    s_indptr, s_indices = zipf_csr(N, S, ms, seed * 2 + 1)
    m_indptr, m_indices = zipf_csr(N, M, mm, seed * 2 + 2)
    table = np.random.default_rng(seed * 2 + 3).standard_normal((S, d), dtype=np.float32)
    return {"N": N, "S": S, "M": M, "d": d, "skill": (s_indptr, s_indices), "member": (m_indptr, m_indices), "table": table}


def init_params(dims, bayesian: bool, seed: int):
    """Random-init weights of the reference's architecture: Xavier-uniform weights + nn.Linear default bias
    (src/mdl/fnn.py:20-23), or N(0,0.1)/N(-3,0.1) for Flipout (bayesian-torch init_parameters)."""
    rng = np.random.default_rng(seed + 1000)
    sd = {}
    for l in range(len(dims) - 1):
        i, o = dims[l], dims[l + 1]
        if bayesian:
            sd[f"layers.{l}.mu_weight"] = (0.1 * rng.standard_normal((o, i))).astype(np.float32)
            sd[f"layers.{l}.rho_weight"] = (-3.0 + 0.1 * rng.standard_normal((o, i))).astype(np.float32)
            sd[f"layers.{l}.mu_bias"] = (0.1 * rng.standard_normal(o)).astype(np.float32)
            sd[f"layers.{l}.rho_bias"] = (-3.0 + 0.1 * rng.standard_normal(o)).astype(np.float32)
        else:
            a = np.sqrt(6.0 / (i + o))
            sd[f"layers.{l}.weight"] = rng.uniform(-a, a, (o, i)).astype(np.float32)
            sd[f"layers.{l}.bias"] = rng.uniform(-1 / np.sqrt(i), 1 / np.sqrt(i), o).astype(np.float32)
    return sd
