"""CPU restatement of `Team.gen_skill_coverage` (/root/reference/src/cmn/team.py:302-337) — TEST INFRASTRUCTURE ONLY (imported by
tests/ and nothing else; the product path is opentf_amd/csrc/ntf_cooc.hip).

The reference computes `scipy.sparse.csr_matrix(np.dot(member.transpose(), skill))` on two uint8 lil matrices after emptying the rows
listed in `skipteams` (team.py:327-331).  scipy's csr_matmat accumulates in the operands' dtype, so a count wraps modulo 256, and it
does not store sums that end up 0.  Restated with numpy: expand every kept team into its (member, skill) pairs, count equal pairs.

Pinned (tests/test_oracle_golden.py::test_cooc_oracle_*) on the four `skillcoverage.pkl` files the reference's authors committed under
output/*/toy.*/splits.f3.r0.85/ together with their `teamsvecs.pkl` / `splits.f3.r0.85.pkl`, and on outputs of the reference's own
expression run in the build container on wrap-around cases (tests/golden/make_golden_cooc.py -> g11_cooc.npz).
"""
import numpy as np


def skill_cooccurrence(m_indptr, m_indices, s_indptr, s_indices, n_members, n_skills, skipteams=None):
    """-> (indptr int64 [n_members+1], indices int32 ascending per row, data uint8)."""
    n = len(m_indptr) - 1
    keep = np.ones(n, bool)
    if skipteams is not None and len(skipteams):
        keep[np.asarray(skipteams, np.int64)] = False
    nm = np.diff(m_indptr) * keep
    ns = np.diff(s_indptr) * keep
    # pair p of team t: member j = p // ns[t], skill k = p % ns[t]
    pairs = nm * ns
    team = np.repeat(np.arange(n), pairs)
    local = np.arange(int(pairs.sum())) - np.repeat(np.cumsum(pairs) - pairs, pairs)
    mem = np.asarray(m_indices, np.int64)[np.asarray(m_indptr)[team] + local // np.maximum(ns[team], 1)]
    skl = np.asarray(s_indices, np.int64)[np.asarray(s_indptr)[team] + local % np.maximum(ns[team], 1)]
    key, cnt = np.unique(mem * n_skills + skl, return_counts=True)
    cnt = (cnt % 256).astype(np.uint8)                 # uint8 accumulation
    key = key[cnt != 0]; cnt = cnt[cnt != 0]           # zero sums are not stored
    rows = key // n_skills
    indptr = np.zeros(n_members + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n_members), out=indptr[1:])
    return indptr, (key % n_skills).astype(np.int32), cnt
