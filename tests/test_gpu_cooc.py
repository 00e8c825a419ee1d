"""Member-skill co-occurrence `member^T . skill` on a real MI355X (reference src/cmn/team.py:302-337) through the
C ABI (`ntf_skill_cooccurrence`): bit-exact against the reference's committed `skillcoverage.pkl` files, the reference expression's
outputs on wrap-around cases, and the oracle on shapes that exercise both row kernels (sorted short rows, histogram long rows)."""
import pickle

import numpy as np
import pytest
import scipy.sparse

from conftest import golden
from oracle import cooc_oracle as CO

pytestmark = pytest.mark.gpu


def _same(got, indptr, indices, data):
    assert got.dtype == np.uint8 and got.has_sorted_indices
    assert np.array_equal(got.indptr, indptr) and np.array_equal(got.indices, indices) and np.array_equal(got.data, data)


@pytest.mark.parametrize("name", ["dblp", "imdb", "uspt", "gith", "wrap", "rand"])
def test_cooc_bitexact_vs_reference_golden(name):
    from opentf_amd.cmn.team import skill_cooccurrence
    g = golden("g11_cooc")
    n, M, S = [int(v) for v in g[f"{name}.shape"]]
    got = skill_cooccurrence((g[f"{name}.m_indptr"], g[f"{name}.m_indices"], (n, M)), (g[f"{name}.s_indptr"], g[f"{name}.s_indices"], (n, S)), g[f"{name}.skip"])
    assert got.shape == (M, S)
    _same(got, g[f"{name}.c_indptr"], g[f"{name}.c_indices"], g[f"{name}.c_data"])


def _random_case(n, M, S, mean_m, mean_s, seed, heavy=0):
    rng = np.random.default_rng(seed)
    def csr(width, mean, force):
        nnz = 1 + rng.poisson(mean - 1, n)
        nnz[::53] = 0                                    # empty rows on either side contribute nothing
        rows = []
        for i, k in enumerate(nnz):
            c = set(rng.choice(width, min(k, width), replace=False).tolist())
            if force and i % 2 == 0 and k: c.update(range(force))   # `force` columns present in half of the teams -> very long rows
            rows.append(np.sort(np.fromiter(c, np.int32, len(c))))
        indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        return indptr, (np.concatenate(rows) if indptr[-1] else np.zeros(0, np.int32)).astype(np.int32)
    return csr(M, mean_m, heavy), csr(S, mean_s, 0)


@pytest.mark.parametrize("n,M,S,mean_m,mean_s,heavy", [(20000, 3000, 700, 3.0, 8.5, 3), (5000, 64, 5000, 2.5, 40.0, 2), (3000, 4000, 40, 5.0, 1.4, 0),
                                                         (1, 5, 7, 3.0, 4.0, 0)])
def test_cooc_bitexact_vs_oracle_short_and_long_rows(n, M, S, mean_m, mean_s, heavy):
    from opentf_amd.cmn.team import skill_cooccurrence
    (m_ip, m_ix), (s_ip, s_ix) = _random_case(n, M, S, mean_m, mean_s, seed=n + M, heavy=heavy)
    skip = np.random.default_rng(1).choice(n, n // 7, replace=False)
    ip, ix, data = CO.skill_cooccurrence(m_ip, m_ix, s_ip, s_ix, M, S, skip)
    got = skill_cooccurrence((m_ip, m_ix, (n, M)), (s_ip, s_ix, (n, S)), skip)
    _same(got, ip, ix, data)
    if heavy:   # the forced columns' rows are longer than the in-LDS sort handles: the histogram kernel produced them
        assert (np.diff(m_ip) > 0).sum() * mean_s / 2 > 2048
    got_all = skill_cooccurrence((m_ip, m_ix, (n, M)), (s_ip, s_ix, (n, S)), None)
    ip, ix, data = CO.skill_cooccurrence(m_ip, m_ix, s_ip, s_ix, M, S, None)
    _same(got_all, ip, ix, data)


def test_cooc_full_dblp_shapes_properties_and_oracle():
    """BASELINE config 2's matrices (N = 1 995 708, M = 233 629, S = 90 671, synthetic): checked against the oracle bit for bit, and through
    a size-independent property — without wrap-around the grand total equals sum_t nnz_member(t) * nnz_skill(t)."""
    from opentf_amd.cmn.team import skill_cooccurrence
    from opentf_amd.synth import make_dataset
    ds = make_dataset("dblp", d=8, seed=0)
    (m_ip, m_ix), (s_ip, s_ix) = ds["member"], ds["skill"]
    N, M, S = ds["N"], ds["M"], ds["S"]
    skip = np.arange(0, N, 7, dtype=np.int64)
    got, ms = skill_cooccurrence((m_ip, m_ix, (N, M)), (s_ip, s_ix, (N, S)), skip, return_ms=True)
    keep = np.ones(N, bool); keep[skip] = False
    ip, ix, data = CO.skill_cooccurrence(m_ip, m_ix, s_ip, s_ix, M, S, skip)
    _same(got, ip, ix, data)
    assert (np.diff(got.indptr) <= S).all()
    for r in (0, 1, M // 2, M - 1):
        seg = got.indices[got.indptr[r]:got.indptr[r + 1]]
        assert (np.diff(seg) > 0).all()
    print(f"co-occurrence of dblp shapes: nnz {got.nnz}, device {ms:.1f} ms")


def test_gen_skill_coverage_writes_and_reuses_the_reference_cache_file(tmp_path):
    """`Team.gen_skill_coverage(teamsvecs, output, skipteams)` (src/main.py:98): lil inputs as `teamsvecs.pkl` holds them, result pickled
    as `{output}/skillcoverage.pkl`, second call served from the file."""
    from opentf_amd.cmn.team import Team, lil_to_csr
    g = golden("g11_cooc")
    n, M, S = [int(v) for v in g["gith.shape"]]
    member = scipy.sparse.csr_matrix((np.ones(len(g["gith.m_indices"]), np.uint8), g["gith.m_indices"], g["gith.m_indptr"]), shape=(n, M)).tolil()
    skill = scipy.sparse.csr_matrix((np.ones(len(g["gith.s_indices"]), np.uint8), g["gith.s_indices"], g["gith.s_indptr"]), shape=(n, S)).tolil()
    ip, ix, shape = lil_to_csr(member)
    assert np.array_equal(ip, g["gith.m_indptr"]) and np.array_equal(ix, g["gith.m_indices"]) and shape == (n, M)
    tv = {"member": member, "skill": skill}
    cov = Team.gen_skill_coverage(tv, str(tmp_path / "splits"), skipteams=g["gith.skip"])
    _same(cov, g["gith.c_indptr"], g["gith.c_indices"], g["gith.c_data"])
    stored = pickle.load(open(tmp_path / "splits" / "skillcoverage.pkl", "rb"))
    assert (stored != cov).nnz == 0
    again = Team.gen_skill_coverage({"member": member, "skill": skill}, str(tmp_path / "splits"), skipteams=None)  # from the cache: skipteams not re-applied
    assert (again != cov).nnz == 0
