"""Golden vectors for the member-skill co-occurrence -> g11_cooc.npz.  Run in the BUILD container only (reads
/root/reference; listed in .gpurunignore):

  * the four toy datasets the reference commits: inputs = its `teamsvecs.pkl` + `splits.f3.r0.85.pkl` (test teams are skipped,
    src/main.py:98), expected = its own committed `splits.f3.r0.85/skillcoverage.pkl` (written by `Team.gen_skill_coverage`);
  * synthetic cases run through the reference's expression itself (src/cmn/team.py:327-335: deep-copied lil matrices, skipped rows
    emptied, `scipy.sparse.csr_matrix(np.dot(member.transpose(), skill))`), built so that counts pass 255 and hit exact multiples of 256.
"""
import copy
import os
import pickle

import numpy as np
import scipy.sparse

REF = "/root/reference/output"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g11_cooc.npz")


def csr_parts(mat):
    m = scipy.sparse.csr_matrix(mat); m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data


def reference_expression(member, skill, skipteams):
    member, skill = copy.deepcopy(member), copy.deepcopy(skill)
    if skipteams is not None:
        for i in skipteams:
            member.rows[i] = []; member.data[i] = []
            skill.rows[i] = []; skill.data[i] = []
    return scipy.sparse.csr_matrix(np.dot(member.transpose(), skill))


def main():
    out = {}
    names = []
    for tag, ds in [("dblp", "dblp/toy.dblp.v12.json"), ("imdb", "imdb/toy.title.basics.tsv"), ("uspt", "uspt/toy.patent.tsv"), ("gith", "gith/toy.repos.csv")]:
        tv = pickle.load(open(f"{REF}/{ds}/teamsvecs.pkl", "rb"))
        spl = pickle.load(open(f"{REF}/{ds}/splits.f3.r0.85.pkl", "rb"))
        cov = pickle.load(open(f"{REF}/{ds}/splits.f3.r0.85/skillcoverage.pkl", "rb"))
        again = reference_expression(tv["member"], tv["skill"], spl["test"])          # the committed file is what the expression gives here
        assert (again != cov).nnz == 0 and again.dtype == cov.dtype == np.uint8
        names.append(tag)
        out[f"{tag}.m_indptr"], out[f"{tag}.m_indices"], _ = csr_parts(tv["member"])
        out[f"{tag}.s_indptr"], out[f"{tag}.s_indices"], _ = csr_parts(tv["skill"])
        out[f"{tag}.shape"] = np.array([tv["member"].shape[0], tv["member"].shape[1], tv["skill"].shape[1]])
        out[f"{tag}.skip"] = np.asarray(spl["test"], np.int64)
        out[f"{tag}.c_indptr"], out[f"{tag}.c_indices"], out[f"{tag}.c_data"] = csr_parts(cov)
    rng = np.random.default_rng(11)
    for tag, (n, M, S, hot) in {"wrap": (1400, 40, 30, True), "rand": (3000, 500, 200, False)}.items():
        member = scipy.sparse.lil_matrix((n, M), dtype=np.uint8); skill = scipy.sparse.lil_matrix((n, S), dtype=np.uint8)
        for i in range(n):
            lo = 3 if hot else 0                                        # columns 0-2 are reserved for the wrap-around pairs
            for c in lo + rng.choice(M - lo, 1 + rng.integers(0, 4), replace=False): member[i, c] = 1
            for c in lo + rng.choice(S - lo, 1 + rng.integers(0, 6), replace=False): skill[i, c] = 1
            if hot:
                member[i, 0] = 1; skill[i, 0] = 1                      # pair (0,0) in every team: 1400 - skipped -> wraps 5 times
                if i < 512: member[i, 1] = 1; skill[i, 1] = 1          # exactly 512 = 0 mod 256 (unless skipped rows fall inside: see below)
                if i < 300: member[i, 2] = 1; skill[i, 2] = 1
        skip = np.arange(600, 600 + 37, dtype=np.int64) if hot else rng.choice(n, 200, replace=False).astype(np.int64)
        cov = reference_expression(member, skill, skip)
        if hot:
            d = cov.toarray()
            assert d[1, 1] == 0 and d[0, 0] == (1400 - 37) % 256 and d[2, 2] == 300 - 256
        names.append(tag)
        out[f"{tag}.m_indptr"], out[f"{tag}.m_indices"], _ = csr_parts(member)
        out[f"{tag}.s_indptr"], out[f"{tag}.s_indices"], _ = csr_parts(skill)
        out[f"{tag}.shape"] = np.array([n, M, S]); out[f"{tag}.skip"] = skip
        out[f"{tag}.c_indptr"], out[f"{tag}.c_indices"], out[f"{tag}.c_data"] = csr_parts(cov)
    out["names"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes", names)


if __name__ == "__main__":
    main()
