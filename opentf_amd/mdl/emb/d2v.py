"""`D2v` (`data.embedding.class_method=mdl.amd.emb.d2v.D2v`): the reference's doc2vec team2vec plugin (src/mdl/emb/d2v.py) with the training on the MI355X.

What the caller (src/main.py:100-153) does and what it gets here:
  * `t2v = cls(output, acceleration, seed, cfg, method)`; `t2v.learn(teamsvecs, splits)`                                       (main.py:112,122)
      teams as documents, skills / members (/ the year) as words (d2v.py:17-50), then gensim's
      `Doc2Vec(min_count=1, dbow_words=1, dm=cfg.dm, vector_size=cfg.d, window=cfg.w, min_alpha=cfg.lr, seed=seed)`, `build_vocab`, and either the per-epoch
      loop with its own alpha bookkeeping and `.e{epoch}.pt` files (cfg.spe set, d2v.py:74-83) or one `train(epochs=cfg.e)` (d2v.py:84).  Here build_vocab's
      tables are made on the host (below), the passes run in opentf_amd/csrc/ntf_d2v.hip (PV-DM / PV-DBOW with negative sampling, gensim 4.3.3's arithmetic as
      restated in oracle/d2v_oracle.py), and directory / file names are the reference's.  The files hold the three tables and the vocabulary as a torch pickle
      (marker `format`), not a gensim object: gensim is not needed to read them, and gensim cannot read them.
  * `skill_vecs = t2v.get_dense_vecs(teamsvecs, vectype='skill')`                                                                  (main.py:148)
      `model.docvecs.vectors`, row i = team i (d2v.py:110-116) - the dense [N, d] input of the Fnn / Bnn plugin.

The reference's log of this stage on dblp mt10.ts2 (output/dblp/dblp.v12.json.mt10.ts2/prep.d2v.skill.log): 224 workers, 276 s per epoch over 19 073 021 words,
100 epochs = 7.7 h; on the MI355X an epoch of that corpus is a single kernel of a fraction of a second.
"""
from __future__ import annotations

import logging
import os
import random
import re

import numpy as np
import scipy.sparse

from ..fnn import parse_devices
from ..ntf import cfg_get, dist_rank
from .t2v import T2v

log = logging.getLogger(__name__)

FORMAT = "opentf_amd.d2v.v1"
ALPHA, NEGATIVE, SAMPLE, NS_EXPONENT = 0.025, 5, 1e-3, 0.75     # gensim's defaults, which the reference's constructor call leaves untouched (d2v.py:69-71)


def _barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def _csr(mat):
    m = scipy.sparse.csr_matrix(mat)
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int64)


def team_documents(teamsvecs, embtype, time_indexes=None):
    """d2v.py:29-46 without the per-team Python loop: (doc_ptr, word ids, key of word id).  Word ids: skill j -> j, member j -> S + j ('skillmember') or j ('member'),
    year y -> S + rank of y ('skilltime').  Inside a document: skills in column order, then members in column order / the year token."""
    skill = teamsvecs.get("original_skill", teamsvecs["skill"]) if hasattr(teamsvecs, "get") else teamsvecs["skill"]
    n, S = skill.shape
    s_ip, s_ix = _csr(skill)
    if embtype == "skill":
        return s_ip, s_ix, lambda w: f"s{w}"
    if embtype in ("member", "skillmember"):
        m_ip, m_ix = _csr(teamsvecs["member"])
        if embtype == "member":
            return m_ip, m_ix, lambda w: f"m{w}"
        ptr = s_ip + m_ip
        words = np.empty(ptr[-1], dtype=np.int64)
        s_len, m_len = np.diff(s_ip), np.diff(m_ip)
        s_dst = np.repeat(ptr[:-1], s_len) + (np.arange(len(s_ix)) - np.repeat(s_ip[:-1], s_len))
        m_dst = np.repeat(ptr[:-1] + s_len, m_len) + (np.arange(len(m_ix)) - np.repeat(m_ip[:-1], m_len))
        words[s_dst] = s_ix; words[m_dst] = m_ix + S
        return ptr, words, lambda w: f"s{w}" if w < S else f"m{w - S}"
    if embtype == "skilltime":
        assert time_indexes, "Temporal skill embedding needs time indexes!"
        i2y = list(time_indexes["i2y"])                      # [(first team of the year, year)], teams sorted by year (d2v.py:28,36-40)
        starts = np.asarray([s for s, _ in i2y], dtype=np.int64)
        years = np.asarray([y for _, y in i2y])
        year_of_team = years[np.clip(np.searchsorted(starts, np.arange(n), side="right") - 1, 0, None)]
        uniq = np.unique(years)
        ptr = s_ip + np.arange(n + 1)
        words = np.empty(ptr[-1], dtype=np.int64)
        s_len = np.diff(s_ip)
        s_dst = np.repeat(ptr[:-1], s_len) + (np.arange(len(s_ix)) - np.repeat(s_ip[:-1], s_len))
        words[s_dst] = s_ix
        words[ptr[1:] - 1] = S + np.searchsorted(uniq, year_of_team)
        return ptr, words, lambda w: f"s{w}" if w < S else f"dt{uniq[w - S]}"
    raise ValueError(f"unknown embtype {embtype}")


def build_vocab(words, sample=SAMPLE, ns_exponent=NS_EXPONENT):
    """gensim's scan_vocab / prepare_vocab / make_cum_table for min_count = 1: vocabulary by descending count (equal counts in first-seen order), the uint32 keep
    thresholds of the frequent-word subsampling and the cumulative count^0.75 table of the negative draws.  -> (keys [V] word id of vocabulary index, count [V],
    sample_int [V], cum_table [V], words as vocabulary indices)"""
    words = np.asarray(words, dtype=np.int64)
    uniq, first, inv, cnt = np.unique(words, return_index=True, return_inverse=True, return_counts=True)
    seen = np.argsort(first, kind="stable")                   # first-seen order
    order = seen[np.argsort(-cnt[seen], kind="stable")]
    keys, count = uniq[order], cnt[order].astype(np.int64)
    rank = np.empty(len(uniq), dtype=np.int64); rank[order] = np.arange(len(uniq))
    total = int(count.sum())
    thr = total if not sample else (sample * total if sample < 1.0 else int(sample * (3 + np.sqrt(5)) / 2))
    p = np.minimum((np.sqrt(count / thr) + 1) * (thr / count), 1.0)
    sample_int = (p * (2 ** 32 - 1)).astype(np.uint32)
    pw = count.astype(np.float64) ** ns_exponent
    cum = np.round(np.cumsum(pw) / pw.sum() * (2 ** 31 - 1)).astype(np.uint32)
    return keys, count, sample_int, cum, rank[inv].astype(np.int32)


def job_progress(doc_ptr, order=None, batch_words=10000):
    """gensim's _job_producer: documents go greedily into jobs of at most `batch_words` raw words (at least one document); a job's alpha is fixed when it is cut, at
    (documents pushed before it) / (all documents) of the way from the pass's start alpha to its end alpha.  -> that fraction for every rank of the pass"""
    n = len(doc_ptr) - 1
    lens = np.diff(np.asarray(doc_ptr)) if order is None else np.diff(np.asarray(doc_ptr))[np.asarray(order)]
    cum = np.concatenate([[0], np.cumsum(lens)])
    prog = np.empty(n, dtype=np.float64)
    s = 0
    while s < n:
        e = max(int(np.searchsorted(cum, cum[s] + batch_words, side="right")) - 1, s + 1)
        prog[s:e] = s / n
        s = e
    return prog


def initial_vectors(n_docs, n_vocab, d, seed):
    """gensim 4's prepare_weights: (default_rng(seed).random(float32) * 2 - 1) / d for the words, the same from seed + 7919 for the doc tags"""
    def prep(shape, s):
        v = np.random.default_rng(seed=s).random(shape, dtype=np.float32)
        v *= 2.0; v -= 1.0; v /= shape[1]
        return v
    return prep((n_vocab, d), seed), prep((n_docs, d), seed + 7919)


class KeyedVectors:
    """the few attributes of gensim's KeyedVectors the reference touches (d2v.py:61,96-116)"""

    def __init__(self, keys, vectors):
        self.index_to_key = list(keys)
        self.key_to_index = {k: i for i, k in enumerate(self.index_to_key)}
        self.vectors = vectors
        self.vector_size = vectors.shape[1]

    def __len__(self): return len(self.index_to_key)

    def __getitem__(self, key): return self.vectors[self.key_to_index[key]]

    def most_similar(self, positive, topn=10):
        v = np.mean(np.asarray(positive, dtype=np.float32).reshape(-1, self.vector_size), axis=0)
        nv = self.vectors / np.maximum(np.linalg.norm(self.vectors, axis=1, keepdims=True), 1e-30)
        sims = nv @ (v / max(float(np.linalg.norm(v)), 1e-30))
        top = np.argsort(-sims)[:topn]
        return [(self.index_to_key[i], float(sims[i])) for i in top]


class Doc2VecTables:
    """what `self.model` is after learn(): .dv / .docvecs (doc tags '0'..'N-1'), .wv (words), .syn1neg and the hyper-parameters"""

    def __init__(self, dv, wv, syn1neg, word_keys, hyper, doc_keys=None):
        self.dv = self.docvecs = KeyedVectors(doc_keys if doc_keys is not None else [str(i) for i in range(len(dv))], dv)
        self.wv = KeyedVectors(word_keys, wv)
        self.syn1neg = syn1neg
        for k, v in hyper.items(): setattr(self, k, v)


class D2v(T2v):
    def _prep(self, teamsvecs, splits=None, time_indexes=None):
        """d2v.py:17-50.  The documents are kept as CSR over word ids (self.data = (doc_ptr, words, key)); building them from the sparse matrices is one vectorised
        pass, so there is no `{embtype}.docs.pkl` cache (the reference's holds gensim TaggedDocument objects)."""
        self.data = team_documents(teamsvecs, cfg_get(self.cfg, "embtype"), time_indexes)
        assert teamsvecs["member"].shape[0] == len(self.data[0]) - 1
        return self

    def _modelstr(self):
        c = self.cfg
        return f"{self.name}.d{cfg_get(c, 'd')}.e{cfg_get(c, 'e')}.w{cfg_get(c, 'w')}.dm{cfg_get(c, 'dm')}.{cfg_get(c, 'embtype')}"

    def _load(self, path, n_teams):
        import torch
        from . import gensim_reader
        if gensim_reader.is_pickle(path):
            # a file the REFERENCE trained (src/mdl/emb/d2v.py:58-63 loads it with gensim's Doc2Vec.load and skips training): read without gensim, same vectors
            g = gensim_reader.read_doc2vec(path)
            assert g["dv"].shape[0] == n_teams, f"Incorrect number of embeddings per team! {g['dv'].shape[0]} != {n_teams}"
            self.model = Doc2VecTables(g["dv"], g["wv"], g["syn1neg"], g["wv_keys"], g["hyper"], doc_keys=g["dv_keys"])
            return self
        ck = torch.load(path, map_location="cpu", weights_only=False)
        if not isinstance(ck, dict) or ck.get("format") != FORMAT:
            raise RuntimeError(f"{path} is not a gensim Doc2Vec file and was not written by opentf_amd.mdl.emb.d2v: remove it to retrain the vectors on the device")
        dv = ck["dv"].numpy()
        assert dv.shape[0] == n_teams, f"Incorrect number of embeddings per team! {dv.shape[0]} != {n_teams}"
        self.model = Doc2VecTables(dv, ck["wv"].numpy(), ck["syn1neg"].numpy(), ck["word_keys"], ck["hyper"])
        return self

    def _save(self, net, word_keys, hyper, path):
        import torch
        if dist_rank() == 0:
            tmp = f"{path}.tmp.{os.getpid()}"
            torch.save({"format": FORMAT, "dv": torch.from_numpy(net.vectors(0)), "wv": torch.from_numpy(net.vectors(1)), "syn1neg": torch.from_numpy(net.vectors(2)),
                        "word_keys": word_keys, "hyper": dict(hyper), "cfg": self.cfg}, tmp)
            os.replace(tmp, path)

    def learn(self, teamsvecs, splits=None, time_indexes=None):
        from ... import libntf
        c = self.cfg
        modelstr = self._modelstr()
        modelpath = f"{self.output}/{modelstr}"
        modelfile = f"{modelpath}/{modelstr}.pt"
        n_teams = teamsvecs["member"].shape[0]
        _barrier()
        if os.path.exists(modelfile):                                     # d2v.py:58-64
            log.info(f"Loading the model {modelfile} for {(n_teams, cfg_get(c, 'd'))} embeddings ...")
            self._load(modelfile, n_teams)
            self.output = modelpath
            return self
        log.info("File not found! Training the embedding model from scratch ...")
        self._prep(teamsvecs, splits, time_indexes)
        self.output = modelpath
        os.makedirs(self.output, exist_ok=True)
        doc_ptr, words, key = self.data
        d, e, w, dm, min_alpha = int(cfg_get(c, "d")), int(cfg_get(c, "e")), int(cfg_get(c, "w")), int(cfg_get(c, "dm")), float(cfg_get(c, "lr"))
        if not 1 <= d <= 256:
            raise ValueError(f"d2v on the device trains vector sizes 1..256 (a wave holds ceil(d / 64) values per lane, at most four); data.embedding.d = {d}")
        longest = int(np.diff(np.asarray(doc_ptr)).max()) if len(doc_ptr) > 1 else 0
        if longest > 10000:
            log.info(f"The longest document has {longest} words: as in gensim, at most 10 000 of a document's words that survive the subsampling are trained on")
        spe = cfg_get(c, "spe")
        seed = int(self.seed) if self.seed is not None else 1                # gensim's default seed
        keys, count, sample_int, cum_table, words_v = build_vocab(words)
        word_keys = [key(int(k)) for k in keys]
        wv0, dv0 = initial_vectors(n_teams, len(keys), d, seed)
        # (gensim's train(epochs=n) leaves model.epochs = n: 1 after the reference's per-epoch loop, d2v.py:78 - which is what infer_vector then runs)
        hyper = {"vector_size": d, "window": w, "dm": dm, "dbow_words": 1, "negative": NEGATIVE, "sample": SAMPLE, "ns_exponent": NS_EXPONENT, "min_alpha": min_alpha,
                 "alpha": ALPHA, "epochs": 1 if cfg_get(c, "spe") else e, "seed": seed, "corpus_count": n_teams, "corpus_total_words": int(len(words)), "count": count}
        order = None
        if spe:                                                               # d2v.py:74-75 (on every rank: the ranks' `random` streams stay in step)
            order = list(range(n_teams))
            random.shuffle(order)                                             # random.shuffle(self.data): the same permutation of the documents
            order = np.asarray(order, dtype=np.int64)
        if dist_rank() == 0:                                                  # one trainer; the other ranks of a torchrun job read its file
            net = libntf.Doc2Vec(doc_ptr, words_v, sample_int, cum_table, wv0, dv0, seed=seed, device=parse_devices(self.device)[0])
            if spe:                                                           # d2v.py:76-83
                progress = job_progress(doc_ptr, order)
                alpha = ALPHA
                for epoch in range(e):
                    _, ms = net.train_epoch(dm, w, alpha, min_alpha, epoch, negative=NEGATIVE, order=order, progress=progress, want_ms=True)
                    alpha = max(alpha - (alpha - min_alpha) / (e - 1), min_alpha) if e > 1 else min_alpha
                    hyper["alpha"] = alpha
                    if epoch == 0 or ((epoch + 1) % spe) == 0:
                        log.info(f"Saving model at {modelfile}.e{epoch} at lr {alpha} (epoch on the device: {ms:.1f} ms) ...")
                        self._save(net, word_keys, hyper, modelfile.replace(".pt", f".e{epoch}.pt"))
            else:                                                             # d2v.py:84: one linear ramp over all e passes
                progress = job_progress(doc_ptr)
                for epoch in range(e):
                    net.train_epoch(dm, w, ALPHA - (ALPHA - min_alpha) * epoch / e, ALPHA - (ALPHA - min_alpha) * (epoch + 1) / e, epoch, negative=NEGATIVE, progress=progress)
            log.info(f"Saving model at {modelfile} ...")
            self._save(net, word_keys, hyper, modelfile)
            net.close()
        _barrier()
        return self._load(modelfile, n_teams)

    def infer_vec(self, words):
        """d2v.py:96-98: gensim's infer_vector (a new doc vector trained against the frozen tables) and its nearest teams.  Host-side: one short document."""
        m = self.model
        idx = [m.wv.key_to_index[x] for x in words if x in m.wv.key_to_index]
        import zlib
        rng = np.random.default_rng(zlib.crc32(" ".join(words).encode()))     # (gensim seeds with Python's salted hash of the words: not reproducible across processes)
        v = ((rng.random(m.vector_size, dtype=np.float32) * 2 - 1) / m.vector_size).astype(np.float32)
        epochs = int(getattr(m, "epochs", 10))
        if getattr(m, "count", None) is None:      # a gensim file whose wv.expandos carries no 'count' (gensim_reader.read_doc2vec): no table to draw negatives from
            raise RuntimeError("infer_vec: the loaded Doc2Vec file holds no vocabulary counts (wv.expandos['count']): the negative-sampling table cannot be rebuilt")
        cum = np.round(np.cumsum(np.asarray(m.count, np.float64) ** m.ns_exponent) / np.sum(np.asarray(m.count, np.float64) ** m.ns_exponent) * (2 ** 31 - 1))
        a0, a1 = float(getattr(m, "alpha", ALPHA)), float(m.min_alpha)
        alpha_delta = (a0 - a1) / max(epochs - 1, 1)                          # gensim's infer_vector: alpha falls to min_alpha over the epochs - 1 steps between passes
        for ep in range(epochs):
            alpha = a0 - alpha_delta * ep
            for i, word in enumerate(idx):
                b = int(rng.integers(m.window))
                ctx = [idx[j] for j in range(max(0, i - m.window + b), min(len(idx), i + m.window + 1 - b)) if j != i]
                l1 = (v + m.wv.vectors[ctx].sum(0)) / (len(ctx) + 1) if m.dm else v
                work = np.zeros_like(v)
                for k in range(m.negative + 1):
                    t = word if k == 0 else int(np.searchsorted(cum, rng.integers(int(cum[-1])), side="left"))
                    if k and t == word: continue
                    f = float(l1 @ m.syn1neg[t])
                    if abs(f) >= 6.0: continue
                    work += np.float32((float(k == 0) - 1.0 / (1.0 + np.exp(-f))) * alpha) * m.syn1neg[t]
                v = v + work
        return v, m.docvecs.most_similar([v])

    @staticmethod
    def natsortvecs(d2v_model_wv):
        """d2v.py:100-106: word vectors in natural order of their keys (['m3', 's10', 's2', 's1'] -> ['m3', 's1', 's2', 's10'])"""
        nat = lambda s: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]
        sorted_words = sorted(d2v_model_wv.index_to_key, key=nat)
        return d2v_model_wv.vectors[np.array([d2v_model_wv.key_to_index[w] for w in sorted_words])]

    def get_dense_vecs(self, teamsvecs, vectype="skill"):
        """d2v.py:108-116"""
        assert cfg_get(self.cfg, "embtype") == vectype, f"Incorrect d2v model ({cfg_get(self.cfg, 'embtype')}) for the requested vector type {vectype}"
        dv = self.model.docvecs
        indices = [dv.key_to_index[str(i)] for i in range(len(dv))]
        assert indices == list(range(len(dv))), "Incorrect embedding for a team due to misorderings of embeddings!"
        return dv.vectors
