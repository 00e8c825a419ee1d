"""CPU oracle of the doc2vec table producer (src/mdl/emb/d2v.py:52-91).  TEST INFRASTRUCTURE ONLY: nothing under opentf_amd/ imports it.

The reference trains the team vectors with `gensim.models.Doc2Vec` (gensim==4.3.3, requirements.txt:44: a third-party dependency that is neither vendored nor
installable here) called as
    Doc2Vec(min_count=1, dbow_words=1, dm=cfg.dm, vector_size=cfg.d, window=cfg.w, min_alpha=cfg.lr, workers=.., seed=seed)       (d2v.py:69-71)
    build_vocab(docs); [random.shuffle(docs); per epoch: train(epochs=1); alpha -= (alpha - min_alpha) / (e - 1)] | train(epochs=e)   (d2v.py:73-84)
so every other hyper-parameter is gensim's default: alpha 0.025, negative 5, hs 0, sample 1e-3, ns_exponent 0.75, dm_mean -> cbow_mean 1, dm_concat 0,
shrink_windows True.  This file restates gensim's published algorithm for that call (word2vec.py prepare_vocab / make_cum_table / prepare_weights,
doc2vec_inner.pyx fast_document_dm_neg / fast_document_dbow_neg):

  vocabulary   words in first-seen order, counted, re-ordered by descending count; keep probability of word w
               p = (sqrt(c_w / t) + 1) * t / c_w, t = sample * total_words, clipped at 1, kept as uint32(p * (2^32 - 1)) ("sample_int");
               negative-sampling table cum_table[i] = round(sum_{j <= i} c_j^0.75 / sum_j c_j^0.75 * (2^31 - 1))
  initial      word vectors  (default_rng(seed).random(float32) * 2 - 1) / d,  doc vectors the same with seed + 7919, syn1neg = 0
  PV-DM        per document: drop words by sample_int, per kept position i a window shrunk by b_i = rand % window; l1 = mean of the doc vector and the window's
               word vectors; for the word (label 1) and `negative` table draws (label 0, a draw equal to the word is skipped): f = l1 . syn1neg[t], skipped when
               |f| >= 6, sigmoid from the 1000-bin table over [-6, 6), g = (label - sigmoid) * alpha, work += g * syn1neg[t], syn1neg[t] += g * l1;
               then the doc vector and every window word vector += work (gensim divides work by the count only when it SUMS the inputs: `if not cbow_mean`)
  PV-DBOW      (dm = 0, dbow_words = 1) per kept position i: for every other position j of the shrunk window the same unit with input word vector j and word i,
               the input += work; then the unit with the doc vector as input
  alpha        linear from the call's start to its end over the documents of the call, in steps: fixed per job of <= 10 000 words (job_progress)

What is NOT gensim's: the random streams.  gensim draws from numpy's RandomState and a 48-bit LCG per worker job and trains Hogwild on `workers` threads, so its
output is not reproducible from a seed when workers > 1 (the reference: all cores).  Here every draw is a Philox4x32-10 word keyed by (seed, epoch) and
counted by (document, position, unit, slot) - the HIP trainer's definition (opentf_amd/csrc/ntf_d2v.hip) - so a sequential pass over the documents is a pure
function of its inputs, and the HIP trainer run with one wave (`serial`) must reproduce `train_epoch` to rounding.

Pinning (tests/golden/g15_d2v_toy.npz, extracted by tests/golden/make_d2v_golden.py from the gensim pickles the reference's authors committed under
output/{dblp,imdb,uspt}/toy.*/splits.f3.r0.85/d2v.d128.e100.w5.dm1.skill/): vocabulary order, counts and sample_int VALUE for value; the initial word vectors
bit for bit (the committed .e0.pt still holds them) and the initial doc vectors bit for bit on the documents epoch 0 left untouched; the alpha the
reference's loop ends at (0.009695595942157981); the trained vectors in distribution (norms).  The training arithmetic itself has no value-level pin:
parity status "pinned at the vocabulary / initialisation / schedule level, in distribution for the trained table".
"""
from __future__ import annotations

import numpy as np

MAX_EXP = 6.0
MAX_DOCUMENT_LEN = 10000     # doc2vec_inner.pyx: the words of a document that survive the subsampling are collected up to this many (`if i == MAX_DOCUMENT_LEN: break`)
EXP_TABLE_SIZE = 1000
NEGATIVE = 5
SAMPLE = 1e-3
NS_EXPONENT = 0.75
ALPHA = 0.025


# ------------------------------------------------------------------------------------------------ documents (d2v.py:17-50)
def team_docs(skill_csr, member_csr=None, embtype="skill", years=None):
    """word lists of the teams: 's{idx}' in column order, then 'm{idx}' (skillmember) or 'dt{year}' (skilltime).  Returned as (doc_ptr, words) with words numbered
    skills 0..S-1, members S..S+M-1 / years S + (year - min year).  skill_csr = (indptr, indices, S)"""
    s_ip, s_ix, S = skill_csr
    n = len(s_ip) - 1
    docs = []
    for i in range(n):
        w = [] if embtype == "member" else list(map(int, s_ix[s_ip[i]:s_ip[i + 1]]))
        if embtype in ("member", "skillmember"):
            m_ip, m_ix, _ = member_csr
            off = 0 if embtype == "member" else S
            w += [off + int(c) for c in m_ix[m_ip[i]:m_ip[i + 1]]]
        if embtype == "skilltime":
            w.append(S + int(years[i]))
        docs.append(w)
    ptr = np.concatenate([[0], np.cumsum([len(d) for d in docs])]).astype(np.int64)
    return ptr, np.asarray([x for d in docs for x in d], dtype=np.int64)


# ------------------------------------------------------------------------------------------------ vocabulary (gensim word2vec.py scan_vocab / prepare_vocab / make_cum_table)
def prepare_vocab(doc_ptr, words, sample=SAMPLE, ns_exponent=NS_EXPONENT):
    """-> dict(keys = word of vocabulary index v (count-descending, gensim's tie order), count, sample_int uint32, cum_table uint32, index_of: word -> v)"""
    words = np.asarray(words, dtype=np.int64)
    first = {}
    for w in words.tolist():
        if w not in first: first[w] = len(first)
    keys0 = np.fromiter(first.keys(), dtype=np.int64, count=len(first))
    cnt0 = np.zeros(len(first), dtype=np.int64)
    np.add.at(cnt0, np.fromiter((first[w] for w in words.tolist()), dtype=np.int64, count=len(words)), 1)
    # KeyedVectors.sort_by_descending_frequency is np.argsort(count)[::-1]: an unstable sort whose order among equal counts depends on the numpy build (the committed
    # toy models show three different patterns).  Words of equal count are exchangeable (same keep probability, same table weight): ties in first-seen order here.
    order = np.argsort(-cnt0, kind="stable")
    keys, count = keys0[order], cnt0[order]
    total = int(count.sum())
    if not sample: thr = total
    elif sample < 1.0: thr = sample * total
    else: thr = int(sample * (3 + np.sqrt(5)) / 2)
    p = (np.sqrt(count / thr) + 1) * (thr / count)
    p = np.minimum(p, 1.0)
    sample_int = (p * (2 ** 32 - 1)).astype(np.uint32)   # np.uint32(word_probability * (2**32 - 1))
    pw = count.astype(np.float64) ** ns_exponent
    cum = np.round(np.cumsum(pw) / pw.sum() * (2 ** 31 - 1)).astype(np.uint32)
    return {"keys": keys, "count": count, "sample_int": sample_int, "cum_table": cum, "index_of": {int(k): v for v, k in enumerate(keys)}, "total_words": total}


def init_vectors(n_docs, n_vocab, d, seed):
    """Word2Vec.init_weights / Doc2Vec.init_weights: (rng.random(float32) * 2 - 1) / d, words from default_rng(seed), doc tags from default_rng(seed + 7919)"""
    def prep(shape, s):
        v = np.random.default_rng(seed=s).random(shape, dtype=np.float32)
        v *= 2.0; v -= 1.0; v /= shape[1]
        return v
    return prep((n_vocab, d), seed), prep((n_docs, d), seed + 7919), np.zeros((n_vocab, d), np.float32)


def alpha_schedule(e, min_alpha, spe, alpha=ALPHA):
    """(start, end) of alpha for each of the e passes.  spe set (d2v.py:76-83): train(epochs=1) decays from model.alpha to min_alpha inside every pass, and
    model.alpha steps down between passes; else one train(epochs=e): one linear ramp cut into e pieces.  Also returns the model.alpha the loop ends at."""
    out = []
    if spe:
        a = alpha
        for _ in range(e):
            out.append((a, min_alpha))
            a = max(a - (a - min_alpha) / (e - 1), min_alpha)
        return out, a
    for k in range(e):
        out.append((alpha - (alpha - min_alpha) * k / e, alpha - (alpha - min_alpha) * (k + 1) / e))
    return out, alpha


# ------------------------------------------------------------------------------------------------ random streams (the HIP trainer's definition)
def _mulhilo(a, b):
    p = np.uint64(a) * np.uint64(b)
    return np.uint32(p >> np.uint64(32)), np.uint32(p & np.uint64(0xFFFFFFFF))


def philox4x32(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11) of one counter (4 x uint32) under one key (2 x uint32)"""
    c = [np.uint32(x) for x in ctr]; k = [np.uint32(x) for x in key]
    with np.errstate(over="ignore"):
        for _ in range(10):
            hi0, lo0 = _mulhilo(0xD2511F53, c[0]); hi1, lo1 = _mulhilo(0xCD9E8D57, c[2])
            c = [hi1 ^ c[1] ^ k[0], lo1, hi0 ^ c[3] ^ k[1], lo0]
            k = [np.uint32((int(k[0]) + 0x9E3779B9) & 0xFFFFFFFF), np.uint32((int(k[1]) + 0xBB67AE85) & 0xFFFFFFFF)]
    return [int(x) for x in c]


def epoch_key(seed, epoch):
    """splitmix64 of (seed, epoch) -> the two key words (ntf_d2v.hip d2v_key)"""
    M = (1 << 64) - 1
    x = (seed ^ ((epoch * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & M)) & M
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & M; x ^= x >> 27; x = (x * 0x94D049BB133111EB) & M; x ^= x >> 31
    return x & 0xFFFFFFFF, x >> 32


SLOT_KEEP, SLOT_WINDOW, SLOT_NEG0, SLOT_NEG1 = 0, 1, 2, 3


def draw(key, doc, pos, unit, slot):
    return philox4x32((doc & 0xFFFFFFFF, (doc >> 32) & 0xFFFFFFFF, (pos << 8 | unit) & 0xFFFFFFFF, slot), key)


def sigmoid_table(f):
    """EXP_TABLE lookup of doc2vec_inner.pyx: bin (int)((f + 6) * (1000 / 6 / 2)); entry i = sigmoid((i / 1000 * 2 - 1) * 6), built in float32"""
    i = int((np.float32(f) + np.float32(MAX_EXP)) * np.float32(EXP_TABLE_SIZE / MAX_EXP / 2))
    e = np.float32(np.exp(np.float32((np.float32(i) / np.float32(EXP_TABLE_SIZE) * np.float32(2) - np.float32(1)) * np.float32(MAX_EXP))))
    return np.float32(e / (e + np.float32(1)))


# ------------------------------------------------------------------------------------------------ one pass over the documents, sequential
def _unit(l1, word, syn1neg, cum_table, alpha, key, doc, pos, unit, negative, stats):
    """negative-sampling unit: returns `work`, updates syn1neg rows in place"""
    work = np.zeros_like(l1)
    r = draw(key, doc, pos, unit, SLOT_NEG0) + draw(key, doc, pos, unit, SLOT_NEG1)
    for k in range(negative + 1):
        if k == 0: target, label = word, np.float32(1)
        else:
            target = int(np.searchsorted(cum_table, np.uint32(r[k - 1] % int(cum_table[-1])), side="left"))
            if target == word: continue
            label = np.float32(0)
        f = np.float32(np.dot(l1.astype(np.float64), syn1neg[target].astype(np.float64)))
        if f <= -MAX_EXP or f >= MAX_EXP: continue
        s = sigmoid_table(f)
        if stats is not None:
            stats[0] += -np.log(max(float(s if label else 1 - s), 1e-30)); stats[1] += 1
        g = np.float32((label - s) * np.float32(alpha))
        work += g * syn1neg[target]
        syn1neg[target] += g * l1
    return work


def job_progress(doc_ptr, order=None, batch_words=10000):
    """gensim's _job_producer (word2vec.py): documents are packed greedily into jobs of at most `batch_words` raw words (at least one document), and a job's alpha
    is fixed when it is cut: start - (start - end) * (documents pushed before it / all documents).  -> that fraction for every rank of the pass.
    (A corpus below batch_words - the reference's toy runs - is ONE job: the whole pass runs at its start alpha.)"""
    n = len(doc_ptr) - 1
    order = np.arange(n) if order is None else np.asarray(order)
    lens = np.diff(np.asarray(doc_ptr))[order]
    cum = np.concatenate([[0], np.cumsum(lens)])
    prog = np.empty(n, dtype=np.float64)
    s = 0
    while s < n:
        e = max(int(np.searchsorted(cum, cum[s] + batch_words, side="right")) - 1, s + 1)
        prog[s:e] = s / n
        s = e
    return prog


def train_epoch(doc_ptr, words_v, vocab, wv, dv, syn1neg, dm, window, alpha_start, alpha_end, seed, epoch, negative=NEGATIVE, order=None, return_loss=False,
                progress=None):
    """one pass (gensim train(epochs=1)) over the documents in `order` (default 0..N-1), in place.  words_v = vocabulary indices of the corpus words.
    progress: job_progress(..) (None: rank / N, the limit of small jobs)"""
    key = epoch_key(seed, epoch)
    n = len(doc_ptr) - 1
    order = np.arange(n) if order is None else np.asarray(order)
    sample_int, cum = vocab["sample_int"], vocab["cum_table"]
    stats = [0.0, 0] if return_loss else None
    for rank, doc in enumerate(order.tolist()):
        alpha = np.float32(alpha_start - (alpha_start - alpha_end) * (rank / n if progress is None else float(progress[rank])))
        w = words_v[doc_ptr[doc]:doc_ptr[doc + 1]]
        kept = [int(x) for p, x in enumerate(w.tolist()) if int(sample_int[x]) >= draw(key, doc, p, 0, SLOT_KEEP)[0]][:MAX_DOCUMENT_LEN]
        K = len(kept)
        for i in range(K):
            b = draw(key, doc, i, 0, SLOT_WINDOW)[0] % window
            lo, hi = max(0, i - window + b), min(K, i + window + 1 - b)
            ctx = [m for m in range(lo, hi) if m != i]
            if dm:
                l1 = dv[doc].copy()
                for m in ctx: l1 += wv[kept[m]]
                inv = np.float32(1.0) / np.float32(len(ctx) + 1)
                l1 *= inv
                work = _unit(l1, kept[i], syn1neg, cum, alpha, key, doc, i, 0, negative, stats)
                dv[doc] += work
                for m in ctx: wv[kept[m]] += work
            else:
                for u, m in enumerate(ctx):
                    wv[kept[m]] += _unit(wv[kept[m]].copy(), kept[i], syn1neg, cum, alpha, key, doc, i, 1 + u, negative, stats)
                dv[doc] += _unit(dv[doc].copy(), kept[i], syn1neg, cum, alpha, key, doc, i, 0, negative, stats)
    if return_loss: return stats[0] / max(stats[1], 1)
