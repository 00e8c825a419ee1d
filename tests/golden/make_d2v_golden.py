"""Build-container only (needs /root/reference; listed in .gpurunignore): extracts the doc2vec golden data from the gensim pickles the reference's authors
committed - output/{dblp,imdb,uspt}/toy.*/skill.docs.pkl (the documents d2v.py:17-50 built) and .../splits.f3.r0.85/d2v.d128.e100.w5.dm1.skill/*.pt (the
Doc2Vec objects after epoch 0 and after the last epoch, d2v.py:76-87) - into tests/golden/g15_d2v_toy.npz.  gensim is not installed: the pickles are read with
an Unpickler that stands plain attribute holders in for gensim's classes; only numpy arrays, lists and scalars are taken out."""
import pickle
import sys

import numpy as np


class _Holder:
    def __new__(cls, *a, **k):
        o = object.__new__(cls); o._args = a
        return o
    def __init__(self, *a, **k): pass
    def __setstate__(self, st): self.__dict__.update(st if isinstance(st, dict) else {"_state": st})


class _U(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("numpy") or module in ("builtins", "collections", "copyreg", "_codecs"):
            return super().find_class(module, name)
        return type(name, (_Holder,), {"__module__": module})


def load(path):
    with open(path, "rb") as f: return _U(f).load()


def main(out="tests/golden/g15_d2v_toy.npz"):
    z = {}
    for ds, f in (("dblp", "toy.dblp.v12.json"), ("imdb", "toy.title.basics.tsv"), ("uspt", "toy.patent.tsv")):
        base = f"/root/reference/output/{ds}/{f}"
        docs = load(base + "/skill.docs.pkl")
        words = [[int(w[1:]) for w in d._args[0]] for d in docs]
        assert all(d._args[1] == [str(i)] for i, d in enumerate(docs))
        z[f"{ds}_doc_ptr"] = np.concatenate([[0], np.cumsum([len(w) for w in words])]).astype(np.int64)
        z[f"{ds}_words"] = np.asarray([x for w in words for x in w], dtype=np.int64)
        stem = base + "/splits.f3.r0.85/d2v.d128.e100.w5.dm1.skill/d2v.d128.e100.w5.dm1.skill"
        for tag, suffix in (("e0", ".e0.pt"), ("final", ".pt")):
            m = load(stem + suffix)
            assert [str(i) for i in range(len(docs))] == list(m.dv.index_to_key)
            z[f"{ds}_{tag}_dv"] = np.asarray(m.dv.vectors, np.float32)
            z[f"{ds}_{tag}_wv"] = np.asarray(m.wv.vectors, np.float32)
            z[f"{ds}_{tag}_syn1neg"] = np.asarray(m.syn1neg, np.float32)
            z[f"{ds}_{tag}_alpha"] = np.float64(m.alpha)
        z[f"{ds}_keys"] = np.asarray([int(k[1:]) for k in m.wv.index_to_key], dtype=np.int64)
        z[f"{ds}_count"] = np.asarray(m.wv.expandos["count"], dtype=np.int64)
        z[f"{ds}_sample_int"] = np.asarray(m.wv.expandos["sample_int"], dtype=np.uint32)
        z[f"{ds}_hyper"] = np.asarray([m.vector_size, m.window, m.negative, m.dm_concat, m.dbow_words, m.sg, m.hs, m.cbow_mean, m.seed, m.min_count, m.train_count,
                                       m.corpus_total_words], dtype=np.int64)       # sg = 0 <=> dm = 1
        z[f"{ds}_hyper_f"] = np.asarray([m.min_alpha, m.sample, m.ns_exponent], dtype=np.float64)
    np.savez_compressed(out, **z)
    print({k: getattr(v, "shape", v) for k, v in z.items()})


if __name__ == "__main__":
    main(*sys.argv[1:])
