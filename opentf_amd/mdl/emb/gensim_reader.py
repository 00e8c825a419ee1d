"""Reads a `gensim.models.Doc2Vec` file - what the reference's D2v.learn saves with `model.save(path)` and loads back with `Doc2Vec.load(path)`
(src/mdl/emb/d2v.py:58-63,81-87) - WITHOUT gensim: `bnn_emb` / `fnn_emb` runs can then consume team vectors the reference trained.

gensim's `save` is `pickle.dump(self, protocol=4)` of the model object (utils.SaveLoad; arrays above 10 MB would go to side files `<path>.*.npy`, which raises here
rather than being silently ignored).  The file is read with a RESTRICTED unpickler: numpy's array reconstruction and a few builtins resolve to the real thing, every
`gensim.*` class to an inert attribute holder (no code of the pickled classes runs, none is needed: only arrays, lists and scalars are taken out), anything else is
refused.  What comes out is the `Doc2VecTables` of opentf_amd/mdl/emb/d2v.py - the attributes the reference touches (`dv` / `docvecs`, `wv`, `syn1neg`, the
hyper-parameters `infer_vec` needs)."""
from __future__ import annotations

import pickle

import numpy as np

_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("builtins", "object"), ("builtins", "hash"), ("builtins", "int"), ("builtins", "float"), ("builtins", "str"), ("builtins", "bool"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "dict"),
    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"), ("builtins", "range"), ("builtins", "complex"), ("builtins", "bytearray"),
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("copyreg", "_reconstructor"), ("_codecs", "encode"),
}


class _Holder:
    """stands in for a pickled gensim object: takes its state, runs none of its code"""
    def __new__(cls, *a, **k):
        o = object.__new__(cls); o._args = a
        return o
    def __init__(self, *a, **k): pass
    def __setstate__(self, st): self.__dict__.update(st if isinstance(st, dict) else {"_state": st})
    def __call__(self, *a, **k): return None       # (a pickled default factory / callback attribute)


class _Restricted(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module == "gensim" or module.startswith("gensim.") or module.startswith("numpy.random"):     # (the model's RandomState: not needed, not rebuilt)
            return type(name, (_Holder,), {"__module__": module})
        raise pickle.UnpicklingError(f"refusing to resolve {module}.{name} while reading a gensim Doc2Vec file")


def is_pickle(path) -> bool:
    """a bare pickle stream (protocol >= 2), as opposed to torch.save's zip container"""
    with open(path, "rb") as f:
        return f.read(2)[:1] == b"\x80"


def read_doc2vec(path):
    """dict of what a gensim 4.x Doc2Vec file holds: dv / wv / syn1neg (float32 arrays), their key lists and the hyper-parameters"""
    with open(path, "rb") as f:
        m = _Restricted(f).load()
    for side in ("dv", "wv"):
        if not hasattr(m, side) or getattr(getattr(m, side), "vectors", None) is None:
            raise RuntimeError(f"{path}: no `{side}.vectors` inside (a gensim < 4 file, or arrays stored in side files {path}.{side}.vectors.npy: not supported)")
    if getattr(m, "syn1neg", None) is None:
        raise RuntimeError(f"{path}: no `syn1neg` inside (hierarchical softmax, or a side file {path}.syn1neg.npy: not supported)")
    count = m.wv.expandos.get("count") if isinstance(getattr(m.wv, "expandos", None), dict) else None
    hyper = {"vector_size": int(m.vector_size), "window": int(m.window), "dm": 0 if int(getattr(m, "sg", 0)) else 1, "dbow_words": int(getattr(m, "dbow_words", 0)),
             "negative": int(m.negative), "sample": float(m.sample), "ns_exponent": float(m.ns_exponent), "min_alpha": float(m.min_alpha), "alpha": float(m.alpha),
             "epochs": int(getattr(m, "epochs", 10)), "seed": int(m.seed), "corpus_count": int(getattr(m, "corpus_count", len(m.dv.index_to_key))),
             "corpus_total_words": int(getattr(m, "corpus_total_words", 0)), "count": None if count is None else np.asarray(count, dtype=np.int64)}
    return {"dv": np.ascontiguousarray(m.dv.vectors, dtype=np.float32), "dv_keys": [str(k) for k in m.dv.index_to_key],
            "wv": np.ascontiguousarray(m.wv.vectors, dtype=np.float32), "wv_keys": [str(k) for k in m.wv.index_to_key],
            "syn1neg": np.ascontiguousarray(m.syn1neg, dtype=np.float32), "hyper": hyper}
