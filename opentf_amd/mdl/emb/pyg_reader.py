"""Reads what the reference's random-walk branch leaves on disk - WITHOUT torch_geometric or omegaconf - so that `bnn_emb` / `fnn_emb` runs can consume a node2vec
table the reference trained (src/mdl/emb/gnn.py:402-405 / :323-326 load an existing `f{k}.pt` instead of training).

Two files:
  * `<...>/{structure}.{dup_edge}.graph.pkl` (gnn.py:21-23,58-59): `pickle.dump(HeteroData)`.  A trained table's rows follow `data.to_homogeneous()`, i.e. the node
    stores in the order `HeteroData._node_store_dict` holds them - the iteration order of a Python `set` of type names when the graph was built (gnn.py:29-47), which
    depends on that process's hash seed and is recorded nowhere else.  `node_blocks` returns [(node type, node count), ...] in that order.
  * `f{k}.pt` / `f{k}.e{e}.pt` (gnn.py:445,453): `torch.save({'model_state_dict': {'embedding.weight': ...}, 'cfg': <omegaconf>, 'e', 't_loss', 'v_loss'})`.
    `reference_table` returns the weight and the scalars.
Both are read with RESTRICTED unpicklers: no code of a pickled class runs.  In the graph file every class resolves to an inert attribute holder and the two torch
tensor-rebuilding functions to stand-ins that keep only the SHAPE (the node features are placeholders; only the counts matter).  In the checkpoint torch's own tensor
rebuild functions and storages resolve to the real thing (torch.load needs them), every other class (omegaconf's) to the inert holder."""
from __future__ import annotations

import pickle

import numpy as np


class _Holder:
    """stands in for a pickled object: takes its state, runs none of its code"""
    def __new__(cls, *a, **k):
        o = object.__new__(cls); o._args = a
        return o
    def __init__(self, *a, **k): pass
    def __setstate__(self, st): self.__dict__.update(st if isinstance(st, dict) else {"_state": st})
    def __call__(self, *a, **k): return None
    # a pickled dict / list / set SUBCLASS (e.g. a config object) is filled through these: kept as plain data
    def __setitem__(self, k, v): self.__dict__.setdefault("_items", {})[k] = v
    def append(self, v): self.__dict__.setdefault("_list", []).append(v)
    def extend(self, vs): self.__dict__.setdefault("_list", []).extend(vs)
    def add(self, v): self.__dict__.setdefault("_list", []).append(v)
    def update(self, *a, **k): self.__dict__.setdefault("_items", {}).update(*a, **k)


def _holder(module, name):
    return type(name, (_Holder,), {"__module__": module})


class _Shape:
    """what is kept of a tensor inside the graph file"""
    def __init__(self, shape): self.shape = tuple(int(s) for s in shape)


def _rebuild_shape_only(storage, storage_offset, size, stride, *rest):
    return _Shape(size)


_SAFE_BUILTINS = {("builtins", n) for n in ("object", "int", "float", "str", "bool", "list", "tuple", "dict", "set", "frozenset", "slice", "range", "complex", "bytearray")} | \
                 {("collections", "OrderedDict"), ("collections", "defaultdict"), ("copyreg", "_reconstructor")}


class _GraphUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE_BUILTINS: return super().find_class(module, name)
        if (module, name) in (("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor")): return _rebuild_shape_only
        if (module, name) == ("torch.storage", "_load_from_bytes"): return lambda b: None          # the bytes are a nested torch pickle of the storage: not needed, not opened
        return _holder(module, name)


def node_blocks(graph_pkl):
    """[(node type, number of nodes), ...] in the order the pickled HeteroData holds its node stores = the row blocks of a table trained on its homogeneous form"""
    with open(graph_pkl, "rb") as f:
        data = _GraphUnpickler(f).load()
    stores = getattr(data, "_node_store_dict", None)
    if not isinstance(stores, dict) or not stores:
        raise RuntimeError(f"{graph_pkl}: no `_node_store_dict` inside (not a pickled torch_geometric HeteroData?)")
    out = []
    for key, st in stores.items():
        mp = getattr(st, "_mapping", None)
        x = mp.get("x") if isinstance(mp, dict) else None
        n = getattr(st, "num_nodes", None) if x is None else (x.shape[0] if getattr(x, "shape", None) else None)
        if isinstance(mp, dict) and x is None and "num_nodes" in mp: n = mp["num_nodes"]
        if n is None:
            raise RuntimeError(f"{graph_pkl}: node store {key!r} carries neither `x` nor `num_nodes`")
        out.append((str(key), int(n)))
    return out


_TORCH_OK = ("torch._utils", "torch", "torch.storage", "torch._tensor", "torch.serialization", "torch.nn.parameter")


def _load_from_bytes_weights_only(b):
    """torch.storage._load_from_bytes is a nested, unrestricted torch.load of a storage's bytes: here the nested load admits tensors and storages only"""
    import io
    import torch
    return torch.load(io.BytesIO(b), map_location="cpu", weights_only=True)


class _CkptUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _SAFE_BUILTINS: return super().find_class(module, name)
        if (module, name) == ("torch.storage", "_load_from_bytes"): return _load_from_bytes_weights_only
        if module in _TORCH_OK and (name.startswith("_rebuild") or name.endswith("Storage") or name in ("Tensor", "Size", "device", "dtype", "Parameter")):
            return super().find_class(module, name)
        return _holder(module, name)


class _PickleModule:
    """what torch.load takes as `pickle_module`"""
    __name__ = "opentf_amd_restricted_pickle"
    Unpickler = _CkptUnpickler
    load = staticmethod(lambda f, **k: _CkptUnpickler(f, **k).load())


def reference_table(path):
    """{'weight': float32 [n, d] = model_state_dict['embedding.weight'], 'e', 't_loss', 'v_loss'} of a checkpoint gnn.py:445,453 wrote, + 'node_order' (str or None) and
    'node_offsets' (tuple of ints or None): the markers opentf_amd.mdl.emb.gnn._save adds - plain builtins, so they survive the restricted load whatever the pickled cfg is"""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)
    sd = ck.get("model_state_dict") if isinstance(ck, dict) else None
    if not isinstance(sd, dict) or "embedding.weight" not in sd:
        raise RuntimeError(f"{path}: no model_state_dict['embedding.weight'] inside (not a node2vec checkpoint of src/mdl/emb/gnn.py)")
    w = sd["embedding.weight"]
    order, offs = ck.get("node_order"), ck.get("node_offsets")
    order = order if isinstance(order, str) else None
    offs = tuple(int(o) for o in offs) if isinstance(offs, (tuple, list)) and all(isinstance(o, int) for o in offs) else None
    return {"weight": np.ascontiguousarray(w.detach().cpu().numpy(), dtype=np.float32), "e": ck.get("e"), "t_loss": ck.get("t_loss"), "v_loss": ck.get("v_loss"),
            "node_order": order, "node_offsets": offs}


def blocks_to_order(weight, blocks, want):
    """rows of `weight` (blocks in the file's order) re-stacked in the order `want` (a list of node types); raises when a type is missing or the counts do not add up"""
    if sum(n for _, n in blocks) != weight.shape[0]:
        raise RuntimeError(f"the graph's node counts {blocks} do not add up to the table's {weight.shape[0]} rows")
    start, at = 0, {}
    for t, n in blocks:
        at[t] = (start, start + n); start += n
    missing = [t for t in want if t not in at]
    if missing:
        raise RuntimeError(f"node type(s) {missing} not in the graph's node stores {[t for t, _ in blocks]}")
    return np.concatenate([weight[at[t][0]: at[t][1]] for t in want], axis=0), {t: at[t][1] - at[t][0] for t in want}
