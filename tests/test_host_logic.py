"""CPU tests of the product's host side: the C-ABI library loads and exports what include/opentf_amd.h declares, the
plugin mirror's control flow / naming / file helpers match the reference goldens, and nothing in the product
reaches for the oracle or a CPU fallback."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden
from oracle import ntf_oracle as O


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "opentf_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ntf_\w+)\s*\(", src)) - {"ntf_engine", "ntf_config", "ntf_inject"})


def test_library_exports_every_declared_symbol():
    import ctypes
    from opentf_amd import libntf
    path = os.path.join(ROOT, "opentf_amd", "libopentf_amd.so")
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(path)
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/opentf_amd.h but not exported"
    assert sorted(libntf.SYMBOLS) == declared  # the ctypes binding covers the whole header, nothing more
    assert libntf.lib().ntf_abi_version() == libntf.NTF_ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    from opentf_amd import libntf
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(libntf.NtfError, match="no HIP device"):
        libntf.Engine([8, 8, 16])
    from opentf_amd.mdl.fnn import parse_devices
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        parse_devices("cpu")
    assert parse_devices("cuda") == [0] and parse_devices("cuda:3") == [3] and parse_devices("cuda:0,1,2") == [0, 1, 2]


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "opentf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|ntf_oracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_schedulers_match_reference_golden():
    from opentf_amd.mdl.earlystopping import EarlyStopping, PlateauLR
    ref = json.load(open(os.path.join(GOLDEN, "g6_sched.json")))
    for name, r in ref.items():
        es, sch = EarlyStopping(patience=3, delta=0.001, trace_func=lambda *_: None), PlateauLR(0.001)
        for v, (lr, counter, stop) in zip(r["seq"], r["rows"]):
            assert sch.step(v) == pytest.approx(lr, rel=1e-12), name
            es(v, None)
            assert (es.counter, es.early_stop) == (counter, stop), name


def test_output_dir_name_matches_reference():
    from opentf_amd.mdl.ntf import cfg2str
    d = json.load(open(os.path.join(GOLDEN, "g5_dirname.json")))
    assert "/fnn." + cfg2str(d["cfg"]) == d["name"]
    lay = json.load(open(os.path.join(GOLDEN, "g8_layout.json")))
    committed = next(k for k in lay if k.startswith("bnn."))  # a directory name the reference itself produced
    cfg = {"b": 1000, "e": 100, "ns": 5, "lr": 0.001, "es": 5, "h": [128], "spe": 10, "l": "bce", "tpw": 10, "tnw": 1, "nsd": "unigram_b", "nmc": 10}
    assert "bnn." + cfg2str(cfg) == committed


def test_config_objects_that_are_not_dicts_name_the_same_directory_and_pickle(tmp_path):
    """VERDICT r5 missing #4: hydra hands the plugin omegaconf `DictConfig` / `ListConfig` objects (absent here): Mappings and Sequences that are NOT dict / list subclasses,
    reached by `cfg.x` and `cfg['x']`.  Duck-typed stand-ins with those properties (class names included: cfg2str recognises a ListConfig by name when omegaconf itself is not
    importable) must give the directory name the reference committed, serve `cfg_get`, and survive the checkpoint's `torch.save` / restricted re-load."""
    import collections.abc
    import torch
    from opentf_amd.mdl.ntf import cfg2str, cfg_get, cfg_items
    from opentf_amd.mdl.emb import pyg_reader

    class ListConfig(collections.abc.Sequence):
        def __init__(self, v): self._v = list(v)
        def __getitem__(self, i): return self._v[i]
        def __len__(self): return len(self._v)

    class DictConfig(collections.abc.Mapping):
        def __init__(self, d): self.__dict__["_d"] = {k: (ListConfig(v) if isinstance(v, list) else v) for k, v in d.items()}
        def __getitem__(self, k): return self._d[k]
        def __iter__(self): return iter(self._d)
        def __len__(self): return len(self._d)
        def __getattr__(self, k):
            try: return self.__dict__["_d"][k]
            except KeyError: raise AttributeError(k)

    lay = json.load(open(os.path.join(GOLDEN, "g8_layout.json")))
    committed = next(k for k in lay if k.startswith("bnn."))
    cfg = DictConfig({"b": 1000, "e": 100, "ns": 5, "lr": 0.001, "es": 5, "h": [128], "spe": 10, "l": "bce", "tpw": 10, "tnw": 1, "nsd": "unigram_b", "nmc": 10})
    assert not isinstance(cfg, dict) and not isinstance(cfg.h, list)
    assert "bnn." + cfg2str(cfg) == committed
    assert cfg_get(cfg, "b") == 1000 and list(cfg_get(cfg, "h")) == [128] and cfg_get(cfg, "absent", 7) == 7 and dict(cfg_items(cfg))["nsd"] == "unigram_b"
    # the checkpoint carries cfg as the reference's does (src/mdl/fnn.py:158-161); classes local to this test cannot be pickled by reference, a module-level config object can:
    # what is checked here is that a non-dict config inside a checkpoint does not stand in the way of reading the weights back without its class
    import types, sys
    mod = types.ModuleType("fake_omegaconf_for_test"); mod.DictConfig = DictConfig; mod.ListConfig = ListConfig
    DictConfig.__module__ = ListConfig.__module__ = "fake_omegaconf_for_test"; DictConfig.__qualname__ = "DictConfig"; ListConfig.__qualname__ = "ListConfig"
    sys.modules["fake_omegaconf_for_test"] = mod
    try:
        torch.save({"model_state_dict": {"embedding.weight": torch.zeros(3, 2)}, "cfg": cfg, "f": 0, "e": 1, "t_loss": 0.5, "v_loss": 0.6}, tmp_path / "f0.pt")
    finally:
        del sys.modules["fake_omegaconf_for_test"]
    t = pyg_reader.reference_table(str(tmp_path / "f0.pt"))
    assert t["weight"].shape == (3, 2) and t["e"] == 1 and t["v_loss"] == 0.6


def _plugin(cls, cfg, tmp_path, seed=0):
    class Cfg(dict):
        def __getattr__(self, k):
            if k.startswith("__"): raise AttributeError(k)
            return self.get(k)
    return cls(str(tmp_path), "cuda:0", seed, Cfg(cfg))


BASE = dict(b=8, e=6, ns=3, lr=0.001, es=5, h=[32], spe=0, l="bce", tpw=10, tnw=1, nsd="uniform")


def test_init_draws_match_the_reference_order(tmp_path):
    from opentf_amd.mdl.bnn import Bnn
    from opentf_amd.mdl.fnn import Fnn
    g = golden("g1_forward_imdb")  # parameters the reference's Fnn.init produced under seed 0
    m = _plugin(Fnn, BASE, tmp_path, seed=0)
    sd = m.init(18, 112)
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g[f"p.{k}"]), k
    torch.manual_seed(5); ref = O.bnn_init(18, [32], 112)
    b = _plugin(Bnn, {**BASE, "nmc": 2}, tmp_path, seed=5)
    assert b.is_bayesian
    got = b.init(18, 112)
    assert list(got) == list(ref) and all(torch.equal(got[k], ref[k]) for k in ref)
    assert os.path.isdir(m.output) and m.output.endswith("/fnn." + ".".join(f"{k}{v}" for k, v in BASE.items()))


def test_loader_order_consumes_rng_like_dataloader():
    from opentf_amd.mdl import fnn as F
    torch.manual_seed(3); a = F.index_order(23, 5, True); a2 = F.index_order(7, 5, False)
    torch.manual_seed(3); b = np.concatenate(O.index_batches(23, 5, True)); b2 = np.concatenate(O.index_batches(7, 5, False))
    assert np.array_equal(a, b) and np.array_equal(a2, b2) and sorted(a) == list(range(23))
    # against the loader the reference builds (src/mdl/fnn.py:95-96: DataLoader over the dataset with its batch size), order AND the global generator afterwards;
    # both the direct draws and the through-the-DataLoader path index_order falls back to when its one-time check disagrees
    assert F._direct_matches_loader()
    for n, bs, shuffle in [(1, 4, True), (999, 1000, True), (1000, 1000, False), (4097, 128, True), (4097, 128, False)]:
        torch.manual_seed(11)
        ref = torch.cat([t for t in torch.utils.data.DataLoader(torch.arange(n), batch_size=bs, shuffle=shuffle)]).numpy(); st = torch.get_rng_state()
        for fn in (lambda: F.index_order(n, bs, shuffle), lambda: F._loader_order(n, shuffle)):
            torch.manual_seed(11)
            got = fn()
            assert got.dtype == np.int64 and np.array_equal(got, ref) and torch.equal(torch.get_rng_state(), st), (n, bs, shuffle)
    assert len(F.index_order(0, 5, False)) == 0
    with pytest.raises(ValueError):
        F.index_order(0, 5, True)


def test_topk_to_coo_matches_reference_topk_sparse():
    from opentf_amd.mdl.fnn import Fnn
    g = golden("g7_topk_sparse")
    probs = torch.from_numpy(g["probs"])
    v, i = torch.topk(probs, 5, dim=1)
    coo = Fnn._coo_from_topk(v.numpy(), i.numpy().astype(np.int32), tuple(probs.shape))
    assert coo.is_coalesced() and np.array_equal(coo.indices().numpy(), g["indices"]) and np.array_equal(coo.values().numpy(), g["values"])
    ref = Fnn._topk_sparse(probs, 5)
    assert torch.equal(ref.indices(), coo.indices()) and torch.equal(ref.values(), coo.values())


def test_shard_bounds_cover_the_batch():
    from opentf_amd.dp import shard_bounds
    for gB in [1, 2, 7, 1000, 1001]:
        for world in [1, 2, 3, 8]:
            spans = [shard_bounds(gB, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gB
            assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_dataset_statistics():
    from opentf_amd.synth import zipf_csr
    indptr, indices = zipf_csr(20000, 5000, 8.57, seed=1)
    nnz = np.diff(indptr)
    assert abs(nnz.mean() - 8.57) < 0.15 and nnz.min() >= 1
    for r in range(0, 20000, 997):  # sorted, distinct columns per row
        row = indices[indptr[r]:indptr[r + 1]]
        assert (np.diff(row) > 0).all()
    freq = np.sort(np.bincount(indices, minlength=5000))[::-1]
    assert freq[0] > 20 * max(freq[2500], 1)  # heavy tail
    i2, x2 = zipf_csr(20000, 5000, 8.57, seed=1)
    assert np.array_equal(indptr, i2) and np.array_equal(indices, x2)  # deterministic


def test_lil_ingestion_as_csr_and_validate():
    """teamsvecs.pkl holds scipy lil uint8 matrices (src/cmn/team.py:215,295): one pass over the row lists gives the CSR the engine wants."""
    import scipy.sparse
    from opentf_amd.cmn.team import lil_to_csr, validate
    rng = np.random.default_rng(0)
    dense = (rng.random((200, 37)) < 0.1).astype(np.uint8)
    dense[np.arange(200), 1 + np.arange(200) % 3] = 1        # no accidental empty teams
    dense[:, 5] = 0; dense[7] = 0
    lil = scipy.sparse.lil_matrix(dense)
    ip, ix, shape = lil_to_csr(lil)
    ref = scipy.sparse.csr_matrix(dense); ref.sort_indices()
    assert ip.dtype == np.int64 and ix.dtype == np.int32 and shape == (200, 37)
    assert np.array_equal(ip, ref.indptr) and np.array_equal(ix, ref.indices)
    ip2, ix2, _ = lil_to_csr(ref)
    assert np.array_equal(ip2, ip) and np.array_equal(ix2, ix)
    ok, msg = validate({"skill": lil, "member": lil})
    assert not ok and "have no skills" in msg and "[7]" in msg
    dense[7, 0] = 1
    ok, msg = validate({"skill": scipy.sparse.lil_matrix(dense), "member": scipy.sparse.lil_matrix(dense)})
    assert not ok and "used in no teams" in msg and "5" in msg
    dense[0, 5] = 1
    assert validate({"skill": scipy.sparse.lil_matrix(dense), "member": scipy.sparse.lil_matrix(dense)}) == (True, "")


def test_micro_auc_on_sparse_predictions_equals_sklearn_dense():
    """`calculate_auc_roc` (src/evl/metric.py:36-41) on top-K sparse predictions without the dense [n_test, M] pair the reference builds:
    equal to sklearn's micro-averaged roc_auc_score of the densified matrices, ties (the implicit zeros, repeated scores) included."""
    import scipy.sparse as sp
    from sklearn import metrics as skm
    from opentf_amd.evl.metric import calculate_auc_roc, micro_auc_sparse
    rng = np.random.default_rng(0)
    for n, M, K, dens in [(40, 300, 10, 0.02), (7, 50, 50, 0.2), (100, 64, 5, 0.1), (3, 1000, 1, 0.003)]:
        Y = (rng.random((n, M)) < dens).astype(np.uint8); Y[0, 0] = 1
        P = np.zeros((n, M), np.float32)
        for i in range(n):
            cols = rng.choice(M, K, replace=False)
            P[i, cols] = np.round(rng.random(K).astype(np.float32), 2 if K > 5 else 6)    # rounded: ties among stored scores too
            if K > 2: P[i, cols[0]] = 0.0                                                 # an explicitly stored zero
        ref = skm.roc_auc_score(Y, P, average="micro")
        S = sp.csr_matrix(P)
        got, curve = calculate_auc_roc(sp.csr_matrix(Y), S)
        assert curve is None and abs(got - ref) < 1e-12, (n, M, K, got, ref)
        S2 = sp.coo_matrix(P).tocsr()   # without the explicit zero: the entry is part of the implicit tie group
        assert abs(micro_auc_sparse(sp.lil_matrix(Y), S2) - ref) < 1e-12
    dense_auc, _ = calculate_auc_roc(sp.csr_matrix(Y), P)   # dense predictions keep the reference's sklearn route
    assert abs(dense_auc - ref) < 1e-12


def test_tntf_interval_folds_match_the_reference_and_the_committed_splits():
    """The per-interval K-fold of src/mdl/tntf.py:27-31: equal to what the reference's class wrote in its run here (g13) AND to the
    `splits.pkl` files the reference's authors committed for their temporal Bnn run on toy dblp."""
    from conftest import golden
    from opentf_amd.mdl.tntf import interval_folds
    g = golden("g13_tntf_dblp")
    year_idx = [(int(a), int(b)) for a, b in g["i2y"]]
    for i, (_, y) in enumerate(year_idx[:-1]):
        for k, (tr, va) in enumerate(interval_folds(year_idx, i, 3, 0)):
            assert np.array_equal(tr, g[f"{y}.train{k}"]) and np.array_equal(va, g[f"{y}.valid{k}"])
            assert np.array_equal(tr, g[f"committed.{y}.train{k}"]) and np.array_equal(va, g[f"committed.{y}.valid{k}"])
    assert np.array_equal(g["committed.2000.test"], np.arange(year_idx[-1][0], 31))
