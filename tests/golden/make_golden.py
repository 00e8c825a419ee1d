#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE ITSELF.

Runs only in the build container, where /root/reference exists.  Nothing of the reference is copied:
the outputs are data (inputs + expected outputs of the reference's own functions).  The GPU box and
the test-suite read only the .npz/.json files this script wrote.

    python tests/golden/make_golden.py            # rewrites every fixture

Shims: stub `omegaconf` and `tensorboardX` modules, `np.Inf`, a harmless
`pkgmgr.install_import` (the real one pip-uninstalls mismatching packages), `verbose=` swallowed by
ReduceLROnPlateau, `torch.load(weights_only=False)`.
"""
import importlib
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import scipy.sparse
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class Cfg(dict):
    """attribute-dict standing in for an omegaconf DictConfig"""
    def __getattr__(self, k):
        try: return self[k]
        except KeyError: raise AttributeError(k)
    __setattr__ = dict.__setitem__


def install_shims():
    om = types.ModuleType("omegaconf")

    class OmegaConf:
        @staticmethod
        def to_container(cfg, resolve=True): return dict(cfg)
        @staticmethod
        def create(d): return Cfg(d)
    om.OmegaConf = OmegaConf
    sys.modules["omegaconf"] = om

    tb = types.ModuleType("tensorboardX")

    class SummaryWriter:
        scalars = []
        def __init__(self, log_dir=None): pass
        def add_scalar(self, tag, scalar_value, global_step): SummaryWriter.scalars.append((tag, float(scalar_value), int(global_step)))
        def close(self): pass
    tb.SummaryWriter = SummaryWriter
    sys.modules["tensorboardX"] = tb

    if not hasattr(np, "Inf"): np.Inf = np.inf

    os.chdir(f"{REF}/src")
    sys.path.insert(0, f"{REF}/src")
    import pkgmgr

    def safe_install_import(pkg_name, import_path=None, from_module=None):
        module = importlib.import_module(import_path or pkg_name)
        return getattr(module, from_module) if from_module else module
    pkgmgr.install_import = safe_install_import
    pkgmgr.install_pkg = pkgmgr.reinstall_pkg = lambda *_a, **_k: (_ for _ in ()).throw(RuntimeError("no installs"))

    lr_log = []
    _RLP = torch.optim.lr_scheduler.ReduceLROnPlateau

    class RLP(_RLP):
        def __init__(self, optimizer, *a, verbose=None, **k): super().__init__(optimizer, *a, **k)
        def step(self, metrics, *a, **k):
            super().step(metrics, *a, **k)
            lr_log.append(float(self.optimizer.param_groups[0]["lr"]))
    torch.optim.lr_scheduler.ReduceLROnPlateau = RLP

    _load = torch.load
    torch.load = lambda *a, **k: _load(*a, **{**k, "weights_only": k.get("weights_only", False)})
    return SummaryWriter, lr_log


def sd_np(sd): return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def main():
    SummaryWriter, lr_log = install_shims()
    from mdl.fnn import Fnn
    from mdl.ntf import Ntf
    from mdl.earlystopping import EarlyStopping
    import pkgmgr
    tmp = tempfile.mkdtemp(prefix="golden_")

    def mk(cfg, seed=0): return Fnn(tmp, "cpu", seed, Cfg(cfg))
    base = dict(b=8, e=6, ns=3, lr=0.001, es=5, h=[32], spe=0, l="bce", tpw=10, tnw=1, nsd="uniform")

    def load_toy(ds):
        with open(f"{REF}/output/{ds}/teamsvecs.pkl", "rb") as f: tv = pickle.load(f)
        with open(f"{REF}/output/{ds}/splits.f3.r0.85.pkl", "rb") as f: sp = pickle.load(f)
        return tv, sp
    imdb, imdb_sp = load_toy("imdb/toy.title.basics.tsv")
    dblp, dblp_sp = load_toy("dblp/toy.dblp.v12.json")

    def save(name, **arrs):
        np.savez_compressed(f"{HERE}/{name}.npz", **arrs)
        print("wrote", name, {k: getattr(v, "shape", v) for k, v in arrs.items()})

    # ---- G1 forward: toy imdb (18 -> 32 -> 112) and a synthetic mid-size (128 -> 128 -> 4096)
    m = mk(base)
    model = m.init(18, 112)
    X = torch.as_tensor(imdb["skill"].toarray()).float()
    with torch.no_grad(): out = model.forward(X)
    save("g1_forward_imdb", X=X.numpy(), logits=out.numpy(), **{f"p.{k}": v for k, v in sd_np(model.state_dict()).items()})

    m = mk({**base, "h": [128]}, seed=1)
    model = m.init(128, 4096)
    X = torch.randn(64, 128)
    with torch.no_grad(): out = model.forward(X)
    save("g1_forward_mid", X=X.numpy(), logits=out.numpy(), **{f"p.{k}": v for k, v in sd_np(model.state_dict()).items()})

    # two hidden layers
    m = mk({**base, "h": [48, 24]}, seed=2)
    model = m.init(18, 112)
    X = torch.as_tensor(imdb["skill"].toarray()).float()
    with torch.no_grad(): out = model.forward(X)
    save("g1_forward_2h", X=X.numpy(), logits=out.numpy(), **{f"p.{k}": v for k, v in sd_np(model.state_dict()).items()})

    # ---- G2/G3 bxe + samplers on toy imdb labels (M=112) with seeds
    y = torch.as_tensor(imdb["member"].toarray()).float()
    y_ = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
    for nsd in ["uniform", "unigram", "unigram_b", None]:
        m = mk({**base, "nsd": nsd, "ns": 5})
        if nsd == "unigram": m.unigram = torch.tensor(imdb["member"].sum(axis=0) / imdb["member"].shape[0])
        captured = {}
        for fn in ["ns_uniform", "ns_unigram"]:
            orig = getattr(m, fn)
            def wrap(yy, _o=orig, _n=fn):
                r = _o(yy); captured["idx"] = r.clone(); return r
            setattr(m, fn, wrap)
        torch.manual_seed(123)
        loss = m.bxe(y_, y)
        arrs = dict(y_=y_.numpy(), y=y.numpy(), loss=loss.numpy(), tpw=10.0, tnw=1.0, seed=123)
        if nsd:
            arrs["idx"] = captured["idx"].numpy()
        if nsd == "unigram": arrs["unigram"] = m.unigram.numpy()
        save(f"g2_bxe_{nsd}", **arrs)

    # fallback row: unigram_b where every negative has zero batch frequency (single-row batch)
    m = mk({**base, "nsd": "unigram_b", "ns": 5})
    y1 = y[:1].clone()
    torch.manual_seed(5)
    idx = m.ns_unigram_batch(y1)
    save("g3_unigram_b_fallback", y=y1.numpy(), idx=idx.numpy(), seed=5)

    # row with fewer than ns negatives for ns_uniform (may return positives)
    yfew = torch.ones(2, 6); yfew[0, 1] = 0; yfew[1, 2] = 0; yfew[1, 4] = 0
    m = mk({**base, "nsd": "uniform", "ns": 3})
    torch.manual_seed(11)
    idx = m.ns_uniform(yfew)
    save("g3_uniform_fewneg", y=yfew.numpy(), idx=idx.numpy(), seed=11)

    # ---- G4 one full train step with injected negatives (uniform) : params -> loss, grads, params'
    for tag, (D, H, M, B, seed) in {"imdb": (18, [32], 112, 19, 3), "mid": (128, [128], 1024, 48, 4)}.items():
        m = mk({**base, "h": H, "ns": 5}, seed=seed)
        model = m.init(D, M)
        if tag == "imdb":
            X = torch.as_tensor(imdb["skill"].toarray()).float(); yb = torch.as_tensor(imdb["member"].toarray()).float()
        else:
            X = torch.randn(B, D)
            yb = (torch.rand(B, M) < 0.004).float(); yb[torch.arange(B), torch.randint(0, M, (B,))] = 1
        p0 = sd_np(model.state_dict())
        opt = torch.optim.Adam(model.parameters(), lr=0.001)
        cap = {}
        orig = m.ns_uniform
        m.ns_uniform = lambda yy, _o=orig: cap.setdefault("idx", _o(yy))
        steps = []
        for s in range(3):
            cap.clear()
            opt.zero_grad()
            out = model.forward(X)
            loss = m.bxe(out, yb).sum(dim=1).mean()
            loss.backward()
            g = {k: v.grad.detach().numpy().copy() for k, v in model.named_parameters()}
            opt.step()
            steps.append((cap["idx"].numpy().copy(), float(loss.item()), g, sd_np(model.state_dict()), out.detach().numpy().copy()))
        arrs = dict(X=X.numpy(), y=yb.numpy(), tpw=10.0, tnw=1.0, lr=0.001)
        arrs.update({f"p0.{k}": v for k, v in p0.items()})
        for s, (idx, l, g, p, o) in enumerate(steps):
            arrs[f"s{s}.idx"] = idx; arrs[f"s{s}.loss"] = l; arrs[f"s{s}.logits"] = o
            arrs.update({f"s{s}.g.{k}": v for k, v in g.items()})
            arrs.update({f"s{s}.p.{k}": v for k, v in p.items()})
        save(f"g4_step_{tag}", **arrs)

    # ---- G5 multi-epoch toy dblp runs through the reference's own Fnn.learn + G7 Fnn.test
    testcfg = Cfg(per_epoch=False, on_train=False, topK=None)
    for nsd in [None, "uniform", "unigram", "unigram_b"]:
        cfg = {**base, "h": [16], "b": 5, "e": 8, "ns": 3, "nsd": nsd, "es": 2, "lr": 0.01}
        SummaryWriter.scalars.clear(); lr_log.clear()
        m = mk(cfg, seed=0)
        sp = {"test": dblp_sp["test"], "folds": {k: dict(v) for k, v in dblp_sp["folds"].items()}}
        m.learn(dblp, sp, None)
        arrs = {"scalars": json.dumps(SummaryWriter.scalars), "lr": np.array(lr_log), "cfg": json.dumps(cfg)}
        for k in sp["folds"]:
            ck = torch.load(f"{m.output}/f{k}.pt")
            arrs.update({f"f{k}.{n}": v.numpy() for n, v in ck["model_state_dict"].items()})
            arrs[f"f{k}.e"] = ck["e"]; arrs[f"f{k}.t_loss"] = ck["t_loss"]; arrs[f"f{k}.v_loss"] = ck["v_loss"]
        m.test(dblp, sp, testcfg)
        for k in sp["folds"]:
            pr = torch.load(f"{m.output}/f{k}.test.pred")
            arrs[f"f{k}.y_pred"] = pr["y_pred"].numpy()
            assert pr["uncertainty"] is None
        save(f"g5_learn_dblp_{nsd}", **arrs)
    # name of the output directory (cfg2str contract)
    with open(f"{HERE}/g5_dirname.json", "w") as f:
        json.dump({"cfg": cfg, "name": m.name()}, f)

    # ---- G6 EarlyStopping / ReduceLROnPlateau on scripted sequences
    seqs = {"flat": [1.0] * 12, "down": [1.0 / (i + 1) for i in range(12)], "bump": [1, .9, .95, .94, .8, .81, .82, .83, .84, .85, .7, .71],
            "tiny": [1 - 1e-5 * i for i in range(12)]}
    out = {}
    for name, seq in seqs.items():
        es = EarlyStopping(torch, patience=3, verbose=False, delta=0.001, save_model=False, trace_func=lambda *_: None)
        p = torch.nn.Parameter(torch.zeros(1)); opt = torch.optim.Adam([p], lr=0.001)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.1, patience=2)
        rows = []
        for v in seq:
            sch.step(v); es(v, None)
            rows.append([float(opt.param_groups[0]["lr"]), int(es.counter), bool(es.early_stop)])
        out[name] = {"seq": [float(s) for s in seq], "rows": rows}
    with open(f"{HERE}/g6_sched.json", "w") as f: json.dump(out, f)

    # ---- G7 topk_sparse
    probs = torch.rand(7, 50, generator=torch.Generator().manual_seed(9))
    sp_ = pkgmgr.topk_sparse(torch, probs, 5)
    save("g7_topk_sparse", probs=probs.numpy(), indices=sp_.indices().numpy(), values=sp_.values().numpy())

    # ---- G8 checkpoint / pred layout of committed files (keys, shapes, dtypes)
    lay = {}
    d = f"{REF}/output/dblp/toy.dblp.v12.json/splits.f3.r0.85"

    class _Stub:  # stands in for omegaconf classes pickled inside the committed checkpoints' 'cfg'
        def __init__(self, *a, **k): pass
        def __setstate__(self, st): self.__dict__["_st"] = st

    class _PM:  # pickle_module for torch.load: unknown globals -> _Stub
        __name__ = "stubpickle"
        Unpickler = None
        load = staticmethod(pickle.load)

    class _U(pickle.Unpickler):
        def find_class(self, mod, name):
            if mod.startswith("omegaconf"): return _Stub
            return super().find_class(mod, name)
    _PM.Unpickler = _U
    _tl = torch.load
    torch.load = lambda f, **k: _tl(f, pickle_module=_PM, **k)
    for sub in sorted(os.listdir(d)):
        if sub.startswith(("fnn.", "bnn.")):
            ck = torch.load(f"{d}/{sub}/f0.pt", map_location="cpu")
            pr = torch.load(f"{d}/{sub}/f0.test.pred", map_location="cpu")
            unc = pr["uncertainty"]
            lay[sub] = {"ckpt_keys": list(ck.keys()),
                        "state": {k: [list(v.shape), str(v.dtype)] for k, v in ck["model_state_dict"].items()},
                        "pred_keys": list(pr.keys()), "y_pred": [list(pr["y_pred"].shape), str(pr["y_pred"].dtype), bool(pr["y_pred"].is_sparse)],
                        "uncertainty": None if unc is None else {k: [[list(a.shape), str(a.dtype)] for a in v] for k, v in unc.items()}}
    with open(f"{HERE}/g8_layout.json", "w") as f: json.dump(lay, f, indent=1)

    # ---- G9 gather: the reference expression (gnn.py:485) on toy dblp with the committed n2v table
    n2v = torch.load(f"{d}/n2v.b1000.e100.ns5.lr0.001.es5.spe10.d128.add.stm.w5.wl5.wn10/f0.pt", map_location="cpu")
    emb = None
    for k, v in (n2v["model_state_dict"] if "model_state_dict" in n2v else n2v).items():
        if k.endswith("embedding.weight"): emb = v
    table = emb[:dblp["skill"].shape[1]].clone()  # first node type block; any [S, d] table pins the expression
    dense = (dblp["skill"] @ table) / dblp["skill"].sum(axis=1)
    csr = scipy.sparse.csr_matrix(dblp["skill"])
    save("g9_gather_dblp", indptr=csr.indptr, indices=csr.indices, table=table.numpy(), X=np.asarray(dense, dtype=np.float32),
         member_indptr=scipy.sparse.csr_matrix(dblp["member"]).indptr, member_indices=scipy.sparse.csr_matrix(dblp["member"]).indices,
         n_skill=dblp["skill"].shape[1], n_member=dblp["member"].shape[1])
    # toy data themselves (CSR) so GPU-box tests can run the toy configs without the reference tree
    for name, tv, sp in [("imdb", imdb, imdb_sp), ("dblp", dblp, dblp_sp)]:
        s, mm = scipy.sparse.csr_matrix(tv["skill"]), scipy.sparse.csr_matrix(tv["member"])
        arrs = dict(skill_indptr=s.indptr, skill_indices=s.indices, member_indptr=mm.indptr, member_indices=mm.indices,
                    shape=np.array([s.shape[0], s.shape[1], mm.shape[1]]), test=sp["test"])
        for k, v in sp["folds"].items(): arrs[f"train{k}"] = v["train"]; arrs[f"valid{k}"] = v["valid"]
        save(f"toy_{name}", **arrs)


if __name__ == "__main__":
    main()
