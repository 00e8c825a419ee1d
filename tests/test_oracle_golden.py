"""The oracle (oracle/ntf_oracle.py) pinned against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import scipy.sparse
import torch

from conftest import GOLDEN, golden, params_from
from oracle import ntf_oracle as O


@pytest.mark.parametrize("name", ["g1_forward_imdb", "g1_forward_mid", "g1_forward_2h"])
def test_forward_matches_reference(name):
    g = golden(name)
    sd = params_from(g, "p.")
    out = O.fnn_forward(sd, torch.from_numpy(g["X"]))
    assert torch.equal(out, torch.from_numpy(g["logits"]))  # same torch ops -> bitwise


@pytest.mark.parametrize("nsd", ["uniform", "unigram", "unigram_b", "None"])
def test_bxe_and_sampler_match_reference(nsd):
    g = golden(f"g2_bxe_{nsd}")
    y_, y = torch.from_numpy(g["y_"]), torch.from_numpy(g["y"])
    torch.manual_seed(int(g["seed"]))
    unigram = torch.from_numpy(g["unigram"]) if nsd == "unigram" else None
    idx = O.draw_negatives(y, None if nsd == "None" else nsd, 5, unigram)
    if nsd != "None":
        assert np.array_equal(idx.numpy(), g["idx"])  # same RNG stream as the reference's sampler
        # invariants the native samplers are held to as well
        assert all(len(set(r)) == 5 for r in g["idx"].tolist())
        assert (g["y"][np.arange(len(g["idx"]))[:, None], g["idx"]] == 0).all()
    loss = O.bxe(y_, y, idx, float(g["tpw"]), float(g["tnw"]))
    assert torch.equal(loss, torch.from_numpy(g["loss"]))


def test_sampler_edge_cases():
    g = golden("g3_unigram_b_fallback")
    torch.manual_seed(int(g["seed"]))
    assert np.array_equal(O.ns_unigram_batch(torch.from_numpy(g["y"]), 5).numpy(), g["idx"])
    g = golden("g3_uniform_fewneg")
    torch.manual_seed(int(g["seed"]))
    idx = O.ns_uniform(torch.from_numpy(g["y"]), 3).numpy()
    assert np.array_equal(idx, g["idx"])
    # row 0 has one negative only: the other picks are positives (reference behaviour, fnn.py:54)
    assert (g["y"][0, idx[0]] == 1).sum() == 2


@pytest.mark.parametrize("tag", ["imdb", "mid"])
def test_train_step_matches_reference(tag):
    g = golden(f"g4_step_{tag}")
    sd = params_from(g, "p0.")
    X, y = torch.from_numpy(g["X"]), torch.from_numpy(g["y"])
    opt = O.Adam(sd, float(g["lr"]))
    for s in range(3):
        logits = O.fnn_forward(sd, X)
        assert torch.allclose(logits, torch.from_numpy(g[f"s{s}.logits"]), rtol=1e-6, atol=1e-7)
        idx = torch.from_numpy(g[f"s{s}.idx"])
        loss, grads = O.train_step(sd, opt, X, y, idx, float(g["tpw"]), float(g["tnw"]))
        assert abs(loss - float(g[f"s{s}.loss"])) <= 1e-6 * abs(loss)
        for k in sd:
            np.testing.assert_allclose(grads[k].numpy(), g[f"s{s}.g.{k}"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(sd[k].numpy(), g[f"s{s}.p.{k}"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("nsd", ["None", "uniform", "unigram", "unigram_b"])
def test_learn_driver_matches_reference(nsd):
    """Whole Fnn.learn on toy dblp: same seed -> same init, batch order, negatives, loss series, LR
    series, early-stop epoch and final weights as the reference."""
    g = golden(f"g5_learn_dblp_{nsd}")
    cfg = json.loads(str(g["cfg"]))
    toy = golden("toy_dblp")
    n, S, M = toy["shape"]
    skill = scipy.sparse.csr_matrix((np.ones(len(toy["skill_indices"]), np.uint8), toy["skill_indices"], toy["skill_indptr"]), shape=(n, S)).tolil()
    member = scipy.sparse.csr_matrix((np.ones(len(toy["member_indices"]), np.uint8), toy["member_indices"], toy["member_indptr"]), shape=(n, M)).tolil()
    splits = {"test": toy["test"], "folds": {k: {"train": toy[f"train{k}"], "valid": toy[f"valid{k}"]} for k in range(3)}}
    torch.manual_seed(0)  # Ntf.__init__ -> set_seed(seed) once per model object (ntf.py:14)
    res = O.learn(skill, member, splits, cfg)
    scalars = json.loads(str(g["scalars"]))
    lr_ref = list(g["lr"])
    pos = 0
    for k in range(3):
        t_ref = [v for tag, v, _ in scalars if tag == f"{k}_t_loss"]
        v_ref = [v for tag, v, _ in scalars if tag == f"{k}_v_loss"]
        assert res[k]["last_epoch"] == int(g[f"f{k}.e"]) and len(t_ref) == len(res[k]["t_loss"])
        np.testing.assert_allclose(res[k]["t_loss"], t_ref, rtol=2e-6)
        np.testing.assert_allclose(res[k]["v_loss"], v_ref, rtol=2e-6)
        np.testing.assert_allclose(res[k]["lr"], lr_ref[pos:pos + len(t_ref)], rtol=1e-12)
        pos += len(t_ref)
        for name, v in res[k]["state"].items():
            np.testing.assert_allclose(v.numpy(), g[f"f{k}.{name}"], rtol=1e-4, atol=1e-6)
        Xt = O.dense_rows(skill, splits["test"])
        np.testing.assert_allclose(O.predict(res[k]["state"], Xt).numpy(), g[f"f{k}.y_pred"], rtol=1e-4, atol=1e-6)


def test_schedulers_match_reference():
    with open(os.path.join(GOLDEN, "g6_sched.json")) as f:
        ref = json.load(f)
    for name, r in ref.items():
        es, sch = O.EarlyStopping(3, 0.001), O.ReduceLROnPlateau(0.001)
        for v, (lr, counter, stop) in zip(r["seq"], r["rows"]):
            got_lr = sch.step(v)
            es(v)
            assert got_lr == pytest.approx(lr, rel=1e-12) and es.counter == counter and es.early_stop == stop, name


def test_adam_matches_torch():
    torch.manual_seed(0)
    p = {"w": torch.randn(5, 3)}
    ref = torch.nn.Parameter(p["w"].clone())
    topt = torch.optim.Adam([ref], lr=0.01)
    opt = O.Adam(p, 0.01)
    for _ in range(5):
        g = torch.randn(5, 3)
        ref.grad = g.clone(); topt.step()
        opt.step(p, {"w": g})
    assert torch.allclose(p["w"], ref.detach(), rtol=1e-6, atol=1e-7)


def test_topk_sparse_matches_reference():
    g = golden("g7_topk_sparse")
    sp = O.topk_sparse(torch.from_numpy(g["probs"]), 5)
    assert np.array_equal(sp.indices().numpy(), g["indices"]) and np.array_equal(sp.values().numpy(), g["values"])


def test_gather_matches_reference():
    g = golden("g9_gather_dblp")
    n = len(g["indptr"]) - 1
    skill = scipy.sparse.csr_matrix((np.ones(len(g["indices"]), np.uint8), g["indices"], g["indptr"]), shape=(n, int(g["n_skill"])))
    X = O.gather_meanpool(skill, g["table"])
    np.testing.assert_allclose(X, g["X"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.gather_meanpool_fast(g["indptr"], g["indices"], g["table"]), g["X"], rtol=1e-6, atol=1e-7)


def test_layout_pins():
    with open(os.path.join(GOLDEN, "g8_layout.json")) as f:
        lay = json.load(f)
    bnn = next(v for k, v in lay.items() if k.startswith("bnn."))
    assert bnn["ckpt_keys"] == ["model_state_dict", "cfg", "f", "e", "t_loss", "v_loss"]
    torch.manual_seed(0)
    sd = O.bnn_init(10, [128], 13)
    assert list(sd.keys()) == list(bnn["state"].keys())
    assert [list(v.shape) for v in sd.values()] == [v[0] for v in bnn["state"].values()]
    assert bnn["pred_keys"] == ["y_pred", "uncertainty"] and set(bnn["uncertainty"].keys()) == {"pred", "model"}
    with open(os.path.join(GOLDEN, "g5_dirname.json")) as f:
        d = json.load(f)
    assert O.model_dirname("Fnn", d["cfg"]) == d["name"]


def test_flipout_self_consistency():
    """UNPINNED part: the restated Flipout reduces to the deterministic layer when rho -> -inf, and the
    KL of the prior against itself is 0."""
    torch.manual_seed(1)
    sd = O.bnn_init(6, [8], 5)
    x = torch.randn(4, 6)
    noise = O.draw_flipout_noise(sd, 4)
    det = {k: (v if "rho" not in k else torch.full_like(v, -60.0)) for k, v in sd.items()}
    plain = {"layers.0.weight": sd["layers.0.mu_weight"], "layers.0.bias": sd["layers.0.mu_bias"],
             "layers.1.weight": sd["layers.1.mu_weight"], "layers.1.bias": sd["layers.1.mu_bias"]}
    assert torch.allclose(O.bnn_forward(det, x, noise), O.fnn_forward(plain, x), atol=1e-6)
    rho1 = float(np.log(np.expm1(1.0)))
    prior = {k: (torch.zeros_like(v) if "mu" in k else torch.full_like(v, rho1)) for k, v in sd.items()}
    assert abs(float(O.get_kl_loss(prior))) < 1e-6


# ------------------------------------------------------------------------------------------ member-skill co-occurrence (§8f rank 3)
@pytest.mark.parametrize("name", ["dblp", "imdb", "uspt", "gith", "wrap", "rand"])
def test_cooc_oracle_matches_reference_skillcoverage(name):
    """dblp/imdb/uspt/gith: the `skillcoverage.pkl` files committed by the reference's authors; wrap/rand: the reference's expression
    (src/cmn/team.py:327-335) run on synthetic lil matrices whose counts pass 255 (uint8 wrap, zero sums dropped)."""
    from oracle import cooc_oracle as CO
    g = golden("g11_cooc")
    n, M, S = [int(v) for v in g[f"{name}.shape"]]
    ip, ix, data = CO.skill_cooccurrence(g[f"{name}.m_indptr"], g[f"{name}.m_indices"], g[f"{name}.s_indptr"], g[f"{name}.s_indices"], M, S, g[f"{name}.skip"])
    assert np.array_equal(ip, g[f"{name}.c_indptr"]) and np.array_equal(ix, g[f"{name}.c_indices"]) and np.array_equal(data, g[f"{name}.c_data"])
    assert data.dtype == np.uint8 and (data != 0).all()
