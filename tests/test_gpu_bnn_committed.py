"""The HIP engine's NATIVE Flipout generators (Philox eps, hashed signs, unigram_b sampler), KL and entropy kernels held to the same
reference-committed bayesian-torch outputs as the oracle (tests/test_bnn_committed.py explains what each quantity pins):
g12_bnn_committed.npz = 40 checkpoints + `.pred` files of the reference's toy Bnn runs.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

NMC, NGROUP = 10, 160


def _csr(g, ds, which):
    return g[f"{ds}.{which}_indptr"].astype(np.int64), g[f"{ds}.{which}_indices"].astype(np.int32)


def _cross(z):
    b = z.shape[0]
    s = z.sum(0)
    return float(((s * s - (z * z).sum(0)) / (b * (b - 1))).mean())


def _entropy(p):
    return -np.sum(p * np.log(p + 1e-15), axis=-1)


@pytest.fixture(scope="module")
def engines():
    from opentf_amd import libntf
    g = golden("g12_bnn_committed")
    out = {}
    for ds in ("dblp", "imdb", "gith", "uspt"):
        _, S, M = (int(v) for v in g[f"{ds}.shape"])
        e = libntf.Engine([S, 128, M], bayesian=True, input_mode=libntf.INPUT_MULTIHOT, max_batch=64, ns=5, nsd="unigram_b", tpw=10.0, tnw=1.0)
        e.set_skill_csr(_csr(g, ds, "skill")); e.set_member(_csr(g, ds, "member"))
        out[ds] = e
    return g, out


def _load(g, e, tag):
    e.load_state_dict({k[len(tag) + 3:]: g[k] for k in g.files if k.startswith(tag + ".p.")})


def test_native_mc_inference_reproduces_the_committed_predictions(engines):
    g, eng = engines
    zs, mi_c, mi_o, cross, pe_self = [], [], [], [], []
    for i, tag in enumerate(g["runs"]):
        ds = tag.split(".")[0]
        e = eng[ds]
        _load(g, e, tag)
        rows = g[f"{ds}.test"]
        means, mis = [], []
        for s in range(NGROUP):
            e.set_seed(1000 * i + s, 0)
            p, pu, mu = e.forward(rows, nmc=NMC, uncertainty=True)
            means.append(p); mis.append(mu)
            if s == 0:  # the engine's predictive entropy is the reference's function of the engine's own MC mean
                pe_self.append(np.abs(pu - _entropy(p)).max() / np.abs(pu).max())
        means = np.stack(means)
        z = (g[f"{tag}.y_pred"] - means.mean(0)) / means.std(0, ddof=1)
        zs.append(z.ravel()); cross.append(_cross(z))
        mi_c.append(g[f"{tag}.unc_model"]); mi_o.append(np.mean(mis, axis=0))
    assert max(pe_self) < 1e-5
    z = np.concatenate(zs)
    assert abs(z.mean()) < 0.04 and 0.93 < np.sqrt((z ** 2).mean()) < 1.10, (z.mean(), np.sqrt((z ** 2).mean()))
    per = np.array([np.median(np.abs(q)) / 0.6745 for q in zs])
    assert per.min() > 0.65 and per.max() < 1.35, per
    ratio = np.concatenate(mi_c).sum() / np.concatenate(mi_o).sum()
    assert abs(ratio - 1) < 0.04, ratio                       # perturbation variance = softplus(rho)^2 x |x|^2, to a few per cent
    c = np.array(cross)
    assert abs(c.mean()) < 4 * c.std() / np.sqrt(len(c)) + 0.01, c.mean()   # signs decorrelate the rows (a shared perturbation gives ~0.36)


def test_native_eval_loss_distribution_contains_the_committed_valid_loss(engines):
    g, eng = engines
    zs = []
    for i, tag in enumerate(g["runs"]):
        ds, fold = tag.split(".")[0], int(tag.split(".")[1][1:])
        if ds == "uspt" and fold > 0:
            continue  # fewer candidates than ns in a row: torch.multinomial's degenerate case, see tests/test_bnn_committed.py
        e = eng[ds]
        _load(g, e, tag)
        rows = g[f"{ds}.valid{fold}"]
        ls = []
        for s in range(200):
            e.set_seed(7000 * i + s, 0)
            ls.append(e.eval_step(rows))
        ls = np.array(ls)
        zs.append((float(g[f"{tag}.v_loss"]) - ls.mean()) / ls.std(ddof=1))
    zs = np.array(zs)
    assert len(zs) >= 34
    # the KL / B term is 2.5-4 sigma of this distribution (measured in the CPU test): a wrong KL form cannot pass
    assert abs(zs.mean()) < 0.5 and np.sqrt((zs ** 2).mean()) < 1.5 and np.abs(zs).max() < 4.5, zs
