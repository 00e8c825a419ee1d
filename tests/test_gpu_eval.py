"""train -> test -> evaluate through the plugin on a real MI355X: the eval stage writes the reference's files
(`src/mdl/ntf.py:32-92`) with the reference's metric names, and its numbers equal the metric oracle (pinned on the reference's
committed pytrec_eval results) applied to the very `.pred` files the plugin wrote."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse
import torch

from conftest import golden
from oracle import metric_oracle as MO
from test_gpu_plugin import Cfg, _toy

pytestmark = pytest.mark.gpu


def test_learn_test_evaluate_pipeline(tmp_path):
    from opentf_amd.mdl.bnn import Bnn
    tv, splits = _toy("dblp")
    g = golden("g10_metrics")
    n_exp, n_skill = tv["member"].shape[1], tv["skill"].shape[1]
    tv["skillcoverage"] = scipy.sparse.csr_matrix((np.ones(len(g["dblp.fnn.cov_indices"]), np.uint8), g["dblp.fnn.cov_indices"], g["dblp.fnn.cov_indptr"]),
                                                 shape=(n_exp, n_skill))
    cfg = Cfg(b=6, e=3, ns=3, lr=0.01, es=5, h=[32], spe=0, l="bce", tpw=10, tnw=1, nsd="uniform", nmc=3)
    m = Bnn(str(tmp_path), "cuda:0", 0, cfg)
    m.learn(tv, splits, None)
    m.test(tv, splits, Cfg(per_epoch=False, on_train=False, topK=8))   # sparse top-8 prediction files
    evalcfg = Cfg(topK=8, per_instance=True, on_train=False, per_epoch=False,
                  metrics=Cfg(trec=["P_2,5", "recall_2,5", "ndcg_cut_2,5", "map_cut_2,5", "success_2,5"], other=["skill_coverage_2,5", "aucroc"]))
    m.evaluate(tv, splits, evalcfg)
    Y = scipy.sparse.csr_matrix(tv["member"])[splits["test"]]
    X = scipy.sparse.csr_matrix(tv["skill"])[splits["test"]]
    cov = tv["skillcoverage"]
    fold_means = []
    for k in range(3):
        pr = torch.load(f"{m.output}/f{k}.test.pred", map_location="cpu", weights_only=False)["y_pred"].to_dense().numpy()
        cols, table = MO.instance_table(pr, Y.indptr, Y.indices, X.indptr, X.indices, cov.indptr, cov.indices, cutoffs=(2, 5), topK=8)
        inst = pd.read_csv(f"{m.output}/f{k}.test.pred.eval.instance.csv")
        assert list(inst.columns) == cols
        # zero-score experts outside the stored top-8 can only matter for ranks > 8; scores inside are distinct floats
        np.testing.assert_allclose(inst.values, table, atol=6e-6)
        mean = pd.read_csv(f"{m.output}/f{k}.test.pred.eval.mean.csv", index_col=0)
        assert mean.index.name == "metrics" and list(mean.columns) == ["mean"]
        assert list(mean.index) == cols[:10] + ["aucroc"] + cols[10:]   # the reference's row order: trec, aucroc, skill coverage
        np.testing.assert_allclose(mean.loc[cols, "mean"].values, table.mean(0), atol=1e-6)
        assert 0.0 <= mean.loc["aucroc", "mean"] <= 1.0
        fold_means.append(mean["mean"])
    agg = pd.read_csv(f"{m.output}/test.pred.eval.mean.csv", index_col=0)
    assert list(agg.columns) == ["mean", "std"]
    np.testing.assert_allclose(agg["mean"].values, pd.concat(fold_means, axis=1).mean(axis=1).values, rtol=1e-9)


def test_device_topk_goes_straight_into_the_rank_metrics():
    """`ntf_forward_topk`'s ranked ids fed to `ntf_rank_metrics` without a prediction matrix on the host (metric.calculate_metrics(ranked=...))
    give the same table as the file route (sparse top-K matrix -> ranking on the host), and aucroc+ writes the curve the reference pickles."""
    from opentf_amd import libntf
    from opentf_amd.evl import metric
    from opentf_amd.synth import init_params, zipf_csr
    N, S, M, B, K = 400, 300, 5000, 256, 20
    s_ip, s_ix = zipf_csr(N, S, 4.0, 1); m_ip, m_ix = zipf_csr(N, M, 3.0, 2)
    e = libntf.Engine([S, 64, M], input_mode=libntf.INPUT_MULTIHOT, max_batch=B, ns=0, nsd=None)
    e.set_skill_csr((s_ip, s_ix)); e.load_state_dict(init_params([S, 64, M], False, 1))
    rows = np.arange(B)
    vals, idx = e.forward_topk(rows, K)
    Y = scipy.sparse.csr_matrix((np.ones(len(m_ix)), m_ix, m_ip), shape=(N, M))[rows]
    names = ["P_2,5,10", "recall_2,5,10", "ndcg_cut_2,5,10", "map_cut_2,5,10", "success_2,5,10"]
    df_a, mean_a = metric.calculate_metrics(Y, None, K, True, names, ranked=idx)
    Y_ = scipy.sparse.csr_matrix((vals.ravel(), (np.repeat(rows, K), idx.ravel())), shape=(B, M))
    df_b, mean_b = metric.calculate_metrics(Y, Y_, K, True, names)
    assert np.array_equal(df_a.values, df_b.values) and np.array_equal(mean_a.values, mean_b.values)
    auc, curve = metric.calculate_auc_roc(Y, Y_, curve=True)
    auc2, none = metric.calculate_auc_roc(Y, Y_)
    assert none is None and abs(auc - auc2) < 1e-12 and len(curve) == 2 and curve[0][0] == 0.0 and curve[1][-1] == 1.0
