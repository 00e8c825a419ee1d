"""node2vec producer, CPU side: the restatement of torch_geometric's Node2Vec (oracle/n2v_oracle.py) held to the tables and losses the
reference's authors committed for toy dblp (g14), and the product's graph builder against the oracle's."""
import numpy as np
import pytest
import scipy.sparse
import torch

from conftest import golden
from oracle import n2v_oracle as N


def _toy():
    toy = golden("toy_dblp")
    n, S, M = [int(v) for v in toy["shape"]]
    return toy, n, S, M


def test_stm_graph_matches_the_oracle_and_the_reference_node_counts():
    from opentf_amd.mdl.emb.gnn import member_team_edges, stm_graph
    toy, n, S, M = _toy()
    skill = scipy.sparse.csr_matrix((np.ones(len(toy["skill_indices"])), toy["skill_indices"], toy["skill_indptr"]), shape=(n, S))
    member = scipy.sparse.csr_matrix((np.ones(len(toy["member_indices"])), toy["member_indices"], toy["member_indptr"]), shape=(n, M))
    drop = np.concatenate([toy["test"], toy["valid0"]])
    rp, col, off, nn = stm_graph(skill, member, drop)
    rp2, col2, off2, nn2 = N.build_graph_sized(toy["skill_indptr"], toy["skill_indices"], toy["member_indptr"], toy["member_indices"], S, M, drop)
    assert nn == nn2 == golden("g14_n2v_dblp")["f0.embedding.weight"].shape[0] == S + M + n     # 54 nodes in the committed table
    assert off == off2 and np.array_equal(rp, rp2) and np.array_equal(col, col2)
    # undirected; no member-team edge of a dropped team; every skill-team edge kept
    A = scipy.sparse.csr_matrix((np.ones(len(col)), col, rp), shape=(nn, nn))
    assert (A != A.T).nnz == 0
    for t in drop:
        nb = col[rp[off["team"] + t]:rp[off["team"] + t + 1]]
        assert (nb < off["member"]).all()
    assert A[off["skill"]:off["member"]].nnz == len(toy["skill_indices"])
    src, dst = member_team_edges(member, toy["valid0"], off)
    assert len(src) == 2 * int(member[toy["valid0"]].nnz) and A[src, dst].sum() == 0


def test_oracle_training_ends_at_the_committed_loss_level():
    """The committed runs: b=1000 (one batch of all 54 nodes per epoch), 100 epochs, lr 0.001, ns 5, w 5, wl 5, wn 10, d 128 -> t_loss 7.06-7.51.
    N(0,1) rows of d = 128 give dot products ~ N(0, 128): the loss starts near 2 * sqrt(128 / 2 pi) = 9 and 100 Adam steps of 1e-3 bring it to ~7;
    a sum instead of a mean, a missing negative term, or walks that ignore the graph would all land elsewhere."""
    toy, n, S, M = _toy()
    g = golden("g14_n2v_dblp")
    finals, firsts = [], []
    for k in range(3):
        drop = np.concatenate([toy["test"], toy[f"valid{k}"]])
        rp, col, off, nn = N.build_graph_sized(toy["skill_indptr"], toy["skill_indices"], toy["member_indptr"], toy["member_indices"], S, M, drop)
        W, hist = N.train(rp, col, nn, 128, 1000, 100, 0.001, 5, 5, 10, 5, seed=k)
        finals.append(hist[-1]); firsts.append(hist[0])
        # committed tables are still close to their N(0,1) draw: same scale of the rows
        assert abs(float(W.std()) - float(g[f"f{k}.embedding.weight"].std())) < 0.05
    committed = [float(g[f"f{k}.t_loss"]) for k in range(3)]
    assert all(int(g[f"f{k}.e"]) == 99 for k in range(3))
    assert abs(np.mean(finals) - np.mean(committed)) < 0.6, (finals, committed)
    assert 8.0 < np.mean(firsts) < 10.5


def test_oracle_sampling_shapes_and_walk_validity():
    toy, n, S, M = _toy()
    rp, col, off, nn = N.build_graph_sized(toy["skill_indptr"], toy["skill_indices"], toy["member_indptr"], toy["member_indices"], S, M)
    gen = torch.Generator().manual_seed(0)
    batch = torch.arange(nn)
    pos = N.pos_sample(rp, col, batch, 6, 4, 3, gen)
    neg = N.neg_sample(nn, batch, 6, 4, 3, 2, gen)
    assert tuple(pos.shape) == (nn * 3 * 3, 4) and tuple(neg.shape) == (nn * 3 * 2 * 3, 4)       # 6 + 1 - 4 = 3 windows per walk
    A = scipy.sparse.csr_matrix((np.ones(len(col)), col, rp), shape=(nn, nn)).toarray()
    p = pos.numpy()
    assert all(A[a, b] == 1 or a == b for row in p for a, b in zip(row[:-1], row[1:]))


def test_the_references_graph_file_and_table_are_read_without_torch_geometric(tmp_path):
    """VERDICT r4 missing #3.  tests/golden/ref_toy_dblp_n2v/ holds two DATA files the reference's authors committed for toy dblp: `stm.add.graph.pkl` (the pickled
    HeteroData, src/mdl/emb/gnn.py:21-23,58-59) and the table `n2v.../f0.pt` trained on it (gnn.py:453).  opentf_amd/mdl/emb/pyg_reader.py reads both with restricted
    unpicklers - neither torch_geometric nor omegaconf is installed here: the node stores in THEIR order (a Python set's iteration order when the graph was built:
    member, team, skill in this file), the weight bit for bit what make_golden_n2v.py extracted with torch in the build container (g14), and the rows re-stacked into
    the plugin's [skill | member | team]."""
    import os
    import pickle
    from conftest import GOLDEN
    from opentf_amd.mdl.emb import pyg_reader as R
    d = os.path.join(GOLDEN, "ref_toy_dblp_n2v")
    toy, n, S, M = _toy()
    blocks = R.node_blocks(f"{d}/stm.add.graph.pkl")
    assert blocks == [("member", M), ("team", n), ("skill", S)] == [("member", 13), ("team", 31), ("skill", 10)]
    t = R.reference_table(f"{d}/f0.pt")
    g = golden("g14_n2v_dblp")
    np.testing.assert_array_equal(t["weight"], g["f0.embedding.weight"])
    assert t["e"] == int(g["f0.e"]) and t["t_loss"] == float(g["f0.t_loss"]) and t["v_loss"] == float(g["f0.v_loss"])
    W, counts = R.blocks_to_order(t["weight"], blocks, ["skill", "member", "team"])
    assert counts == {"skill": S, "member": M, "team": n}
    np.testing.assert_array_equal(W[:S], t["weight"][M + n:]); np.testing.assert_array_equal(W[S:S + M], t["weight"][:M]); np.testing.assert_array_equal(W[S + M:], t["weight"][M:M + n])
    with pytest.raises(RuntimeError, match="do not add up"): R.blocks_to_order(t["weight"][:-1], blocks, ["skill", "member", "team"])
    with pytest.raises(RuntimeError, match="not in the graph"): R.blocks_to_order(t["weight"], blocks, ["skill", "loc"])

    # no code of a pickled class runs: a graph file whose "class" is os.system comes out as an inert holder, not as a call
    class Evil:
        def __reduce__(self): return (os.system, (f"touch {tmp_path}/pwned",))
    evil = tmp_path / "evil.graph.pkl"
    evil.write_bytes(pickle.dumps(Evil(), protocol=4))
    with pytest.raises(RuntimeError, match="_node_store_dict"): R.node_blocks(str(evil))
    assert not (tmp_path / "pwned").exists()


def test_a_plugin_table_is_recognised_under_the_restricted_load_whatever_its_cfg_is(tmp_path):
    """ADVICE r5 (medium).  `_load_fold` decides "this plugin's table" from the node_order / node_offsets markers, which are plain builtins and come through the restricted
    unpickler - it no longer re-opens the file with an unrestricted torch.load (whose failure on an un-unpicklable cfg used to demote the table to "the reference's" and
    re-stack its rows by a graph file lying beside it).  Here the cfg is an instance of a class that no longer exists at load time, and a payload that would run code on
    an unrestricted load sits in the file: the table is taken as the plugin's, rows untouched, nothing runs."""
    import os, sys, types
    from opentf_amd.mdl.emb import pyg_reader as R
    from opentf_amd.mdl.emb.gnn import Gnn, NODE_ORDER
    mod = types.ModuleType("gone_cfg_module")
    class GoneCfg(dict): pass
    GoneCfg.__module__ = "gone_cfg_module"; GoneCfg.__qualname__ = "GoneCfg"; mod.GoneCfg = GoneCfg
    class Evil:
        def __reduce__(self): return (os.system, (f"touch {tmp_path}/pwned",))
    W = np.arange(12 * 4, dtype=np.float32).reshape(12, 4)
    off = {"skill": 0, "member": 3, "team": 7}
    sys.modules["gone_cfg_module"] = mod
    try:
        torch.save({"model_state_dict": {"embedding.weight": torch.from_numpy(W)}, "cfg": GoneCfg(a=1), "f": 0, "e": 2, "t_loss": 1.0, "v_loss": 2.0, "evil": Evil(),
                    "node_order": NODE_ORDER, "node_offsets": (0, 3, 7, 12)}, tmp_path / "f0.pt")
    finally:
        del sys.modules["gone_cfg_module"]
    t = R.reference_table(str(tmp_path / "f0.pt"))
    assert t["node_order"] == NODE_ORDER and t["node_offsets"] == (0, 3, 7, 12) and not (tmp_path / "pwned").exists()
    g = Gnn.__new__(Gnn); g.graph_file = str(tmp_path / "stm.add.graph.pkl")       # (absent: a table taken for the reference's would be refused)
    np.testing.assert_array_equal(g._load_fold(str(tmp_path / "f0.pt"), off, 12), W)
    assert not (tmp_path / "pwned").exists()
    with pytest.raises(RuntimeError, match="node offsets"): g._load_fold(str(tmp_path / "f0.pt"), {"skill": 0, "member": 4, "team": 7}, 12)
    # a table without the markers and without its graph file stays refused
    torch.save({"model_state_dict": {"embedding.weight": torch.from_numpy(W)}, "cfg": None, "e": 0, "t_loss": 0.0, "v_loss": 0.0}, tmp_path / "f1.pt")
    with pytest.raises(RuntimeError, match="graph file"): g._load_fold(str(tmp_path / "f1.pt"), off, 12)
