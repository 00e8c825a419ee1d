"""CPU oracle of the node2vec producer.  TEST INFRASTRUCTURE ONLY: nothing under opentf_amd/ imports it.

The reference trains its node2vec table with `torch_geometric.nn.Node2Vec` (src/mdl/emb/gnn.py:153-168,401-453; torch_geometric==2.6.1
torch_cluster==1.6.3, requirements.txt:51), a third-party dependency that is neither vendored nor installable here.  This file restates
its published algorithm (Node2Vec.pos_sample / neg_sample / loss, p = q = 1 as src/mdl/emb/__config__.yaml:69-70 sets them) with torch
autograd for the gradient, and the graph the reference feeds it (gnn.py:30-60,84-131: 'stm' structure, ToUndirected, homogeneous node ids
by node-type offsets, member-team edges of the test / validation teams removed).

Pinning: no value-level fixture of PyG's outputs exists in the reference tree (random walks under torch's RNG).  What the reference's
authors committed - three trained toy-dblp tables with their `t_loss` / `v_loss` (tests/golden/g14_n2v_dblp.npz) - pins the loss
NORMALISATION and the training schedule in distribution: tests/test_n2v.py requires this restatement, run with the committed
hyper-parameters, to end at the committed loss level.  Parity status: pinned in distribution, not bit-level.
"""
from __future__ import annotations

import numpy as np
import torch

EPS = 1e-15


def build_graph(skill_csr, member_csr, drop_teams=()):
    """Homogeneous CSR of the 'stm' graph (skill - team - member), node ids = [skills | members | teams] (offsets returned), undirected,
    with the member-team edges of `drop_teams` removed in both directions (gnn.py:84-96,107-116).  skill_csr / member_csr = (indptr, indices)."""
    s_ip, s_ix = (np.asarray(a) for a in skill_csr)
    m_ip, m_ix = (np.asarray(a) for a in member_csr)
    n_team = len(s_ip) - 1
    S = int(s_ix.max()) + 1 if len(s_ix) else 0
    return build_graph_sized(s_ip, s_ix, m_ip, m_ix, S, int(m_ix.max()) + 1 if len(m_ix) else 0, drop_teams)


def build_graph_sized(s_ip, s_ix, m_ip, m_ix, S, M, drop_teams=()):
    n_team = len(s_ip) - 1
    off = {"skill": 0, "member": S, "team": S + M}
    n = S + M + n_team
    team_of_s = np.repeat(np.arange(n_team), np.diff(s_ip))
    team_of_m = np.repeat(np.arange(n_team), np.diff(m_ip))
    keep = ~np.isin(team_of_m, np.asarray(list(drop_teams), dtype=np.int64))
    src = np.concatenate([s_ix + off["skill"], m_ix[keep] + off["member"]])
    dst = np.concatenate([team_of_s + off["team"], team_of_m[keep] + off["team"]])
    a, b = np.concatenate([src, dst]), np.concatenate([dst, src])      # ToUndirected
    order = np.lexsort((b, a))
    a, b = a[order], b[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, a + 1, 1)
    return np.cumsum(rowptr), b.astype(np.int32), off, n


def pos_sample(rowptr, col, batch, walk_length, context, walks_per_node, generator):
    """Node2Vec.pos_sample: uniform random walks of `walk_length` NODES from batch.repeat(walks_per_node), windows of `context` concatenated"""
    batch = torch.as_tensor(batch).repeat(walks_per_node)
    rw = torch.empty(len(batch), walk_length, dtype=torch.long)
    rw[:, 0] = batch
    rp, cl = torch.as_tensor(rowptr), torch.as_tensor(col).long()
    for s in range(1, walk_length):
        cur = rw[:, s - 1]
        deg = rp[cur + 1] - rp[cur]
        u = torch.rand(len(cur), generator=generator)
        pick = rp[cur] + (u * deg).long().clamp(max=(deg - 1).clamp(min=0))
        rw[:, s] = torch.where(deg > 0, cl[pick.clamp(max=len(cl) - 1)], cur)
    return windows(rw, context)


def neg_sample(num_nodes, batch, walk_length, context, walks_per_node, num_neg, generator):
    batch = torch.as_tensor(batch).repeat(walks_per_node * num_neg)
    rw = torch.randint(num_nodes, (len(batch), walk_length - 1), generator=generator)
    return windows(torch.cat([batch.view(-1, 1), rw], dim=-1), context)


def windows(rw, context):
    return torch.cat([rw[:, j:j + context] for j in range(rw.shape[1] + 1 - context)], dim=0)


def loss(weight, pos_rw, neg_rw):
    """Node2Vec.loss: -log(sigmoid(<start, rest>) + EPS).mean() over positive pairs - log(1 - sigmoid(.) + EPS).mean() over negative pairs"""
    def part(rw, positive):
        start, rest = rw[:, 0], rw[:, 1:].contiguous()
        h_start = weight[start].view(rw.size(0), 1, -1)
        h_rest = weight[rest.view(-1)].view(rw.size(0), -1, weight.shape[1])
        out = (h_start * h_rest).sum(dim=-1).view(-1)
        return -torch.log(torch.sigmoid(out) + EPS).mean() if positive else -torch.log(1 - torch.sigmoid(out) + EPS).mean()
    return part(torch.as_tensor(pos_rw), True) + part(torch.as_tensor(neg_rw), False)


def edge_bce(weight, src, dst):
    """v_loss of _train_rw before its second division (gnn.py:420-431)"""
    scores = (weight[torch.as_tensor(src)] * weight[torch.as_tensor(dst)]).sum(dim=-1)
    return torch.nn.functional.binary_cross_entropy_with_logits(scores, torch.ones_like(scores), reduction="mean")


def train(rowptr, col, num_nodes, d, b, epochs, lr, walk_length, context, walks_per_node, num_neg, seed=0):
    """_train_rw's loop without validation: returns (weight, per-epoch mean batch loss)"""
    g = torch.Generator().manual_seed(seed)
    emb = torch.nn.Embedding(num_nodes, d)
    with torch.no_grad(): emb.weight.normal_(generator=g)
    opt = torch.optim.Adam(emb.parameters(), lr=lr)
    hist = []
    for _ in range(epochs):
        perm = torch.randperm(num_nodes, generator=g)
        tot, nb = 0.0, 0
        for o in range(0, num_nodes, b):
            batch = perm[o:o + b]
            opt.zero_grad()
            l = loss(emb.weight, pos_sample(rowptr, col, batch, walk_length, context, walks_per_node, g),
                     neg_sample(num_nodes, batch, walk_length, context, walks_per_node, num_neg, g))
            l.backward(); opt.step()
            tot += float(l.detach()); nb += 1
        hist.append(tot / nb)
    return emb.weight.detach(), hist
