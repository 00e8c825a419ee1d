"""The part of the reference's "team2vec" embedding plugins that sits on the hot path: `get_dense_vecs`
(src/mdl/emb/t2v.py:18), i.e. Gnn's mean-pool of skill embeddings `(skill @ E) / skill.sum(1)`
(src/mdl/emb/gnn.py:484-486) as one CSR gather kernel on the MI355X, and D2v's per-team row lookup
(src/mdl/emb/d2v.py:110-116, a plain pass-through).  Training the embeddings themselves (gensim / PyG) is out of
scope: the table comes from wherever the reference produced it."""
from __future__ import annotations

import os

import numpy as np


class T2v:
    """Same constructor and attributes as src/mdl/emb/t2v.py:2-12."""

    def __init__(self, output, device, seed, cfg, model):
        self.data = None
        self.name = model
        self.model = None
        self.output = output
        self.cfg = cfg
        self.device = device
        self.seed = seed
        if not os.path.isdir(self.output):
            os.makedirs(self.output)

    def _prep(self, teamsvecs, splits, time_indexes=None): pass

    def learn(self, teamsvecs, splits, time_indexes=None): pass

    def get_dense_vecs(self, teamsvecs, vectype="skill"): pass


class TableT2v(T2v):
    """A T2v whose node-embedding table is given (e.g. loaded from a reference n2v/m2v checkpoint's
    `embedding.weight`).  `get_dense_vecs` = Gnn.get_dense_vecs on the GPU."""

    def set_table(self, table):
        self.model = np.ascontiguousarray(np.asarray(table, dtype=np.float32))
        return self

    def get_dense_vecs(self, teamsvecs, vectype="skill"):
        if self.model is None:
            raise RuntimeError("no embedding table: call set_table(E) first")
        if vectype not in teamsvecs:
            return self.model  # individual embeddings (gnn.py:486)
        return gather_meanpool(teamsvecs[vectype], self.model, self.device)


def gather_meanpool(sparse_rows, table, device="cuda:0"):
    """[N, d] f32 mean of each row's table entries; rows with no entry give nan (0/0), as the reference does."""
    from ... import libntf
    from ..fnn import parse_devices
    table = np.ascontiguousarray(np.asarray(table, dtype=np.float32))
    n = sparse_rows.shape[0]
    e = libntf.Engine([table.shape[1], 1, 1], input_mode=libntf.INPUT_MEANPOOL, max_batch=1, ns=0, nsd=None, device=parse_devices(device)[0])
    try:
        e.set_skill_table(table)
        e.set_skill_csr(sparse_rows)
        return e.gather_meanpool(n=n)
    finally:
        e.close()
