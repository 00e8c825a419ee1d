"""node2vec trainer (ntf_n2v_*) at the bench dataset's size: the skill - team - member graph of dblp mt10.ts2 shapes, batches of 1000 start nodes with the reference's settings (walk length 5, context 5, 10 walks per node, 5 negatives)"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opentf_amd import libntf
from opentf_amd.synth import make_dataset
from opentf_amd.mdl.emb.gnn import stm_graph
import scipy.sparse as sp
ds = make_dataset("dblp", d=128, seed=0)
N, S, M = ds["N"], ds["S"], ds["M"]
skill = sp.csr_matrix((np.ones(len(ds["skill"][1]), np.uint8), ds["skill"][1], ds["skill"][0]), shape=(N, S))
member = sp.csr_matrix((np.ones(len(ds["member"][1]), np.uint8), ds["member"][1], ds["member"][0]), shape=(N, M))
t0 = time.perf_counter(); rowptr, col, off, n = stm_graph(skill, member); print("graph", n, "nodes", len(col), "edges", round(time.perf_counter() - t0, 2), "s")
w = (np.random.default_rng(0).standard_normal((n, 128)) ).astype(np.float32)
net = libntf.Node2Vec(rowptr, col, w, seed=0)
b = 1000
order = np.random.default_rng(1).permutation(n)
net.train_batch(order[:b], 5, 5, 10, 5, 1e-3)
t0 = time.perf_counter(); k = 50
for i in range(k): l = net.train_batch(order[(i + 1) * b:(i + 2) * b], 5, 5, 10, 5, 1e-3, want_loss=(i == k - 1))
dt = time.perf_counter() - t0
print("train_batch of %d nodes: %.3f ms; an epoch of %d batches: %.1f s; loss %.4f" % (b, dt / k * 1e3, (n + b - 1) // b, dt / k * ((n + b - 1) // b), l))
