#!/usr/bin/env python3
"""`test()`'s inference (src/mdl/fnn.py:200-211 + src/pkgmgr.py:125-134) timed at config 2's expert count: ntf_forward_topk of 1 000 teams, Bnn at nmc = 10 and Fnn,
K = 100, with the per-family kernel times.  NTF_EVAL_PREFETCH=0: every MC pass produces its own operands in front of its forward kernel (round 5)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opentf_amd import libntf
from opentf_amd.synth import make_dataset, init_params
ds = make_dataset("dblp", d=128, seed=0, n_rows=20000)
for bayes, nmc in ((True, 10), (False, 1)):
    dims = [128, 128, ds["M"]]
    e = libntf.Engine(dims, bayesian=bayes, input_mode=libntf.INPUT_MEANPOOL, max_batch=1000, ns=5, nsd="uniform", seed=3, fuse_adam=1)
    e.set_skill_table(ds["table"]); e.set_skill_csr(ds["skill"]); e.set_member(ds["member"]); e.load_state_dict(init_params(dims, bayes, 0))
    rows = np.arange(1000)
    e.forward_topk(rows, nmc=nmc, K=100)
    t0 = time.perf_counter()
    for _ in range(5): e.forward_topk(rows, nmc=nmc, K=100)      # (timed without events; the call returns the top-K to the host, i.e. it is synchronous)
    dt = (time.perf_counter() - t0) / 5
    e.kernel_times(enable=True)
    for _ in range(5): e.forward_topk(rows, nmc=nmc, K=100)      # (a second loop with events around every kernel family, for the breakdown only)
    print("bayes", bayes, "nmc", nmc, "NTF_EVAL_PREFETCH", os.environ.get("NTF_EVAL_PREFETCH", "1"), "forward_topk(1000 teams, K=100):", round(dt * 1e3, 2), "ms",
          {k: round(v[0] / 5, 3) for k, v in e.kernel_times(enable=False).items() if v[1]}, flush=True)
    e.close()
