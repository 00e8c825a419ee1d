#!/usr/bin/env python3
"""Calibrate bench.py's CPU baseline (kind "port": oracle/ntf_oracle.py::reference_shaped_step) against THE REFERENCE ITSELF.

The port densifies a minibatch's labels with one vectorised `member_csr[rows].toarray()`; the reference pays `NtfDataset.__getitem__` once per team plus the
default collate (src/mdl/ntf.py:22-24, ~22 % of its step in SURVEY.md section 6's profile).  This script times both on the same host, same threads, at a size
both fit: Fnn, dense input d = 128, h = [128], M = 20 000 experts, B = 1000, uniform negatives ns = 5 (src/mdl/fnn.py:118-140), four train batches.

Build container only (imports /root/reference with the shims of SURVEY.md 8c; listed in .gpurunignore).  Prints the ratio that BASELINE.md section 3 and
bench.py's cpu_baseline.note quote.
"""
import os
import sys
import tempfile
import time

import numpy as np
import scipy.sparse
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import install_shims, Cfg  # noqa: E402


def main():
    threads = int(os.environ.get("CAL_THREADS", os.cpu_count() or 1))
    torch.set_num_threads(threads)
    install_shims()
    from mdl.fnn import Fnn
    from oracle import ntf_oracle as O
    from collections import OrderedDict
    N_TRAIN, M, D, B = 4000, 20_000, 128, 1000
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N_TRAIN + 8, D)).astype(np.float32)
    rows = np.repeat(np.arange(N_TRAIN + 8), 3); cols = rng.integers(0, M, len(rows))
    member = scipy.sparse.csr_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(N_TRAIN + 8, M)).tolil()
    tv = {"skill": X, "member": member}
    splits = {"test": np.arange(0), "folds": {0: {"train": np.arange(N_TRAIN), "valid": np.arange(N_TRAIN, N_TRAIN + 8)}}}
    base = dict(b=B, ns=5, lr=0.001, es=100, h=[128], spe=0, l="bce", tpw=10, tnw=1, nsd="uniform")

    def run_ref(epochs):
        m = Fnn(tempfile.mkdtemp(prefix="cal_"), "cpu", 0, Cfg({**base, "e": epochs}))
        t0 = time.perf_counter(); m.learn(tv, splits, None); return time.perf_counter() - t0
    run_ref(1)                                          # warm-up (thread pools, allocator)
    t1, t3 = run_ref(1), run_ref(3)
    per_epoch = (t3 - t1) / 2                           # 4 train batches of 1000 + one 8-row validation batch
    ref_rate = N_TRAIN / per_epoch

    sd = OrderedDict((k, v.clone()) for k, v in O.fnn_init(D, [128], M).items())
    opt = O.Adam(sd, 1e-3)
    member_csr = member.tocsr()
    cfg = {"ns": 5, "nsd": "uniform", "tpw": 10.0, "tnw": 1.0}
    # dense input: the "table" is the input matrix itself, one "skill" per team -> the mean pool of one row = that row
    ip, ix = np.arange(N_TRAIN + 9, dtype=np.int64), np.arange(N_TRAIN + 8, dtype=np.int32)
    O.reference_shaped_step(sd, opt, X, ip, ix, np.arange(64), member_csr, cfg)
    t0 = time.perf_counter()
    for b in range(N_TRAIN // B): O.reference_shaped_step(sd, opt, X, ip, ix, np.arange(b * B, (b + 1) * B), member_csr, cfg)
    port_rate = N_TRAIN / (time.perf_counter() - t0)
    print(f"threads {threads}  torch {torch.__version__}  M {M}  B {B}")
    print(f"reference Fnn.learn (imported, shims): {ref_rate:9.1f} teams/s   ({per_epoch / 4 * 1e3:.0f} ms per train batch)")
    print(f"port reference_shaped_step           : {port_rate:9.1f} teams/s")
    print(f"port / reference = {port_rate / ref_rate:.2f}  (the port skips the per-team __getitem__ + collate of src/mdl/ntf.py:22-24)")


if __name__ == "__main__":
    main()
