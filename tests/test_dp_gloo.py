"""world_size-2 CPU (gloo) test of the data-parallel host logic (opentf_amd/dp.py).  The HIP engine cannot run here,
so a stand-in with the engine's staged-step interface computes each rank's shard with the ORACLE; what is under test
is dp.py itself: the contiguous sharding of every global minibatch, the 1/global_B scaling contract, the gradient
all-reduce, the loss aggregation — two ranks must reproduce the single-process oracle trajectory."""
import os
import socket
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ntf_oracle as O


class OracleEngine:
    """Same surface as libntf.Engine's staged API, CPU math from the oracle (test stand-in only)."""

    def __init__(self, sd, X, y, tpw, tnw, lr, bayesian_noise=None):
        self.sd = OrderedDict((k, v.clone()) for k, v in sd.items())
        self.X, self.y, self.tpw, self.tnw = X, y, tpw, tnw
        self.opt = O.Adam(self.sd, lr)
        self.keys = list(self.sd)
        self.flat = torch.zeros(sum(v.numel() for v in self.sd.values()))
        self.acc, self.steps, self.order = 0.0, 0, None
        self.noise = bayesian_noise  # dict global_offset -> per-layer noise for the GLOBAL batch (same eps on all ranks)

    def grad_tensor(self): return self.flat

    def stage_order(self, order): self.order = np.asarray(order)

    def epoch_loss(self):
        s, k = self.acc, self.steps
        self.acc, self.steps = 0.0, 0
        return s, k

    def step_staged(self, offset, B, global_offset=None, global_B=None, train=True, apply=True, want_loss=False):
        rows = self.order[offset:offset + B]
        X, y = self.X[rows], self.y[rows]
        noise = None
        if self.noise is not None:
            full = self.noise[int(global_offset)]
            lo = offset - global_offset
            noise = [{"eps_w": n["eps_w"], "eps_b": n["eps_b"], "s_in": n["s_in"][lo:lo + B], "s_out": n["s_out"][lo:lo + B]} for n in full]
        leaf = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in self.sd.items())
        # the engine's contract: sum over the shard's rows / global_B  (+ KL * (B/global_B) / global_B)
        loss = O.bxe(O.model_forward(leaf, X, noise), y, None, self.tpw, self.tnw).sum() / global_B
        if O.is_bayesian(leaf):
            loss = loss + O.get_kl_loss(leaf) * (B / global_B) / global_B
        if train:
            loss.backward()
            self.flat.copy_(torch.cat([leaf[k].grad.reshape(-1) for k in self.keys]))
            if apply: self.apply()
        self.acc += float(loss.item()); self.steps += 1

    # ---- the pipelined (chunked output-layer dW) surface of the engine
    N_CHUNKS = 3

    def _last_weight_keys(self):
        last = O.n_layers(self.sd) - 1
        return [k for k in self.keys if k.startswith(f"layers.{last}.") and k.endswith("weight")]  # weight | mu_weight, rho_weight

    def _offset(self, key):
        o = 0
        for k in self.keys:
            if k == key: return o
            o += self.sd[k].numel()

    def dw_chunks(self): return self.N_CHUNKS

    def dw_chunk_range(self, k):
        keys = self._last_weight_keys()
        rows, cols = self.sd[keys[0]].shape
        r0, r1 = rows * k // self.N_CHUNKS, rows * (k + 1) // self.N_CHUNKS
        ow = self._offset(keys[0]) + r0 * cols
        orr = self._offset(keys[1]) + r0 * cols if len(keys) > 1 else -1
        return ow, orr, (r1 - r0) * cols

    def rest_ranges(self):
        cuts = sorted((self._offset(k), self._offset(k) + self.sd[k].numel()) for k in self._last_weight_keys())
        out, pos = [], 0
        for lo, hi in cuts:
            if lo > pos: out.append((pos, lo))
            pos = hi
        if pos < self.flat.numel(): out.append((pos, self.flat.numel()))
        return out

    def step_staged_deferred(self, offset, B, global_offset, global_B):
        self.step_staged(offset, B, global_offset, global_B, train=True, apply=False)
        self._full = self.flat.clone()
        for k in range(self.N_CHUNKS):   # the deferred kernel has not produced these yet: poison them
            ow, orr, cnt = self.dw_chunk_range(k)
            self.flat[ow:ow + cnt] = float("nan")
            if orr >= 0: self.flat[orr:orr + cnt] = float("nan")

    def dw_chunk(self, k):
        ow, orr, cnt = self.dw_chunk_range(k)
        self.flat[ow:ow + cnt] = self._full[ow:ow + cnt]
        if orr >= 0: self.flat[orr:orr + cnt] = self._full[orr:orr + cnt]

    def apply(self):
        grads, o = OrderedDict(), 0
        for k in self.keys:
            n = self.sd[k].numel(); grads[k] = self.flat[o:o + n].view_as(self.sd[k]).clone(); o += n
        self.opt.step(self.sd, grads)


def _case(bayesian):
    torch.manual_seed(0)
    D, H, M, N = 12, [16], 40, 37
    sd = O.bnn_init(D, H, M) if bayesian else O.fnn_init(D, H, M)
    X = torch.randn(N, D)
    y = (torch.rand(N, M) < 0.1).float()
    order = torch.randperm(N).numpy()
    gB = 10  # 37 rows -> batches of 10,10,10,7: the last one splits 4+3
    noise = None
    if bayesian:
        noise = {off: O.draw_flipout_noise(sd, min(gB, N - off)) for off in range(0, N, gB)}
    return sd, X, y, order, gB, noise


def _single_process(bayesian):
    sd, X, y, order, gB, noise = _case(bayesian)
    sd = OrderedDict((k, v.clone()) for k, v in sd.items())
    opt = O.Adam(sd, 1e-2)
    losses = []
    for off in range(0, len(order), gB):
        rows = order[off:off + gB]
        loss, _ = O.train_step(sd, opt, X[rows], y[rows], None, 10.0, 1.0, noise[off] if noise else None)
        losses.append(loss)
    return sd, float(np.mean(losses))


def _worker(rank, world, port, bayesian, overlap, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opentf_amd.dp import DataParallel
    sd, X, y, order, gB, noise = _case(bayesian)
    eng = OracleEngine(sd, X, y, 10.0, 1.0, 1e-2, noise)
    dp = DataParallel(eng, overlap=overlap)
    assert dp.n_chunks == (OracleEngine.N_CHUNKS if overlap else 0)
    mean_loss = dp.train_epoch(order, gB)
    eval_loss = dp.eval_epoch(order, gB)
    if rank == 0:
        out.put((mean_loss, eval_loss, {k: v.numpy() for k, v in eng.sd.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("bayesian,overlap", [(False, False), (True, False), (False, True), (True, True)])
def test_two_rank_data_parallel_equals_single_process(bayesian, overlap):
    ref_sd, ref_loss = _single_process(bayesian)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bayesian, overlap, out)) for r in range(2)]
    for p in procs: p.start()
    mean_loss, eval_loss, sd = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(mean_loss - ref_loss) <= 1e-5 * abs(ref_loss)
    for k in ref_sd:
        np.testing.assert_allclose(sd[k], ref_sd[k].numpy(), rtol=2e-4, atol=1e-6)
    # eval phase: mean over batches of the global-batch loss under the final weights
    sd0, X, y, order, gB, noise = _case(bayesian)
    ev = [float(O.batch_loss(ref_sd, X[order[o:o + gB]], y[order[o:o + gB]], None, 10.0, 1.0, noise[o] if noise else None)) for o in range(0, len(order), gB)]
    assert abs(eval_loss - np.mean(ev)) <= 1e-4 * abs(np.mean(ev))
