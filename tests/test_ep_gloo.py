"""world_size-2/3 CPU (gloo) test of the expert-sharding host logic (opentf_amd/ep.py).  The HIP engine cannot run here, so a stand-in with the
engine's two-phase step interface computes one shard's part with the ORACLE; what is under test is ep.py itself: where the d(hidden) all-reduce
sits between the phases, the loss aggregation, the state_dict gather - the ranks together must reproduce the single-process oracle trajectory."""
import os
import socket
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ntf_oracle as O
from test_dp_gloo import _case, _single_process


class OracleShard:
    """One expert shard [lo, hi) with libntf.Engine's expert-sharded surface; CPU math from the oracle (test stand-in only)."""

    def __init__(self, sd, X, y, tpw, tnw, lr, lo, hi, world, noise=None):
        self.L = O.n_layers(sd)
        self.last = f"layers.{self.L - 1}."
        self.M = y.shape[1]
        self.lo, self.hi, self.world = lo, hi, world
        self.p = OrderedDict((k, (v[lo:hi] if k.startswith(self.last) else v).clone()) for k, v in sd.items())   # hidden replicated, output rows [lo, hi)
        wkey = self.last + ("mu_weight" if O.is_bayesian(sd) else "weight")
        self.dims = [sd[wkey].shape[1], hi - lo]       # what ExpertParallel reads: dims[-2] = width of d(hidden)
        self.X, self.y, self.tpw, self.tnw, self.lr = X, y, tpw, tnw, lr
        self.opt = O.Adam(self.p, lr)
        self.noise = noise
        self.acc, self.steps, self.order = 0.0, 0, None
        H = self.dims[-2]
        self.dh = torch.zeros(64 * H)      # max_batch * h[-1]
        self.calls = []

    def dh_tensor(self): return self.dh if self.L > 1 else None

    def stage_order(self, order): self.order = np.asarray(order)

    def epoch_loss(self):
        s, k = self.acc, self.steps
        self.acc, self.steps = 0.0, 0
        return s, k

    def _noise(self, off, B):
        if self.noise is None: return None, None
        full = self.noise[int(off)]
        last = dict(full[-1]); last["eps_w"] = last["eps_w"][self.lo:self.hi]; last["eps_b"] = last["eps_b"][self.lo:self.hi]; last["s_out"] = last["s_out"][:, self.lo:self.hi]
        return full[:-1], [last]

    def _forward(self, leaf, off, B):
        rows = self.order[off:off + B]
        X, y = self.X[rows], self.y[rows][:, self.lo:self.hi]
        n_hid, n_out = self._noise(off, B)
        hid = OrderedDict((k, v) for k, v in leaf.items() if not k.startswith(self.last))
        out = OrderedDict((k.replace(self.last, "layers.0."), v) for k, v in leaf.items() if k.startswith(self.last))
        h = O.model_forward(hid, X, n_hid) if hid else X
        return h, out, y, n_out

    def _loss(self, h, out, y, n_out, B):
        loss = O.bxe(O.model_forward(out, h, n_out), y, None, self.tpw, self.tnw).sum() / B
        if O.is_bayesian(out):   # the output layer's KL is a mean over the WHOLE layer's elements: this shard's share of the sums
            kw = O.kl_div(out["layers.0.mu_weight"], O.softplus_rho(out["layers.0.rho_weight"])) * (self.hi - self.lo) / self.M
            kb = O.kl_div(out["layers.0.mu_bias"], O.softplus_rho(out["layers.0.rho_bias"])) * (self.hi - self.lo) / self.M
            loss = loss + (kw + kb) / B
        return loss

    def step_staged(self, offset, B, global_offset=None, global_B=None, train=True, apply=True, want_loss=False):
        assert not train, "a shard's train step needs the exchange"
        with torch.no_grad():
            h, out, y, n_out = self._forward(self.p, offset, B)
            loss = self._loss(h, out, y, n_out, B)
            hid = OrderedDict((k, v) for k, v in self.p.items() if not k.startswith(self.last))
            if hid and O.is_bayesian(hid): loss = loss + O.get_kl_loss(hid) / B / self.world
        self.acc += float(loss); self.steps += 1

    def step_staged_ep(self, offset, B, phase):
        self.calls.append(phase)
        if phase == 1:
            self.leaf = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in self.p.items())
            h, out, y, n_out = self._forward(self.leaf, offset, B)
            self.h = h
            h_leaf = h.detach().clone().requires_grad_(True) if self.L > 1 else h
            loss = self._loss(h_leaf, out, y, n_out, B)
            loss.backward()
            hid = OrderedDict((k, v) for k, v in self.leaf.items() if not k.startswith(self.last))
            if hid and O.is_bayesian(hid): loss = loss.detach() + O.get_kl_loss(hid).detach() / B / self.world   # counted once over the shards
            self.acc += float(loss); self.steps += 1
            self.B = B
            if self.L > 1: self.dh[: B * h.shape[1]] = h_leaf.grad.reshape(-1)        # this shard's PARTIAL d loss / d hidden
        elif phase == 2:
            pass                                                                          # the output layer's gradients are ready (autograd made them in phase 1)
        else:
            B = self.B
            if self.L > 1:
                g = self.dh[: B * self.h.shape[1]].view(B, -1).clone()                # the SUM over the shards, put there by the all-reduce
                hid = OrderedDict((k, v) for k, v in self.leaf.items() if not k.startswith(self.last))
                tail = (O.get_kl_loss(hid) / B) if O.is_bayesian(hid) else None         # the replicated layers' KL gradient is not shared out
                self.h.backward(g, retain_graph=tail is not None)
                if tail is not None: tail.backward()
            grads = OrderedDict((k, v.grad) for k, v in self.leaf.items())
            self.opt.step(self.p, grads)

    def state_dict(self):
        return OrderedDict((k, v.detach().numpy().copy()) for k, v in self.p.items())


def _worker(rank, world, port, bayesian, out, hidden):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opentf_amd.ep import ExpertParallel
    sd, X, y, order, gB, noise = _case(bayesian)
    M = y.shape[1]
    cuts = [M * r // world for r in range(world + 1)]        # the stand-in has no 256-expert tiles: any contiguous split
    eng = OracleShard(sd, X, y, 10.0, 1.0, 1e-2, cuts[rank], cuts[rank + 1], world, noise)
    ep = ExpertParallel(eng)
    mean_loss = ep.train_epoch(order, gB)
    assert eng.calls == [1, 2, 3] * ((len(order) + gB - 1) // gB)
    eval_loss = ep.eval_epoch(order, gB)
    full = ep.state_dict()
    for k, v in full.items():      # replicated layers identical on every rank, the gathered output layer complete
        t = torch.from_numpy(np.ascontiguousarray(v)).clone(); ref = t.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, t), k
    if rank == 0:
        out.put((mean_loss, eval_loss, full))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,bayesian", [(2, False), (2, True), (3, True)])
def test_expert_parallel_equals_single_process(world, bayesian):
    ref_sd, ref_loss = _single_process(bayesian)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bayesian, out, True)) for r in range(world)]
    for p in procs: p.start()
    mean_loss, eval_loss, sd = out.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(mean_loss - ref_loss) <= 1e-5 * abs(ref_loss)
    assert set(sd) == set(ref_sd)
    for k in ref_sd:
        assert sd[k].shape == tuple(ref_sd[k].shape), k
        np.testing.assert_allclose(sd[k], ref_sd[k].numpy(), rtol=2e-4, atol=1e-6, err_msg=k)
    sd0, X, y, order, gB, noise = _case(bayesian)
    ev = [float(O.batch_loss(ref_sd, X[order[o:o + gB]], y[order[o:o + gB]], None, 10.0, 1.0, noise[o] if noise else None)) for o in range(0, len(order), gB)]
    assert abs(eval_loss - np.mean(ev)) <= 1e-4 * abs(np.mean(ev))


def test_expert_shards_partition():
    from opentf_amd.ep import expert_shards, can_shard
    for M, w in [(233_629, 8), (233_629, 3), (5_022_955, 8), (39_204, 4), (1024, 4), (257, 2), (3000, 5)]:
        sh = expert_shards(M, w)
        assert len(sh) == w and sh[0][0] == 0 and sh[-1][1] == M
        assert all(a[1] == b[0] for a, b in zip(sh, sh[1:])) and all(lo % 256 == 0 and hi > lo for lo, hi in sh)
        tiles = [-(-(hi - lo) // 256) for lo, hi in sh]
        assert max(tiles) - min(tiles) <= 1
    with pytest.raises(ValueError):
        expert_shards(700, 4)          # 3 tiles of 256 for 4 ranks
    assert can_shard([128, 128, 233_629], 8) and not can_shard([128, 100, 233_629], 8) and not can_shard([18, 32, 112], 2)
