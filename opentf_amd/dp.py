"""Data parallelism over the GPUs of one node: one process per GPU, the global minibatch's rows split
contiguously over ranks, gradient all-reduce (RCCL over xGMI through torch.distributed) per step.

The reference has no multi-GPU path (src/__config__.yaml:10 "TODO: multiple gpus"); semantics here are those of
the single-process step on the GLOBAL minibatch (src/mdl/fnn.py:122-140): every rank's backward is scaled by
1/global_B, so the SUM over ranks of the gradient buffers is the single-process gradient; the KL term is added
in shares B_rank/global_B.

Overlap: the output layer holds >99 % of the parameters.  Its weight-gradient kernel is launched in expert chunks
(`dw_chunk`), and the all-reduce of chunk k is issued asynchronously as soon as that chunk's kernel is queued, so
that RCCL moves chunk k over xGMI while the GPU computes chunk k+1; the small remainder (hidden layers, biases)
goes last.  All collectives of a step are waited for before Adam.

The engine argument is duck-typed (`stage_order`, `step_staged`, `apply`, `grad_tensor`, `epoch_loss`, and
optionally `step_staged_deferred` / `dw_chunks` / `dw_chunk_range` / `dw_chunk` / `rest_ranges`), which is what
lets the world_size-2 gloo tests drive this logic on CPU with a stand-in engine.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(global_B: int, world: int, rank: int):
    """Contiguous split of a global minibatch; sizes differ by at most one (first ranks get the extra row)."""
    base, extra = divmod(global_B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class DataParallel:
    def __init__(self, engine, group=None, overlap=True):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._grad = engine.grad_tensor()  # flat view of the engine's gradient buffer (HBM; aliases, no copy)
        # NTF_DP_FORCE_ALLREDUCE=1: run the collectives even at world_size 1 (exercises RCCL on the aliased buffer on a 1-GPU box)
        self.force_allreduce = dist.is_initialized() and os.environ.get("NTF_DP_FORCE_ALLREDUCE", "0") == "1"
        self.n_chunks = engine.dw_chunks() if (overlap and hasattr(engine, "dw_chunks")) else 0
        if self.n_chunks:
            self._chunk_ranges = [engine.dw_chunk_range(k) for k in range(self.n_chunks)]  # identical on every rank
            self._rest = engine.rest_ranges()

    def _reduce(self, lo, hi):
        return dist.all_reduce(self._grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _skip(self, zero_grad=True):
        """this rank's shard of the global minibatch is empty (a last batch smaller than the world size): contribute a zero gradient and
        advance the engine's step counter all the same - it keys the Flipout eps, which every rank must draw identically"""
        if zero_grad:
            self._grad.zero_()
        if hasattr(self.engine, "skip_step"):
            self.engine.skip_step()

    def _train_step(self, goff, gB, lo, hi):
        """backward of this rank's shard + gradient all-reduce + Adam for one global minibatch"""
        e, have_rows = self.engine, hi > lo
        works = []
        if self.n_chunks:
            if have_rows:
                e.step_staged_deferred(goff + lo, hi - lo, goff, gB)
            else:
                self._skip()
            for k, (ow, orr, cnt) in enumerate(self._chunk_ranges):
                if have_rows:
                    e.dw_chunk(k)                     # queue chunk k's kernel ...
                works.append(self._reduce(ow, ow + cnt))  # ... and let RCCL take its gradients as soon as it finishes
                if orr >= 0:
                    works.append(self._reduce(orr, orr + cnt))
            for rlo, rhi in self._rest:
                works.append(self._reduce(rlo, rhi))
        else:
            if have_rows:
                e.step_staged(goff + lo, hi - lo, global_offset=goff, global_B=gB, train=True, apply=False)
            else:
                self._skip()
            works.append(self._reduce(0, self._grad.numel()))
        for w in works:
            w.wait()
        e.apply()

    def _phase(self, order, global_B, train):
        """One `for batch in loader` phase (src/mdl/fnn.py:118) over `order`; returns the mean batch loss."""
        order = np.ascontiguousarray(np.asarray(order, dtype=np.int64))
        n = len(order)
        self.engine.stage_order(order)
        self.engine.epoch_loss()  # clear
        steps = 0
        collective = self.world > 1 or self.force_allreduce
        for goff in range(0, n, global_B):
            gB = min(global_B, n - goff)
            steps += 1
            if not collective:
                # one GPU: backward and Adam in one call (lets the engine overlap / fuse the output layer's Adam with its dW kernel)
                self.engine.step_staged(goff, gB, global_offset=goff, global_B=gB, train=train, apply=train)
                continue
            lo, hi = shard_bounds(gB, self.world, self.rank)
            if train:
                self._train_step(goff, gB, lo, hi)
            elif hi > lo:
                self.engine.step_staged(goff + lo, hi - lo, global_offset=goff, global_B=gB, train=False, apply=False)
            else:
                self._skip(zero_grad=False)
        s, _ = self.engine.epoch_loss()  # sum over steps of this rank's share of each batch loss
        t = torch.tensor([s], dtype=torch.float64, device=self._grad.device if self._grad.is_cuda else "cpu")
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item()) / max(steps, 1)

    def train_epoch(self, order, global_B):
        return self._phase(order, global_B, True)

    def eval_epoch(self, order, global_B):
        return self._phase(order, global_B, False)
