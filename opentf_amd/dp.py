"""Data parallelism over the GPUs of one node: one process per GPU, the global minibatch's rows split
contiguously over ranks, one gradient all-reduce (RCCL over xGMI through torch.distributed) per step.

The reference has no multi-GPU path (src/__config__.yaml:10 "TODO: multiple gpus"); semantics here are those of
the single-process step on the GLOBAL minibatch (src/mdl/fnn.py:122-140): every rank's backward is scaled by
1/global_B, so the SUM over ranks of the gradient buffers is the single-process gradient; the KL term is added
in shares B_rank/global_B.

The engine argument is duck-typed (`stage_order`, `step_staged`, `apply`, `grad_tensor`, `epoch_loss`), which is
what lets the world_size-2 gloo tests drive this logic on CPU with a stand-in engine.
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
    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._grad = engine.grad_tensor()  # flat view of the engine's gradient buffer (HBM; aliases, no copy)
        # NTF_DP_FORCE_ALLREDUCE=1: run the collective even at world_size 1 (exercises RCCL on the aliased buffer on a 1-GPU box)
        self.force_allreduce = dist.is_initialized() and os.environ.get("NTF_DP_FORCE_ALLREDUCE", "0") == "1"

    def _phase(self, order, global_B, train):
        """One `for batch in loader` phase (src/mdl/fnn.py:118) over `order`; returns the mean batch loss."""
        order = np.ascontiguousarray(np.asarray(order, dtype=np.int64))
        n = len(order)
        self.engine.stage_order(order)
        self.engine.epoch_loss()  # clear
        steps = 0
        for goff in range(0, n, global_B):
            gB = min(global_B, n - goff)
            lo, hi = shard_bounds(gB, self.world, self.rank)
            if self.world == 1 and not self.force_allreduce:
                # one GPU: backward and Adam in one call (lets the engine fuse the output layer's Adam into its dW kernel)
                self.engine.step_staged(goff, gB, global_offset=goff, global_B=gB, train=train, apply=train)
                steps += 1
                continue
            if hi > lo:
                self.engine.step_staged(goff + lo, hi - lo, global_offset=goff, global_B=gB, train=train, apply=False)
            elif train:
                self._grad.zero_()
            if train:
                if self.world > 1 or self.force_allreduce:
                    dist.all_reduce(self._grad, op=dist.ReduceOp.SUM, group=self.group)
                self.engine.apply()
            steps += 1
        s, _ = self.engine.epoch_loss()  # sum over steps of this rank's share of each batch loss
        t = torch.tensor([s], dtype=torch.float64, device=self._grad.device if self._grad.is_cuda else "cpu")
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item()) / max(steps, 1)

    def train_epoch(self, order, global_B):
        return self._phase(order, global_B, True)

    def eval_epoch(self, order, global_B):
        return self._phase(order, global_B, False)
