"""Expert sharding of the output layer over the GPUs of one node: one process per GPU, every process steps the
WHOLE minibatch, each on the experts [lo, hi) of the output layer it owns.

Why: the output layer holds > 99 % of the parameters (60.3 M of them at BASELINE config 2).  Data parallelism (dp.py) moves their
gradients over xGMI every step (241 MB reduce-scatter + all-gather) and repeats the Flipout operand producer, the KL term and - unless
sharded - Adam on every GPU.  Split along the expert axis, nothing of the output layer is exchanged or repeated: each GPU draws the
noise, runs the forward / loss / backward kernels and Adam for its own experts only, and the one exchange of a step is the sum over
GPUs of d(hidden) - [B, h[-1]] floats, 0.5 MB at B = 1000 - issued before the output layer's backward, which hides it, and waited for
before the (replicated, tiny) hidden layers' backward.  The reference has no multi-GPU path (src/__config__.yaml:10 "TODO: multiple gpus"); the semantics are those of its
single-process step on the same minibatch (src/mdl/fnn.py:122-140): labels and sampled negatives keep global expert ids, and the
device generators are keyed by global ids, so G shards draw exactly what one engine holding the whole layer draws.

Shard boundaries are multiples of 256 experts (the dW kernel's tile; also keeps the 32-expert sign words whole).

Stream contract as in dp.py: engine kernels and the all-reduce are ordered through ONE stream - the caller runs under
`torch.cuda.stream(s)` with the engine created on `s`.

The engine argument is duck-typed (`stage_order`, `step_staged`, `step_staged_ep`, `dh_tensor`, `epoch_loss`, `dims`), which lets the
world_size-2 gloo test drive this logic on CPU with a stand-in engine.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from .dp import CollectiveTrace

TILE = 256  # experts: fused_dw_tile() of the engine


def expert_shards(M: int, world: int):
    """[(lo, hi)] per rank: contiguous, boundaries at multiples of 256 experts, tile counts differ by at most one.
    Raises when the layer has fewer 256-expert tiles than ranks (use data parallelism for such a model)."""
    tiles = -(-int(M) // TILE)
    if tiles < world:
        raise ValueError(f"{M} experts are {tiles} tiles of {TILE}: fewer than {world} shards")
    base, extra = divmod(tiles, world)
    out, t = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((min(t * TILE, M), min((t + n) * TILE, M)))
        t += n
    return out


def can_shard(dims, world: int) -> bool:
    """the fused output-layer path (h[-1] in {32, 64, 128}) and at least one 256-expert tile per rank"""
    return len(dims) >= 2 and dims[-2] in (32, 64, 128) and -(-int(dims[-1]) // TILE) >= world


class ExpertParallel:
    def __init__(self, engine, group=None, two_phase=False):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._dh = engine.dh_tensor()          # flat alias of the engine's d(hidden) buffer (None: no hidden layer, nothing to exchange)
        self._H = engine.dims[-2]
        # NTF_EP_FORCE_EXCHANGE=1: run the two-phase step and the all-reduce even at world_size 1 (exercises RCCL on a 1-GPU box)
        # two_phase: the two-phase step without any process group (bench.py --ep-emulate: one rank's compute on a 1-GPU box)
        self.force = bool(two_phase) or (dist.is_initialized() and os.environ.get("NTF_EP_FORCE_EXCHANGE", "0") == "1")
        self.trace = CollectiveTrace(f"ExpertParallel rank {self.rank}/{self.world}", stream_ordered=dist.is_initialized() and dist.get_backend(group) == "nccl")   # bounded waits that name the collective a hang sits behind (dp.py)
        self._step_no = 0
        if self._dh is not None and self._dh.is_cuda and (self.world > 1 or self.force) and hasattr(engine, "stream_handle"):
            assert engine.stream_handle is not None and torch.cuda.current_stream().cuda_stream == engine.stream_handle, \
                "ExpertParallel must run under torch.cuda.stream(s) with the engine created on s: kernels and the all-reduce are ordered through that one stream"

    def _phase(self, order, global_B, train):
        """One `for batch in loader` phase (src/mdl/fnn.py:118) over `order`; returns the mean batch loss."""
        order = np.ascontiguousarray(np.asarray(order, dtype=np.int64))
        n = len(order)
        e = self.engine
        e.stage_order(order)
        e.epoch_loss()  # clear
        steps = 0
        exchange = self.world > 1 or self.force
        for off in range(0, n, global_B):
            B = min(global_B, n - off)
            steps += 1
            if train and exchange:
                self._step_no += 1
                e.step_staged_ep(off, B, 1)        # forward, loss, fix-up: this shard's partial d(hidden)
                work = None
                if self._dh is not None and dist.is_initialized():   # the one exchange; RCCL moves it on its own stream ...
                    work = self.trace.add(f"step {self._step_no} all_reduce d(hidden)[{B} x {self._H}]",
                                          dist.all_reduce(self._dh[: B * self._H], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                e.step_staged_ep(off, B, 2)        # ... while this shard's output-layer backward (+ its Adam) runs
                if work is not None:
                    self.trace.wait(work)          # stream-ordered under RCCL: the engine's stream waits, the host does not (gloo: bounded host wait)
                e.step_staged_ep(off, B, 3)        # hidden layers' backward + Adam, identical on every rank
            else:
                e.step_staged(off, B, train=train, apply=train)   # evaluation needs no exchange: the loss shares are summed below
        on_gpu = self._dh is not None and self._dh.is_cuda
        if exchange and on_gpu:
            self.trace.sync("the phase's kernels and all-reduces")      # bounded: everything below would wait for ever behind a hung collective
        if train and self.world > 1:
            self._resync_replicas()
        s, _ = e.epoch_loss()      # sum over steps of this shard's share of each batch loss
        dev = self._dh.device if on_gpu else ("cuda" if (dist.is_initialized() and dist.get_backend(self.group) == "nccl") else "cpu")
        t = torch.tensor([s], dtype=torch.float64, device=dev)
        if self.world > 1:
            self.trace.wait(self.trace.add("phase loss all_reduce", dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            if t.is_cuda: self.trace.sync("the loss all_reduce")
        return float(t.item()) / max(steps, 1)

    def _resync_replicas(self):
        """The hidden layers are replicated, not exchanged: every rank updates them from the same summed d(hidden) with deterministic kernels, so
        the replicas stay bit-identical - except behind a multi-hot first layer, whose weight gradient is a scatter-add by float atomics
        (summation order varies from run to run).  Once per epoch rank 0's replicated parameters are broadcast, which bounds any such
        drift at rounding level and makes the checkpoint (rank 0's hidden layers + everyone's experts) the model every rank holds."""
        e = self.engine
        if e.L < 2 or not hasattr(e, "param_tensor"):
            return
        from .libntf import P_WEIGHT
        end, _ = e.param_segment(e.L - 1, P_WEIGHT)     # segments are laid out layer by layer: [0, end) = the hidden layers
        if end > 0:
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            self.trace.wait(self.trace.add("epoch-end broadcast of the replicated parameters", dist.broadcast(e.param_tensor()[:end], src=src, group=self.group, async_op=True)))
            if hasattr(e, "moment_tensors"):     # ... and Adam's moments of those layers: replicas that keep their own would drift apart again at once
                for t in e.moment_tensors():
                    self.trace.wait(self.trace.add("epoch-end broadcast of the replicated Adam moments", dist.broadcast(t[:end], src=src, group=self.group, async_op=True)))
            if hasattr(e, "params_touched"): e.params_touched()

    def train_epoch(self, order, global_B):
        return self._phase(order, global_B, True)

    def eval_epoch(self, order, global_B):
        return self._phase(order, global_B, False)

    def state_dict(self):
        """the WHOLE model's state_dict (reference layout) on every rank: the output layer's rows are gathered from the shards"""
        return gather_state_dict(self.engine.state_dict(), self.engine.L - 1, self.group)


def gather_state_dict(local_sd, last_layer: int, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_sd
    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    out = OrderedDict()
    prefix = f"layers.{last_layer}."
    for k, v in local_sd.items():
        if not k.startswith(prefix):
            out[k] = v
            continue
        a = torch.from_numpy(np.ascontiguousarray(v))
        rows = torch.tensor([a.shape[0]], dtype=torch.int64)
        if on_gpu:
            a, rows = a.cuda(), rows.cuda()
        counts = [torch.zeros_like(rows) for _ in range(world)]
        dist.all_gather(counts, rows, group=group)
        counts = [int(c.item()) for c in counts]
        pad = torch.zeros((max(counts),) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
        pad[: a.shape[0]] = a
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out[k] = torch.cat([p[:c] for p, c in zip(parts, counts)]).cpu().numpy()
    return out
