"""Data parallelism over the GPUs of one node: one process per GPU, the global minibatch's rows split contiguously over ranks,
gradients exchanged over xGMI (RCCL through torch.distributed) once per step.

The reference has no multi-GPU path (src/__config__.yaml:10 "TODO: multiple gpus"); semantics here are those of the
single-process step on the GLOBAL minibatch (src/mdl/fnn.py:122-140): every rank's backward is scaled by 1/global_B, so the SUM
over ranks of the gradient buffers is the single-process gradient; the KL term is added in shares B_rank/global_B.

Exchange, `shard_optimizer=True` (default when the engine offers `apply_ranges`):
    reduce-scatter(gradients)  ->  Adam on the 1/G shard this rank owns  ->  all-gather(parameters)
Every gradient range is cut into G equal parts (a remainder of < G*4 floats is all-reduced and updated by every rank).  Against
all-reduce + replicated Adam this moves the same bytes over xGMI (reduce-scatter + all-gather = one ring all-reduce) but divides the
optimiser's HBM traffic (28 B / parameter: 1.69 GB per step at BASELINE config 2) by G; Adam's moments of the other shards are
never touched on this rank.  `shard_optimizer=False` keeps all-reduce + replicated Adam.

Overlap: the output layer holds >99 % of the parameters.  Its weight-gradient kernel is launched in expert chunks (`dw_chunk`), and
the collective of chunk k is issued asynchronously as soon as that chunk's kernel is queued, so that RCCL moves chunk k over xGMI
while the GPU computes chunk k+1; the small remainder (hidden layers, biases: ~1 MB) is all-reduced last and updated on every rank.
All collectives of a step are waited for before Adam; the parameter all-gathers are waited for before the next step's first kernel.

Stream contract: engine kernels and collectives are ordered through ONE stream - the caller runs under `torch.cuda.stream(s)` with
the engine created on `s` (opentf_amd/mdl/fnn.py does); this is asserted at construction when the engine exposes its stream.

The engine argument is duck-typed (`stage_order`, `step_staged`, `apply`, `grad_tensor`, `epoch_loss`, and optionally
`step_staged_deferred` / `dw_chunks` / `dw_chunk_range` / `dw_chunk` / `rest_ranges` / `apply_ranges` / `param_tensor` /
`skip_step`), which is what lets the world_size-2/3 gloo tests drive this logic on CPU with a stand-in engine.
"""
from __future__ import annotations

import collections
import datetime
import os
import time

import numpy as np
import torch
import torch.distributed as dist

ALIGN = 4  # floats: shard boundaries stay 16-byte aligned (the Adam kernel's access width)


def shard_bounds(global_B: int, world: int, rank: int):
    """Contiguous split of a global minibatch; sizes differ by at most one (first ranks get the extra row)."""
    base, extra = divmod(global_B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_range(lo: int, hi: int, world: int):
    """(part, tail_lo): [lo, lo + world*part) is cut into `world` equal, 16-byte aligned parts; [tail_lo, hi) is the remainder"""
    part = ((hi - lo) // world) // ALIGN * ALIGN
    return part, lo + world * part


class _Done:
    def wait(self, timeout=None): pass
    def is_completed(self): return True


class CollectiveTimeout(RuntimeError):
    """a collective (or the device work queued behind it) did not finish within NTF_COLLECTIVE_TIMEOUT_S: the caller must leave with a non-zero exit code"""


def collective_timeout_s():
    return float(os.environ.get("NTF_COLLECTIVE_TIMEOUT_S", "120"))


class CollectiveTrace:
    """Every collective a step issues, by name, in a short ring: a bounded wait that expires says which one completed last and which one it is stuck behind
    (a hang on xGMI would otherwise be a silent `torch.cuda.synchronize()` that never returns).
    Under RCCL (`stream_ordered`) a `Work.wait()` WITHOUT a timeout only makes the current stream wait for the collective - the host runs ahead, which is what keeps
    the launch-ahead and the overlap of a step; with `timeout=` set, torch blocks the CPU thread until the work completes (Work.wait's docstring; ADVICE r4: every
    per-step wait then serialised host and device).  So the per-step waits of an RCCL run pass no timeout, and what is bounded on the host is the device synchronisation
    at the end of a phase (`sync`, an event poll).  gloo works complete on the host: there `wait` takes the timeout."""

    def __init__(self, who, stream_ordered=None):
        self.who, self.ring, self.issued = who, collections.deque(maxlen=256), 0
        # measure_waits (bench.py's instrumented pass, never the timed regions): an event pair around every stream-ordered wait - the stream does nothing between the two but
        # wait for the collective, so their distance IS the exposed wait of that collective class ('reduce_scatter', 'all_reduce', 'all_gather'); read by exposed_waits()
        self.measure_waits, self._wait_events, self._labels = False, [], {}
        if stream_ordered is None:
            stream_ordered = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
        self.stream_ordered = bool(stream_ordered)

    def add(self, label, work):
        self.issued += 1
        self.ring.append((self.issued, label, work))
        if self.measure_waits: self._labels[id(work)] = label
        return work

    def exposed_waits(self):
        """{collective class: total ms the stream sat waiting for it} since measure_waits was set; clears the record (synchronises the device)"""
        out = {}
        if self._wait_events: torch.cuda.synchronize()
        for cls, e0, e1 in self._wait_events:
            out[cls] = out.get(cls, 0.0) + e0.elapsed_time(e1)
        self._wait_events, self._labels = [], {}
        return out

    def describe(self):
        done = next((f"#{n} {lab}" for n, lab, w in reversed(self.ring) if _completed(w)), "none of the last %d" % len(self.ring))
        stuck = next((f"#{n} {lab}" for n, lab, w in self.ring if not _completed(w)), "none (device work behind them)")
        return f"last completed collective: {done}; first incomplete: {stuck}; issued: {self.issued}"

    def wait(self, work):
        if self.stream_ordered:
            ev = None
            if self.measure_waits and torch.cuda.is_available():
                label = self._labels.pop(id(work), "")
                cls = next((c for c in ("reduce_scatter", "all_gather", "all_reduce") if c in label), "other")
                ev = (cls, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[1].record()
            try:
                work.wait()      # orders the current stream behind the collective; the host does not block (a hang surfaces in `sync`, bounded)
                if ev: ev[2].record(); self._wait_events.append(ev)
            except RuntimeError as ex:      # an RCCL error raised at the wait (communicator aborted, a peer gone) keeps the contract: a naming message, a non-zero exit
                raise CollectiveTimeout(f"{self.who}: {type(ex).__name__}: {ex} [{self.describe()}]") from ex
            return
        try:
            work.wait(timeout=datetime.timedelta(seconds=collective_timeout_s()))
        except TypeError:
            work.wait()
        except RuntimeError as ex:
            raise CollectiveTimeout(f"{self.who}: {type(ex).__name__}: {ex} [{self.describe()}]") from ex

    def sync(self, what):
        """host waits for everything queued on the current stream, at most NTF_COLLECTIVE_TIMEOUT_S"""
        if not torch.cuda.is_available():
            return
        ev = torch.cuda.Event()
        ev.record()
        deadline, pause = time.monotonic() + collective_timeout_s(), 5e-5
        while not ev.query():
            if time.monotonic() > deadline:
                raise CollectiveTimeout(f"{self.who}: {what} did not finish within {collective_timeout_s():.0f} s [{self.describe()}]")
            time.sleep(pause)
            pause = min(pause * 2, 2e-3)


def _completed(work):
    try:
        return bool(work.is_completed())
    except Exception:
        return False


class DataParallel:
    def __init__(self, engine, group=None, overlap=True, shard_optimizer=None, emulate_world=0):
        """emulate_world = G > 1 (no process group; bench.py --dp-emulate): ONE rank's share of a G-rank step on one GPU - its 1/G of the global minibatch's rows through
        the deferred / chunked dW path, Adam on the 1/G shard it would own, NO exchange (the reduce-scatter / all-gather calls are skipped, their byte counts tallied in
        `emulated_bytes`): the per-rank compute time behind a scaling projection, not a measurement of the collectives."""
        self.engine = engine
        self.group = group
        self.emulate = int(emulate_world) if emulate_world and int(emulate_world) > 1 else 0
        self.world = self.emulate or (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = 0 if self.emulate else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.emulated_bytes = {"reduce_scatter_in": 0, "all_reduce": 0, "all_gather_out": 0, "steps": 0}
        self.sync_every = max(1, int(os.environ.get("NTF_DP_SYNC_EVERY", "64")))
        # The ranged head (forward kernel range by range behind its own chunks' all-gathers) costs a rank compute - four forward launches instead of one: +0.10 ms in the
        # one-rank emulation (profiles/r5_dp), +0.02 ms at world 1 under real RCCL self-collectives (profiles/r6_ab_dp_world1_ranges_*.json) - and pays only where an
        # all-gather has bytes to move.  Default by that evidence (round 6): ranged iff the group really has peers; NTF_DP_RANGES=1 / 0 forces it on / off (A/B runs, tests).
        rng_env = os.environ.get("NTF_DP_RANGES")
        self.ranged_head = (rng_env != "0") if rng_env is not None else (self.world > 1 and not self.emulate)
        self._grad = engine.grad_tensor()  # flat view of the engine's gradient buffer (HBM; aliases, no copy)
        # NTF_DP_FORCE_ALLREDUCE=1: run the collectives even at world_size 1 (exercises RCCL on the aliased buffers on a 1-GPU box)
        self.force_allreduce = (dist.is_initialized() and os.environ.get("NTF_DP_FORCE_ALLREDUCE", "0") == "1") or bool(self.emulate)
        self.n_chunks = engine.dw_chunks() if (overlap and hasattr(engine, "dw_chunks")) else 0
        if self.n_chunks:
            self._chunk_ranges = [engine.dw_chunk_range(k) for k in range(self.n_chunks)]  # identical on every rank
            self._rest = engine.rest_ranges()
        can_shard = hasattr(engine, "apply_ranges") and hasattr(engine, "param_tensor")
        self.shard = can_shard if shard_optimizer is None else (bool(shard_optimizer) and can_shard)
        self._param = engine.param_tensor() if self.shard else None
        # gloo (the CPU tests) has no reduce-scatter: there it is an all-reduce of which this rank keeps its part - same result
        self._native_rs = (not self.emulate) and dist.is_initialized() and dist.get_backend(group) == "nccl"
        self._pending = []   # parameter all-gathers of the previous step ...
        self._pending_chunk = {}   # ... those of the output layer's dW chunks by chunk index: a step whose head is pipelined waits for them range by range
        self.trace = CollectiveTrace(f"DataParallel rank {self.rank}/{self.world}", stream_ordered=self._native_rs)
        self._step_no = 0
        if self._grad.is_cuda and (self.world > 1 or self.force_allreduce) and hasattr(engine, "stream_handle"):
            assert engine.stream_handle is not None and torch.cuda.current_stream().cuda_stream == engine.stream_handle, \
                "DataParallel must run under torch.cuda.stream(s) with the engine created on s: kernels and collectives are ordered through that one stream"

    # ---- collectives on [lo, hi) of the flat buffers
    def _all_reduce(self, lo, hi):
        if self.emulate:
            self.emulated_bytes["all_reduce"] += 4 * (hi - lo)
            return _Done()
        return self.trace.add(f"step {self._step_no} all_reduce grad[{lo}:{hi}]",
                              dist.all_reduce(self._grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _reduce_scatter(self, lo, hi, owned, works):
        """sum over ranks of grad[lo:hi): this rank ends up with the reduced values of ITS part (and of the tail); records the part in
        `owned`, the collectives in `works`"""
        part, tail = shard_range(lo, hi, self.world)
        if part:
            mine = (lo + self.rank * part, lo + (self.rank + 1) * part)
            if self.emulate:
                self.emulated_bytes["reduce_scatter_in"] += 4 * (tail - lo)
            elif self._native_rs:
                works.append(self.trace.add(f"step {self._step_no} reduce_scatter grad[{lo}:{tail}]",
                                            dist.reduce_scatter_tensor(self._grad[mine[0]:mine[1]], self._grad[lo:tail], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            else:
                works.append(self._all_reduce(lo, tail))
            owned.append(mine)
        if tail < hi:
            works.append(self._all_reduce(tail, hi))
            owned.append((tail, hi))       # every rank updates the few tail elements (replicated, identical)

    def _all_gather(self, lo, hi):
        part, tail = shard_range(lo, hi, self.world)
        if not part:
            return _Done()
        if self.emulate:
            self.emulated_bytes["all_gather_out"] += 4 * (tail - lo)
            return _Done()
        mine = self._param[lo + self.rank * part: lo + (self.rank + 1) * part]
        label = f"step {self._step_no} all_gather param[{lo}:{tail}]"
        if self._native_rs:
            return self.trace.add(label, dist.all_gather_into_tensor(self._param[lo:tail], mine, group=self.group, async_op=True))
        parts = [self._param[lo + r * part: lo + (r + 1) * part] for r in range(self.world)]
        return self.trace.add(label, dist.all_gather(parts, mine.clone(), group=self.group, async_op=True))

    def _skip(self, zero_grad=True):
        """this rank's shard of the global minibatch is empty (a last batch smaller than the world size): contribute a zero gradient and
        advance the engine's step counter all the same - it keys the Flipout eps, which every rank must draw identically"""
        if zero_grad:
            self._grad.zero_()
        if hasattr(self.engine, "skip_step"):
            self.engine.skip_step()

    def _finish_gathers(self, keep_chunks=False):
        """wait for the parameter all-gathers of the previous step; keep_chunks: all but those of the output layer's dW chunks (the pipelined head waits for them itself)"""
        touched = bool(self._pending) or bool(self._pending_chunk)
        for w in self._pending:
            self.trace.wait(w)
        self._pending = []
        if not keep_chunks:
            for k in sorted(self._pending_chunk):
                for w in self._pending_chunk[k]:
                    self.trace.wait(w)
            self._pending_chunk = {}
        if touched and hasattr(self.engine, "params_touched"):
            self.engine.params_touched()      # the all-gather wrote the parameters through the raw view (include/opentf_amd.h: ntf_params_touched)

    def _train_step(self, goff, gB, lo, hi):
        """backward of this rank's shard of the batch + gradient exchange + Adam for one global minibatch"""
        e, have_rows = self.engine, hi > lo
        self._step_no += 1
        self.emulated_bytes["steps"] += 1
        # Pipelined head (round 5, engine.fwd_ranges): the output layer's operand producer and forward kernel run range by range, each behind the all-gathers of ITS dW
        # chunks only - RCCL moves the next range while the forward kernel works on this one.  Everything else the step reads (hidden layers, biases: updated on every
        # rank) is complete before it starts.
        spans = self.engine.fwd_ranges(hi - lo) if (self.ranged_head and have_rows and self.n_chunks and self.shard and self._pending_chunk and hasattr(self.engine, "fwd_ranges")) else []
        self._finish_gathers(keep_chunks=bool(spans))
        works, owned, gathered = [], [], []

        ranges_seen = []

        def before_range(j):
            ranges_seen.append(j)
            for k in range(spans[j][0], spans[j][1]):
                for w in self._pending_chunk.pop(k, ()):
                    self.trace.wait(w)
        reduce = (lambda a, b: self._reduce_scatter(a, b, owned, works)) if self.shard else (lambda a, b: works.append(self._all_reduce(a, b)))
        if self.n_chunks:
            if have_rows and spans:
                e.step_staged_deferred(goff + lo, hi - lo, goff, gB, before_range=before_range)
                if ranges_seen != list(range(len(spans))):
                    # Python's fwd_ranges() and the engine's in-step decision disagreed (the engine ran the whole-layer head): its producer and forward kernel were queued
                    # in front of the all-gathers still pending here, i.e. read parameters that had not arrived.  Never silently: this is a bug, not a slow path.
                    raise RuntimeError(f"data-parallel head: the engine called back for ranges {ranges_seen}, {len(spans)} were planned; the step read parameters whose all-gathers were still pending")
                self._finish_gathers()            # (every chunk belongs to a range: nothing is left; kept as the invariant)
            elif have_rows:
                e.step_staged_deferred(goff + lo, hi - lo, goff, gB)
            else:
                self._skip()
            for k, (ow, orr, cnt) in enumerate(self._chunk_ranges):
                if have_rows:
                    e.dw_chunk(k)                     # queue chunk k's kernel ...
                reduce(ow, ow + cnt); gathered.append((k, ow, ow + cnt))   # ... and let RCCL take its gradients as soon as it finishes
                if orr >= 0:
                    reduce(orr, orr + cnt); gathered.append((k, orr, orr + cnt))
            for rlo, rhi in self._rest:               # hidden layers, biases: small, replicated update
                works.append(self._all_reduce(rlo, rhi)); owned.append((rlo, rhi))
        else:
            if have_rows:
                e.step_staged(goff + lo, hi - lo, global_offset=goff, global_B=gB, train=True, apply=False)
            else:
                self._skip()
            n = self._grad.numel()
            reduce(0, n); gathered.append((None, 0, n))
        for w in works:
            self.trace.wait(w)
        if not self.shard:
            e.apply()
            return
        e.apply_ranges(sorted(owned))
        for k, a, b in gathered:
            w = self._all_gather(a, b)
            if k is None: self._pending.append(w)
            else: self._pending_chunk.setdefault(k, []).append(w)

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
            if self.trace.stream_ordered and self._grad.is_cuda and steps % self.sync_every == 0:
                # RCCL waits are stream-ordered and unbounded: without this the host queues the whole epoch behind a hung collective and blocks inside a full launch
                # queue, never reaching the bounded sync at the end of the phase.  One event poll every `sync_every` steps (NTF_DP_SYNC_EVERY, default 64).
                self.trace.sync(f"steps up to {steps} of the phase")
            if train:
                self._train_step(goff, gB, lo, hi)
            elif hi > lo:
                self.engine.step_staged(goff + lo, hi - lo, global_offset=goff, global_B=gB, train=False, apply=False)
            else:
                self._skip(zero_grad=False)
        self._finish_gathers()
        if collective and self._grad.is_cuda:
            self.trace.sync("the phase's kernels and collectives")      # bounded: the loss read-back below would wait for ever behind a hung collective
        s, _ = self.engine.epoch_loss()  # sum over steps of this rank's share of each batch loss
        t = torch.tensor([s], dtype=torch.float64, device=self._grad.device if self._grad.is_cuda else "cpu")
        if self.world > 1 and not self.emulate:
            self.trace.wait(self.trace.add("phase loss all_reduce", dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            if t.is_cuda: self.trace.sync("the loss all_reduce")
        return float(t.item()) / max(steps, 1)

    def train_epoch(self, order, global_B):
        return self._phase(order, global_B, True)

    def eval_epoch(self, order, global_B):
        return self._phase(order, global_B, False)
