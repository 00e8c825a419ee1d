"""`Fnn`: the reference's feed-forward multilabel team-formation classifier (src/mdl/fnn.py:13-219) with its
minibatch hot loop running on one or more MI355X through libopentf_amd.so.

What stays on the host, mirroring the reference line by line: parameter initialisation (same torch CPU draws in
the same order, so a given seed gives the reference's initial weights), the loader's batch order (same RNG
consumption as `DataLoader(..., shuffle=True)`), fold / epoch / early-stop / LR-plateau control, checkpoint and
prediction files (same keys, same layouts).  What moves to the GPU: everything inside `for batch` —
densification (labels stay CSR), forward, negative sampling, weighted BCE, backward, Adam.

There is no CPU path: `device` must name a GPU ('cuda', 'cuda:N' or 'cuda:0,1,...'); 'cpu' raises.
"""
from __future__ import annotations

import logging
import os
import re
import time
from collections import OrderedDict

import numpy as np
import scipy.sparse

from .earlystopping import EarlyStopping, PlateauLR
from .ntf import Ntf, cfg2str, cfg_get, dist_rank

log = logging.getLogger(__name__)


def parse_devices(device):
    """'cuda' -> [0]; 'cuda:3' -> [3]; 'cuda:0,1,2' -> [0,1,2] (the list form src/mdl/nmt.py:70 already parses)."""
    d = str(device)
    if not d.startswith("cuda"):
        raise RuntimeError(f"opentf_amd runs the fnn/bnn hot path on MI355X only; acceleration='{device}' has no implementation "
                           f"(no CPU fallback). Use the reference's mdl.fnn.Fnn for CPU runs.")
    if ":" not in d:
        return [0]
    return [int(x) for x in d.split(":", 1)[1].split(",") if x != ""]


class _Positions:
    """a dataset whose items are their own positions, fetched a whole batch at a time (`__getitems__`): the loader's sampler does all the work"""
    def __init__(self, n): self.n = int(n)
    def __len__(self): return self.n
    def __getitem__(self, i): return i
    def __getitems__(self, idx): return idx


def _loader_order(n, shuffle):
    """the order by running torch's own DataLoader (one batch of n: the batch size does not enter the draws or the order)"""
    import torch
    dl = torch.utils.data.DataLoader(_Positions(n), batch_size=max(int(n), 1), shuffle=shuffle, collate_fn=lambda x: x)
    parts = [np.asarray(t, np.int64) for t in dl]
    return np.concatenate(parts) if parts else np.empty(0, np.int64)


def _direct_order(n, shuffle):
    """the same draws without the loader's per-item Python: `_BaseDataLoaderIter.__init__` draws a base seed from the global CPU generator; a RandomSampler without a
    generator of its own then draws its seed the same way and yields `torch.randperm(n, generator=Generator().manual_seed(seed))`"""
    import torch
    torch.empty((), dtype=torch.int64).random_()
    if not shuffle: return np.arange(n, dtype=np.int64)
    g = torch.Generator(); g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    return torch.randperm(n, generator=g).numpy()


_direct_ok = None


def _direct_matches_loader():
    """once per process: does `_direct_order` leave the global generator and the order exactly as this torch's DataLoader does?  Checked on a scratch copy of the
    generator state, which is put back."""
    global _direct_ok
    if _direct_ok is None:
        import torch
        keep = torch.get_rng_state()
        try:
            ok = True
            for shuffle in (True, False):
                torch.set_rng_state(keep); a = _loader_order(257, shuffle); sa = torch.get_rng_state()
                torch.set_rng_state(keep); b = _direct_order(257, shuffle); sb = torch.get_rng_state()
                ok = ok and np.array_equal(a, b) and torch.equal(sa, sb)
            _direct_ok = bool(ok)
        finally:
            torch.set_rng_state(keep)
    return _direct_ok


def index_order(n, batch_size, shuffle):
    """Row positions in the order the reference's loader yields them (src/mdl/fnn.py:95-96,118), drawing from torch's
    global CPU generator exactly as `DataLoader(dataset, batch_size, shuffle)` does (one base-seed draw per iterator,
    one more for the RandomSampler permutation).  The draws are made directly (2 M positions in milliseconds) when a one-time check
    against this torch's own DataLoader agrees bit for bit - order and generator state - and through the DataLoader itself otherwise."""
    if int(n) <= 0:
        if shuffle: raise ValueError("num_samples should be a positive integer value, but got num_samples=0")   # what RandomSampler raises
        return _loader_order(0, False)
    return _direct_order(int(n), shuffle) if _direct_matches_loader() else _loader_order(int(n), shuffle)


def make_fnn(base):
    class Fnn(base):
        # ---- model definition, src/mdl/fnn.py:15-30
        def init(self, input_size, output_size):
            """Fresh parameters with the reference's draws: nn.Linear default init per layer in construction order, then
            xavier_uniform_ on every weight.  Returns (and keeps in self.model) the CPU state_dict."""
            import torch
            h = list(cfg_get(self.cfg, "h"))
            dims = [int(input_size)] + [int(x) for x in h] + [int(output_size)]
            layers = [torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
            for m in layers:
                torch.nn.init.xavier_uniform_(m.weight)
            sd = OrderedDict()
            for i, m in enumerate(layers):
                sd[f"layers.{i}.weight"] = m.weight.detach().clone()
                sd[f"layers.{i}.bias"] = m.bias.detach().clone()
            self.model = sd
            self._dims = dims
            return self.model

        # ---- engine plumbing
        def _engine(self, teamsvecs, max_batch, train=False):
            """The engine for this dataset.  With `self.keep_engine` set (tNtf's streaming loop does) the engine - CSR / table / dense
            input resident in HBM - survives learn() / test() and is reused while the caller passes the same matrices.
            train=True under torchrun may return an expert SHARD of the model (see _parallel_mode); test() always gets a whole-model engine."""
            key = (id(teamsvecs.get("skill_table")) if hasattr(teamsvecs, "get") else None, id(teamsvecs["skill"]), id(teamsvecs["member"]),
                   int(max_batch), str(self.device))
            if not hasattr(self, "_resident") or self._resident is None:
                self._resident = {}
            slot = "train" if train else "whole"
            cached = self._resident.get(slot)
            if cached is None and train:
                # an UNSHARDED training engine (one GPU, or data parallel) serves both roles and lives in the "whole" slot (below): reuse it when it is this
                # dataset's - tNtf's intervals then keep ONE engine resident instead of building a second one per interval
                w = self._resident.get("whole")
                if w is not None and w[0] == key and getattr(w[1], "parallel_mode", None) != "ep" and \
                        getattr(w[1], "parallel_mode", None) == self._parallel_mode(w[2], getattr(self, "_world", 1)):
                    return w[1], w[2]
            if cached is not None:
                if cached[0] == key:
                    return cached[1], cached[2]
                cached[1].close()
                del self._resident[slot]
            e, dims = self._new_engine(teamsvecs, max_batch, train)
            if getattr(self, "keep_engine", False):
                if getattr(e, "parallel_mode", None) != "ep" and train:     # an unsharded engine serves both roles
                    slot = "whole"
                    if slot in self._resident: self._resident[slot][1].close()
                self._resident[slot] = (key, e, dims)
            return e, dims

        def _release(self, engine):
            if not any(v[1] is engine for v in (getattr(self, "_resident", None) or {}).values()):
                engine.close()

        def release_engine(self):
            for v in (getattr(self, "_resident", None) or {}).values():
                v[1].close()
            self._resident = {}

        @staticmethod
        def _parallel_mode(dims, world):
            """How torchrun's processes share a training step (world > 1).  "ep": the output layer is split along the expert axis - every GPU
            steps the whole minibatch of cfg.b rows on its experts, the only exchange is d(hidden) (opentf_amd/ep.py); "dp": the rows are
            split, the gradients reduce-scattered (opentf_amd/dp.py).  NTF_PARALLEL = ep | dp forces one; default: ep whenever the model
            shards (h[-1] in {32, 64, 128} and at least one 256-expert tile per GPU) - it moves ~500x fewer bytes per step."""
            from ..ep import can_shard
            want = os.environ.get("NTF_PARALLEL", "auto").lower()
            if world <= 1:   # NTF_PARALLEL=ep with NTF_EP_FORCE_EXCHANGE=1: the sharded code path on one GPU (validation; the shard is the whole layer)
                return "ep" if (want == "ep" and os.environ.get("NTF_EP_FORCE_EXCHANGE", "0") == "1" and can_shard(dims, 1)) else None
            if want != "dp" and can_shard(dims, world):
                return "ep"
            if want == "ep":
                raise ValueError(f"NTF_PARALLEL=ep: a model of dims {dims} does not shard over {world} GPUs")
            return "dp"

        def _barrier(self):
            """ranks other than 0 must not read files rank 0 is still writing (checkpoints feed test() and tNtf's warm start)"""
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.barrier()

        def _model_dims(self, teamsvecs):
            """[D, *h, M] of the model for these matrices (D = the table's width when a skill table is registered)"""
            table = teamsvecs.get("skill_table") if hasattr(teamsvecs, "get") else None
            n_in = int(np.asarray(table).shape[1]) if table is not None else int(teamsvecs["skill"].shape[1])
            return [n_in] + [int(x) for x in cfg_get(self.cfg, "h")] + [int(teamsvecs["member"].shape[1])]

        def _new_engine(self, teamsvecs, max_batch, train=False):
            from .. import libntf
            skill, member = teamsvecs["skill"], teamsvecs["member"]
            dims = self._model_dims(teamsvecs)
            table = teamsvecs.get("skill_table") if hasattr(teamsvecs, "get") else None
            if table is not None:
                mode = libntf.INPUT_MEANPOOL
            elif scipy.sparse.issparse(skill):
                mode = libntf.INPUT_MULTIHOT
            else:
                mode = libntf.INPUT_DENSE
            devs = parse_devices(self.device)
            import torch
            self._rank, self._world = 0, 1
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                # one process per GPU (torchrun): rank r drives the r-th listed GPU, or GPU LOCAL_RANK when one is listed
                self._rank, self._world = torch.distributed.get_rank(), torch.distributed.get_world_size()
                local = int(os.environ.get("LOCAL_RANK", self._rank))
                devs = [devs[local % len(devs)] if len(devs) > 1 else local]
            torch.cuda.set_device(devs[0])
            # data parallel: engine kernels and the RCCL all-reduce must be ordered on ONE stream -> run both on a torch stream
            pmode = self._parallel_mode(dims, self._world) if train else None
            self._stream = torch.cuda.Stream() if (self._world > 1 or pmode) else None
            shard = None
            if pmode == "ep":
                from ..ep import expert_shards
                shard = expert_shards(dims[-1], self._world)[self._rank]
            nsd = cfg_get(self.cfg, "nsd")
            e = libntf.Engine(dims, bayesian=self.is_bayesian, input_mode=mode, max_batch=max_batch, ns=int(cfg_get(self.cfg, "ns", 0) or 0),
                              nsd=nsd if nsd else None, tpw=float(cfg_get(self.cfg, "tpw", 1)), tnw=float(cfg_get(self.cfg, "tnw", 1)),
                              lr=float(cfg_get(self.cfg, "lr")), seed=int(self.seed or 0), device=devs[0],
                              stream=self._stream.cuda_stream if self._stream is not None else None,
                              fuse_adam=1,   # the output layer's Adam in the dW epilogue (-0.16 ms per step at config 2); ignored under data parallelism
                              expert_shard=shard, ep_world=self._world if pmode == "ep" else 1)
            e.parallel_mode = pmode
            e.torch_stream = self._stream      # learn() runs under THIS engine's stream (a cached engine may be older than self._stream)
            if mode == libntf.INPUT_MEANPOOL:
                src = teamsvecs.get("original_skill", skill)
                e.set_skill_table(np.asarray(table, dtype=np.float32)); e.set_skill_csr(src)
            elif mode == libntf.INPUT_MULTIHOT:
                e.set_skill_csr(skill)
            else:
                e.set_dense_input(np.asarray(skill, dtype=np.float32))  # ndarray from D2v, np.matrix from Gnn (main.py:150)
            e.set_member(member)
            return e, dims

        @staticmethod
        def _to_torch(sd):
            import torch
            return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items())

        # ---- training, src/mdl/fnn.py:78-170
        def learn(self, teamsvecs, splits, prev_model):
            import contextlib
            import torch
            engine, dims = self._engine(teamsvecs, int(cfg_get(self.cfg, "b")), train=True)
            stream = getattr(engine, "torch_stream", None)
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                self._learn(engine, dims, teamsvecs, splits, prev_model)

        def _learn(self, engine, dims, teamsvecs, splits, prev_model):
            import torch
            member = teamsvecs["member"]
            b, lr = int(cfg_get(self.cfg, "b")), float(cfg_get(self.cfg, "lr"))
            spe = cfg_get(self.cfg, "spe")
            if cfg_get(self.cfg, "nsd") == "unigram":  # frequency of each expert over ALL teams, float64 (fnn.py:82)
                engine.set_unigram(np.asarray(member.sum(axis=0), dtype=np.float64).reshape(-1) / member.shape[0])

            runner = engine
            mode = getattr(engine, "parallel_mode", None)
            if mode == "ep":    # every GPU steps the cfg.b rows on its range of experts (ep.py)
                from ..ep import ExpertParallel
                runner = ExpertParallel(engine)
            elif mode == "dp":  # data parallel over the node's GPUs: global minibatch = cfg.b rows, split over ranks (dp.py)
                from ..dp import DataParallel
                runner = DataParallel(engine)
            self._runner = runner
            w = self.writer(log_dir=f"{self.output}/logs4tboard/run_{int(time.time())}")
            for foldidx in splits["folds"].keys():
                tr = np.asarray(splits["folds"][foldidx]["train"], dtype=np.int64)
                va = np.asarray(splits["folds"][foldidx]["valid"], dtype=np.int64)
                self.init(input_size=dims[0], output_size=dims[-1])
                if prev_model:
                    self.model = torch.load(prev_model[foldidx], map_location="cpu", weights_only=False)["model_state_dict"]
                engine.load_state_dict(self.model)
                engine.reset_optimizer()                       # fresh Adam(lr) per fold (fnn.py:104)
                scheduler = PlateauLR(lr, factor=0.1, patience=2)
                earlystopping = EarlyStopping(patience=int(cfg_get(self.cfg, "es")), verbose=True, delta=lr, trace_func=log.info)
                e = -1
                t_loss = v_loss = 0.0
                for e in range(int(cfg_get(self.cfg, "e"))):
                    t_loss = runner.train_epoch(tr[index_order(len(tr), b, True)], b)
                    v_loss = runner.eval_epoch(va[index_order(len(va), b, False)], b)
                    w.add_scalar(tag=f"{foldidx}_t_loss", scalar_value=t_loss, global_step=e)
                    w.add_scalar(tag=f"{foldidx}_v_loss", scalar_value=v_loss, global_step=e)
                    log.info(f"Fold {foldidx}/{len(splits['folds']) - 1}, Epoch {e}, Train Loss: {t_loss:.4f}")
                    log.info(f"Fold {foldidx}/{len(splits['folds']) - 1}, Epoch {e}, Valid Loss: {v_loss:.4f}")
                    if spe and (e == 0 or ((e + 1) % spe) == 0):
                        self._save(engine, foldidx, e, t_loss, v_loss, f"{self.output}/f{foldidx}.e{e}.pt")
                    engine.set_lr(scheduler.step(v_loss))
                    if earlystopping(v_loss, None).early_stop:
                        log.info(f"Early stopping triggered at epoch: {e}")
                        break
                self._save(engine, foldidx, e, t_loss, v_loss, f"{self.output}/f{foldidx}.pt")
                log.info(f"{self.name()} model with {cfg2str(self.cfg)} saved at {self.output}/f{foldidx}.pt")
            w.close()
            self._barrier()   # every f{k}.pt is complete on disk before any rank goes on to test() / the next interval
            self._release(engine)

        def _save(self, engine, foldidx, e, t_loss, v_loss, path):
            """Same keys and order as src/mdl/fnn.py:160,168; tensors are CPU f32 so `map_location` loads work anywhere."""
            import torch
            # expert shards: the output layer's rows are gathered from the ranks (a collective: every rank calls _save)
            sd = self._runner.state_dict() if getattr(engine, "parallel_mode", None) == "ep" else engine.state_dict()
            self.model = self._to_torch(sd)
            if getattr(self, "_rank", 0) != 0:
                return  # every rank holds the same weights; rank 0 writes the files
            torch.save({"model_state_dict": self.model, "cfg": self.cfg, "f": foldidx, "e": e, "t_loss": t_loss, "v_loss": v_loss}, path)

        # ---- inference, src/mdl/fnn.py:172-219
        def test(self, teamsvecs, splits, testcfg):
            import torch
            assert os.path.isdir(self.output), f"No folder for {self.output} exist!"
            b = int(cfg_get(self.cfg, "b"))
            import torch.distributed as dist
            world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
            # expert-sharded runs infer sharded too: every rank computes the probabilities (or the top-K candidates, and its share of the entropy sums) of its own
            # experts - the generators are keyed by global expert ids, so these ARE the whole model's columns - and rank 0 merges and writes
            sharded = world > 1 and self._parallel_mode(self._model_dims(teamsvecs), world) == "ep"
            writer = dist_rank() == 0
            if not sharded and not writer:
                # one writer: rank 0 runs the inference and writes the .pred files (the reference's test() is single-process);
                # the others wait for it so that evaluate() finds complete files
                self._barrier()
                return
            engine, dims = self._engine(teamsvecs, b, train=sharded)
            lo = engine.expert_lo if sharded else 0
            M = dims[-1]
            topK = cfg_get(testcfg, "topK")
            nmc = int(cfg_get(self.cfg, "nmc", 1) or 1)
            for foldidx in splits["folds"].keys():
                modelfiles = [f"{self.output}/f{foldidx}.pt"]
                if cfg_get(testcfg, "per_epoch"):
                    modelfiles += [f"{self.output}/{_}" for _ in os.listdir(self.output) if re.match(rf"f{foldidx}.e\d+.pt", _)]
                for modelfile in sorted(sorted(modelfiles), key=len):
                    self.model = torch.load(modelfile, map_location="cpu", weights_only=False)["model_state_dict"]
                    engine.load_state_dict(self.model)
                    for pred_set in (["test", "train", "valid"] if cfg_get(testcfg, "on_train") else ["test"]):
                        rows = np.asarray(splits["test"] if pred_set == "test" else splits["folds"][foldidx][pred_set], dtype=np.int64)
                        sparse_out = bool(topK) and topK < M
                        on_gpu_topk = sparse_out and topK <= 2048
                        dense, vals, idxs = [], [], []
                        pred_uncertainty, model_uncertainty = [], []
                        for o in range(0, len(rows), b):
                            rr = rows[o:o + b]
                            if self.is_bayesian:
                                pred_uncertainty, model_uncertainty = [], []  # re-initialised per batch, as the reference does (fnn.py:203)
                            if on_gpu_topk:
                                out = engine.forward_topk(rr, min(int(topK), engine.dims[-1]), nmc=nmc, uncertainty=self.is_bayesian)
                                v, i = out[0], out[1]
                                if sharded: v, i = self._merge_topk(v, i.astype(np.int64) + lo, int(topK))
                                vals.append(v); idxs.append(i)
                            else:
                                out = engine.forward(rr, nmc=nmc, uncertainty=self.is_bayesian)
                                p = out[0] if self.is_bayesian else out
                                dense.append(self._gather_columns(p) if sharded else p)
                            if self.is_bayesian:   # both entropies are sums over experts: a shard returns its share
                                pu, mu = (self._sum_over_ranks(out[-2]), self._sum_over_ranks(out[-1])) if sharded else (out[-2], out[-1])
                                pred_uncertainty.append(pu); model_uncertainty.append(mu)
                        if not writer:
                            continue
                        if on_gpu_topk:
                            y_pred = self._coo_from_topk(np.concatenate(vals), np.concatenate(idxs), (len(rows), M))
                        else:
                            y_pred = torch.from_numpy(np.concatenate(dense)) if dense else torch.empty(0, M)
                            if sparse_out:
                                y_pred = self._topk_sparse(y_pred, int(topK))
                        match = re.search(r"(e\d+)\.pt$", os.path.basename(modelfile))
                        epoch = (match.group(1) + ".") if match else ""
                        torch.save({"y_pred": y_pred, "uncertainty": {"pred": pred_uncertainty, "model": model_uncertainty} if self.is_bayesian else None},
                                   f"{self.output}/f{foldidx}.{pred_set}.{epoch}pred", pickle_protocol=4)
                        log.info(f"{self.name()} model predictions for fold{foldidx}.{pred_set}.{epoch} has saved at {self.output}/f{foldidx}.{pred_set}.{epoch}pred")
            self._barrier()
            self._release(engine)

        # ---- collectives of the sharded test() (small host arrays; nccl moves them through the GPU, gloo directly)
        @staticmethod
        def _all_gather_cols(a):
            """[B, m_r] per rank -> list of the ranks' arrays, on every rank"""
            import torch
            import torch.distributed as dist
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            world = dist.get_world_size()
            t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            n = torch.tensor([t.shape[1]], dtype=torch.int64, device=dev)
            counts = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(counts, n)
            counts = [int(c.item()) for c in counts]
            pad = torch.zeros((t.shape[0], max(counts)), dtype=t.dtype, device=dev)
            pad[:, : t.shape[1]] = t
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
            return [p[:, :c].cpu().numpy() for p, c in zip(parts, counts)]

        def _gather_columns(self, p):
            return np.concatenate(self._all_gather_cols(p), axis=1)

        def _merge_topk(self, vals, idx, k):
            """per-shard top-K candidates (global expert ids) -> the K largest of their union per row; ties go to the smaller expert id"""
            v = np.concatenate(self._all_gather_cols(vals), axis=1)
            i = np.concatenate(self._all_gather_cols(idx), axis=1)
            o = np.argsort(i, axis=1, kind="stable")
            v, i = np.take_along_axis(v, o, axis=1), np.take_along_axis(i, o, axis=1)
            o = np.argsort(-v, axis=1, kind="stable")[:, :k]
            return np.take_along_axis(v, o, axis=1), np.take_along_axis(i, o, axis=1)

        @staticmethod
        def _sum_over_ranks(a):
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.cpu().numpy()

        @staticmethod
        def _topk_sparse(probs, k):
            """src/pkgmgr.py:125-134"""
            import torch
            v, i = torch.topk(probs, k, dim=1)
            rows = torch.arange(probs.shape[0]).unsqueeze(1).expand(-1, k)
            return torch.sparse_coo_tensor(torch.stack([rows, i], dim=0).reshape(2, -1), v.reshape(-1), size=probs.shape).coalesce()

        @staticmethod
        def _coo_from_topk(vals, idx, shape):
            """The coalesced COO tensor topk_sparse() returns (indices sorted by row, then column), from the GPU top-K."""
            import torch
            order = np.argsort(idx, axis=1, kind="stable")
            idx_s = np.take_along_axis(idx, order, axis=1).astype(np.int64)
            val_s = np.take_along_axis(vals, order, axis=1)
            rows = np.repeat(np.arange(shape[0], dtype=np.int64), idx.shape[1])
            ind = torch.from_numpy(np.stack([rows, idx_s.reshape(-1)]))
            return torch.sparse_coo_tensor(ind, torch.from_numpy(val_s.reshape(-1)), size=shape, is_coalesced=True)

    Fnn.__qualname__ = "Fnn"
    return Fnn


Fnn = make_fnn(Ntf)
