"""ctypes binding of libopentf_amd.so (include/opentf_amd.h) and a thin `Engine` object over the handle.

There is no CPU fallback: if the shared library is missing or the HIP device cannot be opened, this
module raises.  Build the library with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C opentf_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import numpy as np

NTF_ABI_VERSION = 1
NTF_MAX_LAYERS = 8
INPUT_DENSE, INPUT_MEANPOOL, INPUT_MULTIHOT = 0, 1, 2
NSD = {None: 0, "": 0, "None": 0, "uniform": 1, "unigram": 2, "unigram_b": 3}
P_WEIGHT, P_BIAS, P_RHO_WEIGHT, P_RHO_BIAS = 0, 1, 2, 3

_RANGE_CB = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_void_p)
_LIB_PATH = os.environ.get("NTF_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libopentf_amd.so")   # (NTF_LIB_PATH: A/B builds of the same ABI)


class NtfError(RuntimeError):
    pass


class ntf_config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p), ("n_layers", C.c_int32),
                ("dims", C.c_int32 * (NTF_MAX_LAYERS + 1)), ("bayesian", C.c_int32), ("input_mode", C.c_int32),
                ("max_batch", C.c_int32), ("ns", C.c_int32), ("nsd", C.c_int32), ("tpw", C.c_float), ("tnw", C.c_float),
                ("lr", C.c_float), ("seed", C.c_uint64), ("fused", C.c_int32), ("fuse_adam", C.c_int32), ("mfma", C.c_int32),
                ("expert_lo", C.c_int32), ("experts_global", C.c_int32), ("ep_world", C.c_int32), ("reserved", C.c_int32 * 2)]


class ntf_inject(C.Structure):
    _fields_ = [("neg_idx", C.c_void_p), ("eps_w", C.c_void_p * NTF_MAX_LAYERS), ("eps_b", C.c_void_p * NTF_MAX_LAYERS),
                ("s_in", C.c_void_p * NTF_MAX_LAYERS), ("s_out", C.c_void_p * NTF_MAX_LAYERS)]


# every symbol declared in include/opentf_amd.h: name -> (restype, argtypes)
_P, _I32, _I64, _F, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64
SYMBOLS = {
    "ntf_engine_create": (C.c_int, [C.POINTER(ntf_config), C.POINTER(_P)]),
    "ntf_engine_destroy": (None, [_P]),
    "ntf_last_error": (C.c_char_p, [_P]),
    "ntf_abi_version": (C.c_int, []),
    "ntf_set_member_csr": (C.c_int, [_P, _P, _P, _I64]),
    "ntf_set_skill_csr": (C.c_int, [_P, _P, _P, _I64]),
    "ntf_set_skill_table": (C.c_int, [_P, _P, _I64, _I32]),
    "ntf_set_dense_input": (C.c_int, [_P, _P, _I64, _I32]),
    "ntf_set_unigram": (C.c_int, [_P, _P, _I64]),
    "ntf_set_param": (C.c_int, [_P, C.c_int, C.c_int, _P, _I64]),
    "ntf_get_param": (C.c_int, [_P, C.c_int, C.c_int, _P, _I64]),
    "ntf_get_grad": (C.c_int, [_P, C.c_int, C.c_int, _P, _I64]),
    "ntf_reset_optimizer": (C.c_int, [_P]),
    "ntf_set_lr": (C.c_int, [_P, _F]),
    "ntf_set_seed": (C.c_int, [_P, _U64, _U64]),
    "ntf_skip_step": (C.c_int, [_P]),
    "ntf_range_fallbacks": (C.c_int, [_P, C.POINTER(_I64)]),
    "ntf_prefetched_steps": (C.c_int, [_P, C.POINTER(_I64)]),
    "ntf_head_prefetch_hits": (C.c_int, [_P, C.POINTER(_I64)]),
    "ntf_first_layer_sweeps": (C.c_int, [_P, C.POINTER(_I64)]),
    "ntf_get_dlogits": (C.c_int, [_P, _P, _I64]),
    "ntf_get_negatives": (C.c_int, [_P, _P, _I64]),
    "ntf_fwd_ranges": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ntf_step_staged_deferred_cb": (C.c_int, [_P, _I64, C.c_int32, _I64, C.c_int32, _P, _RANGE_CB, _P]),
    "ntf_get_noise": (C.c_int, [_P, _U64, C.c_int32, C.c_int32, C.c_int32, _P, _I64]),
    "ntf_train_step": (C.c_int, [_P, _P, _I32, _P, _P]),
    "ntf_eval_step": (C.c_int, [_P, _P, _I32, _P, _P]),
    "ntf_backward": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "ntf_apply": (C.c_int, [_P]),
    "ntf_apply_ranges": (C.c_int, [_P, _P, _I32]),
    "ntf_train_epoch": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "ntf_eval_epoch": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "ntf_epoch_loss": (C.c_int, [_P, _P, _P]),
    "ntf_stage_order": (C.c_int, [_P, _P, _I64]),
    "ntf_step_staged": (C.c_int, [_P, _I64, _I32, _I64, _I32, _I32, _I32, _P]),
    "ntf_step_staged_deferred": (C.c_int, [_P, _I64, _I32, _I64, _I32, _P]),
    "ntf_dw_chunks": (C.c_int, [_P, C.POINTER(_I32)]),
    "ntf_dw_chunk_range": (C.c_int, [_P, _I32, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    "ntf_dw_chunk": (C.c_int, [_P, _I32]),
    "ntf_param_segment": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_I64), C.POINTER(_I64)]),
    "ntf_step_staged_ep": (C.c_int, [_P, _I64, _I32, _I32]),
    "ntf_dh_buffer": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(_I64)]),
    "ntf_forward": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P, _P]),
    "ntf_logits": (C.c_int, [_P, _P, _I32, _P, _P]),
    "ntf_forward_topk": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P]),
    "ntf_gather_meanpool": (C.c_int, [_P, _P, _I64, _P]),
    "ntf_grad_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "ntf_moment_buffers": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(_I64)]),
    "ntf_param_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "ntf_params_touched": (C.c_int, [_P]),
    "ntf_synchronize": (C.c_int, [_P]),
    "ntf_kernel_times": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int]),
    "ntf_rank_metrics": (C.c_int, [C.c_int, _P, _I64, _I32, _P, _P, _I64, _P, _P, _I32, _P]),
    "ntf_skill_coverage": (C.c_int, [C.c_int, _P, _I64, _I32, _P, _P, _I64, _P, _P, _P, _I64, _P, _I32, _P]),
    "ntf_skill_cooccurrence": (C.c_int, [C.c_int, _I64, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "ntf_csr_result_fetch": (C.c_int, [_P, _P, _P, _P, _P]),
    "ntf_csr_result_free": (None, [_P]),
    "ntf_n2v_create": (C.c_int, [C.c_int, _I64, _I32, _P, _P, _P, _U64, C.POINTER(_P)]),
    "ntf_n2v_destroy": (None, [_P]),
    "ntf_n2v_last_error": (C.c_char_p, [_P]),
    "ntf_n2v_walks": (C.c_int, [_P, _P, _I64, _I32, _U64, _P]),
    "ntf_n2v_train_batch": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _F, _P, _I64, _P, _I64, _I32, _P]),
    "ntf_n2v_get": (C.c_int, [_P, C.c_int, _P]),
    "ntf_n2v_edge_bce": (C.c_int, [_P, _P, _P, _I64, _P]),
    "ntf_d2v_create": (C.c_int, [C.c_int, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P, _U64, C.POINTER(_P)]),
    "ntf_d2v_destroy": (None, [_P]),
    "ntf_d2v_last_error": (C.c_char_p, [_P]),
    "ntf_d2v_train_epoch": (C.c_int, [_P, _I32, _I32, _I32, C.c_double, C.c_double, _U64, _I32, _P, _P, _P, _P]),
    "ntf_d2v_get": (C.c_int, [_P, C.c_int, _P]),
    "ntf_d2v_set": (C.c_int, [_P, C.c_int, _P]),
    "ntf_k_gemm_f32": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _I64, _I64, _P, _I64, _I64, _P, _I64]),
    "ntf_k_fill_normal": (C.c_int, [_P, _U64, _U64, C.c_int, _I64, _P]),
    "ntf_k_fill_sign": (C.c_int, [_P, _U64, _U64, C.c_int, C.c_int, C.c_int, _P]),
}

_lib = None


def lib():
    """The loaded shared library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NtfError(f"{_LIB_PATH} is missing: build it (make -C opentf_amd/csrc). There is no CPU fallback.")
        import torch  # noqa: F401  (load torch's bundled HIP runtime first: this library must share it, not bring a second copy)
        l = C.CDLL(_LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if l.ntf_abi_version() != NTF_ABI_VERSION:
            raise NtfError("libopentf_amd.so ABI version mismatch")
        _lib = l
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class DeviceView:
    """`__cuda_array_interface__` view of an engine-owned HBM buffer (for torch.as_tensor / RCCL)."""

    def __init__(self, ptr, n, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}


class Engine:
    """One Fnn/Bnn model resident on one MI355X.  dims = [D, *h, M].
    expert_shard = (lo, hi), ep_world = G: this engine owns the experts [lo, hi) of the output layer only (one of G such engines, one per GPU;
    hidden layers replicated; see ntf_config.expert_lo in include/opentf_amd.h and opentf_amd/ep.py).  `dims[-1]` then is hi - lo, `experts_global` is M."""

    def __init__(self, dims, bayesian=False, input_mode=INPUT_DENSE, max_batch=1000, ns=5, nsd="uniform", tpw=10.0, tnw=1.0,
                 lr=1e-3, seed=0, device=0, stream=None, fused=True, fuse_adam=False, mfma=None, expert_shard=None, ep_world=1):
        self.dims = [int(d) for d in dims]
        self.experts_global = self.dims[-1]
        self.expert_lo, self.ep_world = 0, int(ep_world)
        if expert_shard is not None:
            lo, hi = (int(v) for v in expert_shard)
            if not 0 <= lo < hi <= self.experts_global:
                raise NtfError(f"expert_shard {expert_shard} outside [0, {self.experts_global})")
            self.expert_lo, self.dims[-1] = lo, hi - lo
        self.L = len(self.dims) - 1
        if not 1 <= self.L <= NTF_MAX_LAYERS:
            raise NtfError("between 1 and 8 layers supported")
        self.bayesian = bool(bayesian)
        self.ns = int(ns) if NSD[nsd] else 0
        self.max_batch = int(max_batch)
        cfg = ntf_config()
        cfg.abi_version, cfg.device, cfg.stream, cfg.n_layers = NTF_ABI_VERSION, int(device), stream, self.L
        for i, d in enumerate(self.dims):
            cfg.dims[i] = d
        cfg.bayesian, cfg.input_mode, cfg.max_batch = int(self.bayesian), int(input_mode), self.max_batch
        cfg.ns, cfg.nsd, cfg.tpw, cfg.tnw, cfg.lr, cfg.seed, cfg.fused = max(self.ns, 0), NSD[nsd], float(tpw), float(tnw), float(lr), int(seed) & (2**64 - 1), int(bool(fused))
        if mfma == "bf16x6": raise ValueError("mfma='bf16x6' was retired in round 6 (fp16x3, the default, has its accuracy at half the matrix work); use None or 'f32'")
        cfg.mfma = {None: 0, "default": 0, "f32": 1, "fp16x3": 3}[mfma] if not isinstance(mfma, int) or isinstance(mfma, bool) else int(mfma)
        cfg.fuse_adam = int(fuse_adam)  # 0: flat Adam kernel; 1: inside the dW epilogue; 2: chunked beside the dW kernel on a side stream
        if expert_shard is not None or self.ep_world > 1:
            cfg.expert_lo, cfg.experts_global, cfg.ep_world = self.expert_lo, self.experts_global, self.ep_world
        self.stream_handle = stream   # the caller's hipStream_t (int) the engine runs on, None: a stream of its own
        self._h = C.c_void_p()
        rc = lib().ntf_engine_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = lib().ntf_last_error(None)
            self._h = None
            raise NtfError(f"ntf_engine_create failed ({rc}): {msg.decode() if msg else ''}")
        self._keep = []

    # ---- plumbing
    def _ck(self, rc):
        if rc != 0:
            raise NtfError(f"libopentf_amd error {rc}: {lib().ntf_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            lib().ntf_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data
    @staticmethod
    def _csr(mat_or_tuple):
        if isinstance(mat_or_tuple, tuple):
            indptr, indices = mat_or_tuple
        else:
            import scipy.sparse
            m = scipy.sparse.csr_matrix(mat_or_tuple)
            m.sort_indices()
            indptr, indices = m.indptr, m.indices
        return np.ascontiguousarray(indptr, dtype=np.int64), np.ascontiguousarray(indices, dtype=np.int32)

    def set_member(self, member):
        ip, ix = self._csr(member)
        self._ck(lib().ntf_set_member_csr(self._h, _ptr(ip), _ptr(ix), len(ip) - 1))

    def set_skill_csr(self, skill):
        ip, ix = self._csr(skill)
        self._ck(lib().ntf_set_skill_csr(self._h, _ptr(ip), _ptr(ix), len(ip) - 1))

    def set_skill_table(self, table):
        t = _f32(table)
        self._ck(lib().ntf_set_skill_table(self._h, _ptr(t), t.shape[0], t.shape[1]))

    def set_dense_input(self, X):
        x = _f32(X)
        self._ck(lib().ntf_set_dense_input(self._h, _ptr(x), x.shape[0], x.shape[1]))

    def set_unigram(self, freq):
        f = np.ascontiguousarray(np.asarray(freq, dtype=np.float64).reshape(-1))
        self._ck(lib().ntf_set_unigram(self._h, _ptr(f), len(f)))

    # ---- state (reference state_dict layout, src/mdl/fnn.py:20-22 / bayesian-torch LinearFlipout)
    def _kinds(self):
        return ([("mu_weight", P_WEIGHT), ("rho_weight", P_RHO_WEIGHT), ("mu_bias", P_BIAS), ("rho_bias", P_RHO_BIAS)] if self.bayesian
                else [("weight", P_WEIGHT), ("bias", P_BIAS)])

    def _shape(self, layer, kind):
        o, i = self.dims[layer + 1], self.dims[layer]
        return (o, i) if kind in (P_WEIGHT, P_RHO_WEIGHT) else (o,)

    def load_state_dict(self, sd):
        for l in range(self.L):
            for name, kind in self._kinds():
                a = sd[f"layers.{l}.{name}"]
                a = _f32(a.detach().cpu().numpy() if hasattr(a, "detach") else a)
                if l == self.L - 1 and a.shape[0] == self.experts_global != self.dims[-1]:   # an expert shard takes its rows of the whole layer
                    a = _f32(a[self.expert_lo: self.expert_lo + self.dims[-1]])
                if a.shape != self._shape(l, kind):
                    raise NtfError(f"layers.{l}.{name}: shape {a.shape} != {self._shape(l, kind)}")
                self._ck(lib().ntf_set_param(self._h, l, kind, _ptr(a), a.size))

    def state_dict(self):
        sd = OrderedDict()
        for l in range(self.L):
            for name, kind in self._kinds():
                a = np.empty(self._shape(l, kind), dtype=np.float32)
                self._ck(lib().ntf_get_param(self._h, l, kind, _ptr(a), a.size))
                sd[f"layers.{l}.{name}"] = a
        return sd

    def grads(self):
        """p.grad of every parameter after backward()/train_step() (state_dict key order)."""
        sd = OrderedDict()
        for l in range(self.L):
            for name, kind in self._kinds():
                a = np.empty(self._shape(l, kind), dtype=np.float32)
                self._ck(lib().ntf_get_grad(self._h, l, kind, _ptr(a), a.size))
                sd[f"layers.{l}.{name}"] = a
        return sd

    def reset_optimizer(self):
        self._ck(lib().ntf_reset_optimizer(self._h))

    def set_lr(self, lr):
        self._ck(lib().ntf_set_lr(self._h, float(lr)))

    def set_seed(self, seed, step=0):
        self._ck(lib().ntf_set_seed(self._h, int(seed) & (2**64 - 1), int(step)))

    def skip_step(self):
        """advance the generators' step counter without running a step (a data-parallel rank with an empty shard)"""
        self._ck(lib().ntf_skip_step(self._h))

    def range_fallbacks(self):
        """steps / inference calls that ran on the exact-f32 kernels because an operand left the fp16x3 window"""
        n = C.c_int64()
        self._ck(lib().ntf_range_fallbacks(self._h, C.byref(n)))
        return n.value

    def prefetched_steps(self):
        """steps that started on output-layer operands written by the previous step's dW + Adam epilogue (no producer pass of their own)"""
        n = C.c_int64()
        self._ck(lib().ntf_prefetched_steps(self._h, C.byref(n)))
        return n.value

    def head_prefetch_hits(self):
        """steps whose sampler / gather / hidden layer had run beside the previous step's dW kernel (include/opentf_amd.h)"""
        n = C.c_int64()
        self._ck(lib().ntf_head_prefetch_hits(self._h, C.byref(n)))
        return n.value

    def first_layer_sweeps(self):
        """steps that started on a multi-hot Flipout first-layer operand written by the previous step's one-pass finalize + Adam + producer (include/opentf_amd.h)"""
        n = C.c_int64()
        self._ck(lib().ntf_first_layer_sweeps(self._h, C.byref(n)))
        return n.value

    def dlogits(self, B):
        """d loss / d z [B, M] of the output layer after the last backward (B = that step's batch)"""
        out = np.empty((int(B), self.dims[-1]), dtype=np.float32)
        self._ck(lib().ntf_get_dlogits(self._h, _ptr(out), out.size))
        return out

    def negatives(self, B):
        """the last step's sampled negatives [B, ns] (global expert ids)"""
        out = np.empty((int(B), int(self.ns)), dtype=np.int64)
        self._ck(lib().ntf_get_negatives(self._h, _ptr(out), out.size))
        return out

    def noise(self, step, B):
        """the device generators' own Flipout draws of step index `step` (include/opentf_amd.h ntf_get_noise), as the oracle's per-layer list of
        {eps_w [out, in], eps_b [out], s_in [B, in], s_out [B, out]}"""
        out = []
        for l in range(len(self.dims) - 1):
            n_in, n_out = int(self.dims[l]), int(self.dims[l + 1])
            d = {"eps_w": np.empty((n_out, n_in), np.float32), "eps_b": np.empty(n_out, np.float32),
                 "s_in": np.empty((int(B), n_in), np.float32), "s_out": np.empty((int(B), n_out), np.float32)}
            for kind, key in enumerate(("eps_w", "eps_b", "s_in", "s_out")):
                self._ck(lib().ntf_get_noise(self._h, int(step), l, kind, int(B), _ptr(d[key]), d[key].size))
            out.append(d)
        return out

    # ---- steps
    def _inject(self, inject, B):
        if not inject:
            return None, []
        keep, st = [], ntf_inject()
        if inject.get("neg_idx") is not None:
            a = np.ascontiguousarray(np.asarray(inject["neg_idx"], dtype=np.int64)); keep.append(a)
            assert a.shape == (B, self.ns), (a.shape, B, self.ns)
            st.neg_idx = a.ctypes.data
        for key in ("eps_w", "eps_b", "s_in", "s_out"):
            for l, a in enumerate(inject.get(key) or []):
                if a is None:
                    continue
                a = _f32(a.detach().cpu().numpy() if hasattr(a, "detach") else a); keep.append(a)
                getattr(st, key)[l] = a.ctypes.data
        return st, keep

    @staticmethod
    def _rows(rows):
        return np.ascontiguousarray(np.asarray(rows, dtype=np.int64).reshape(-1))

    def train_step(self, rows, inject=None, want_loss=True):
        r = self._rows(rows); st, keep = self._inject(inject, len(r)); loss = C.c_float()
        self._ck(lib().ntf_train_step(self._h, _ptr(r), len(r), C.byref(st) if st else None, C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def eval_step(self, rows, inject=None, want_loss=True):
        r = self._rows(rows); st, keep = self._inject(inject, len(r)); loss = C.c_float()
        self._ck(lib().ntf_eval_step(self._h, _ptr(r), len(r), C.byref(st) if st else None, C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def backward(self, rows, global_B=None, inject=None, want_loss=True):
        r = self._rows(rows); st, keep = self._inject(inject, len(r)); loss = C.c_float()
        self._ck(lib().ntf_backward(self._h, _ptr(r), len(r), int(global_B or len(r)), C.byref(st) if st else None, C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def apply(self):
        self._ck(lib().ntf_apply(self._h))

    def apply_ranges(self, ranges):
        """one Adam step on the [lo, hi) float ranges of the flat buffers only (sharded optimiser state, dp.py)"""
        a = np.ascontiguousarray(np.asarray(ranges, dtype=np.int64).reshape(-1))
        self._ck(lib().ntf_apply_ranges(self._h, _ptr(a), len(a) // 2))

    def param_tensor(self):
        """torch tensor aliasing the flat parameter buffer in HBM (what the sharded step all-gathers)"""
        import torch
        return torch.as_tensor(self.param_view(), device=f"cuda:{torch.cuda.current_device()}")

    def params_touched(self):
        """after writing parameters through param_tensor() / param_view(): drop operands a fused step prepared from the old values"""
        self._ck(lib().ntf_params_touched(self._h))

    def moment_tensors(self):
        """torch tensors aliasing Adam's exp_avg / exp_avg_sq buffers (flat, the parameters' layout)"""
        import torch
        m, v, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        self._ck(lib().ntf_moment_buffers(self._h, C.byref(m), C.byref(v), C.byref(n)))
        dev = f"cuda:{torch.cuda.current_device()}"
        return (torch.as_tensor(DeviceView(m.value, n.value, self), device=dev), torch.as_tensor(DeviceView(v.value, n.value, self), device=dev))

    def train_epoch(self, order, B):
        o = self._rows(order); loss = C.c_float()
        self._ck(lib().ntf_train_epoch(self._h, _ptr(o), len(o), int(B), C.byref(loss)))
        return loss.value

    def eval_epoch(self, order, B):
        o = self._rows(order); loss = C.c_float()
        self._ck(lib().ntf_eval_epoch(self._h, _ptr(o), len(o), int(B), C.byref(loss)))
        return loss.value

    def stage_order(self, order):
        o = self._rows(order)
        self._ck(lib().ntf_stage_order(self._h, _ptr(o), len(o)))

    def step_staged(self, offset, B, global_offset=None, global_B=None, train=True, apply=True, want_loss=False):
        loss = C.c_float()
        self._ck(lib().ntf_step_staged(self._h, int(offset), int(B), int(offset if global_offset is None else global_offset),
                                       int(B if global_B is None else global_B), int(train), int(apply), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    # ---- expert-sharded output layer (see include/opentf_amd.h, opentf_amd/ep.py)
    def step_staged_ep(self, offset, B, phase):
        self._ck(lib().ntf_step_staged_ep(self._h, int(offset), int(B), int(phase)))

    def dh_tensor(self):
        """torch tensor aliasing the [max_batch * h[-1]] buffer of d(hidden): phase 1 leaves this shard's partial sum there, phase 2 reads the total"""
        import torch
        p, n = C.c_void_p(), C.c_int64()
        self._ck(lib().ntf_dh_buffer(self._h, C.byref(p), C.byref(n)))
        if not n.value:
            return None
        return torch.as_tensor(DeviceView(p.value, n.value, self), device=f"cuda:{torch.cuda.current_device()}")

    # ---- data-parallel pipelining (see include/opentf_amd.h)
    def step_staged_deferred(self, offset, B, global_offset, global_B, before_range=None):
        """before_range(j): called in front of forward range j (see fwd_ranges) - the caller orders the engine's stream behind that range's parameter all-gathers there"""
        if before_range is None:
            self._ck(lib().ntf_step_staged_deferred(self._h, int(offset), int(B), int(global_offset), int(global_B), None))
            return
        failure = []

        def _cb(j, _user):
            try:
                before_range(int(j))
                return 0
            except BaseException as ex:      # (never let an exception cross the C frame: report it, re-raise below)
                failure.append(ex)
                return 1
        cb = _RANGE_CB(_cb)
        rc = lib().ntf_step_staged_deferred_cb(self._h, int(offset), int(B), int(global_offset), int(global_B), None, cb, None)
        if failure: raise failure[0]
        self._ck(rc)

    def fwd_ranges(self, B):
        """[(k0, k1), ...]: the dW chunks of each forward range of a data-parallel step of B rows (empty: the head is not pipelined for this model)"""
        n = C.c_int32(); spans = (C.c_int32 * 8)()
        self._ck(lib().ntf_fwd_ranges(self._h, int(B), C.byref(n), spans))
        return [(int(spans[2 * j]), int(spans[2 * j + 1])) for j in range(n.value)]

    def dw_chunks(self):
        n = C.c_int32()
        self._ck(lib().ntf_dw_chunks(self._h, C.byref(n)))
        return n.value

    def dw_chunk_range(self, k):
        ow, orr, cnt = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(lib().ntf_dw_chunk_range(self._h, int(k), C.byref(ow), C.byref(orr), C.byref(cnt)))
        return ow.value, orr.value, cnt.value

    def dw_chunk(self, k):
        self._ck(lib().ntf_dw_chunk(self._h, int(k)))

    def param_segment(self, layer, kind):
        """(offset, count) in floats of a parameter inside the flat parameter / gradient buffers"""
        o, c = C.c_int64(), C.c_int64()
        self._ck(lib().ntf_param_segment(self._h, int(layer), int(kind), C.byref(o), C.byref(c)))
        return o.value, c.value

    def rest_ranges(self):
        """[lo, hi) float ranges of the flat gradient buffer outside the output layer's weight / rho_weight segments."""
        n = self.grad_view().__cuda_array_interface__["shape"][0]
        cuts = []
        for kind in ([P_WEIGHT, P_RHO_WEIGHT] if self.bayesian else [P_WEIGHT]):
            o, c = C.c_int64(), C.c_int64()
            self._ck(lib().ntf_param_segment(self._h, self.L - 1, kind, C.byref(o), C.byref(c)))
            cuts.append((o.value, o.value + c.value))
        out, pos = [], 0
        for lo, hi in sorted(cuts):
            if lo > pos: out.append((pos, lo))
            pos = hi
        if pos < n: out.append((pos, n))
        return out

    def epoch_loss(self):
        s, k = C.c_double(), C.c_int64()
        self._ck(lib().ntf_epoch_loss(self._h, C.byref(s), C.byref(k)))
        return s.value, k.value

    # ---- inference
    def logits(self, rows, inject=None):
        r = self._rows(rows); st, keep = self._inject(inject, len(r))
        out = np.empty((len(r), self.dims[-1]), dtype=np.float32)
        self._ck(lib().ntf_logits(self._h, _ptr(r), len(r), C.byref(st) if st else None, _ptr(out)))
        return out

    def forward(self, rows, nmc=1, injects=None, uncertainty=False):
        r = self._rows(rows); B = len(r)
        out = np.empty((B, self.dims[-1]), dtype=np.float32)
        pu = np.empty(B, dtype=np.float32) if uncertainty else None
        mu = np.empty(B, dtype=np.float32) if uncertainty else None
        arr, keep = None, []
        if injects:
            arr = (ntf_inject * len(injects))()
            for i, inj in enumerate(injects):
                st, k = self._inject(inj, B); keep.append(k); arr[i] = st
        self._ck(lib().ntf_forward(self._h, _ptr(r), B, int(nmc), arr, _ptr(out), _ptr(pu), _ptr(mu)))
        return (out, pu, mu) if uncertainty else out

    def forward_topk(self, rows, K, nmc=1, uncertainty=False):
        r = self._rows(rows); B = len(r)
        vals = np.empty((B, K), dtype=np.float32); idx = np.empty((B, K), dtype=np.int32)
        pu = np.empty(B, dtype=np.float32) if uncertainty else None
        mu = np.empty(B, dtype=np.float32) if uncertainty else None
        self._ck(lib().ntf_forward_topk(self._h, _ptr(r), B, int(nmc), int(K), _ptr(vals), _ptr(idx), _ptr(pu), _ptr(mu)))
        return (vals, idx, pu, mu) if uncertainty else (vals, idx)

    def gather_meanpool(self, rows=None, n=None, to_host=True):
        r = None if rows is None else self._rows(rows)
        n = len(r) if r is not None else int(n)
        out = np.empty((n, self._table_d()), dtype=np.float32) if to_host else None
        self._ck(lib().ntf_gather_meanpool(self._h, _ptr(r), n, _ptr(out)))
        return out

    def _table_d(self):
        return self.dims[0]

    # ---- views / measurement
    def grad_view(self):
        p, n = C.c_void_p(), C.c_int64()
        self._ck(lib().ntf_grad_buffer(self._h, C.byref(p), C.byref(n)))
        return DeviceView(p.value, n.value, self)

    def param_view(self):
        p, n = C.c_void_p(), C.c_int64()
        self._ck(lib().ntf_param_buffer(self._h, C.byref(p), C.byref(n)))
        return DeviceView(p.value, n.value, self)

    def grad_tensor(self):
        """torch tensor aliasing the flat gradient buffer in HBM (what RCCL all-reduces)."""
        import torch
        return torch.as_tensor(self.grad_view(), device=f"cuda:{torch.cuda.current_device()}")

    def synchronize(self):
        self._ck(lib().ntf_synchronize(self._h))

    def kernel_times(self, enable=True):
        """enable: False / True (every kernel family) / 2 (only the output layer's forward and dW kernels) / 3 (only its forward kernel) / 4 (only its dW kernel)"""
        cap = 32
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); calls = (C.c_int64 * cap)()
        n = lib().ntf_kernel_times(self._h, int(enable), names, ms, calls, cap)
        return {names[i].decode(): (ms[i], calls[i]) for i in range(min(n, cap))}


class Node2Vec:
    """torch_geometric.nn.Node2Vec (p = q = 1) resident on one MI355X: embedding.weight [num_nodes, d] + dense Adam (include/opentf_amd.h)."""

    def __init__(self, rowptr, col, init_weight, seed=0, device=0):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64); self.col = np.ascontiguousarray(col, dtype=np.int32)
        w = _f32(init_weight)
        self.num_nodes, self.d = w.shape
        if len(self.rowptr) != self.num_nodes + 1:
            raise NtfError("rowptr must have num_nodes + 1 entries")
        self._h = C.c_void_p()
        rc = lib().ntf_n2v_create(int(device), self.num_nodes, self.d, _ptr(self.rowptr), _ptr(self.col), _ptr(w), int(seed) & (2**64 - 1), C.byref(self._h))
        if rc != 0:
            msg = lib().ntf_n2v_last_error(None); self._h = None
            raise NtfError(f"ntf_n2v_create failed ({rc}): {msg.decode() if msg else ''}")

    def _ck(self, rc):
        if rc != 0:
            raise NtfError(f"libopentf_amd n2v error {rc}: {lib().ntf_n2v_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            lib().ntf_n2v_destroy(self._h); self._h = None

    def __del__(self):
        try: self.close()
        except Exception: pass

    def walks(self, start, walk_length, step=0):
        s = np.ascontiguousarray(start, dtype=np.int64)
        out = np.empty((len(s), int(walk_length)), dtype=np.int64)
        self._ck(lib().ntf_n2v_walks(self._h, _ptr(s), len(s), int(walk_length), int(step), _ptr(out)))
        return out

    def train_batch(self, batch, walk_length, context, walks_per_node, num_neg, lr, apply=True, want_loss=True):
        b = np.ascontiguousarray(batch, dtype=np.int64); loss = C.c_float()
        self._ck(lib().ntf_n2v_train_batch(self._h, _ptr(b), len(b), int(walk_length), int(context), int(walks_per_node), int(num_neg), float(lr),
                                           None, 0, None, 0, int(apply), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def loss_on(self, pos_rw, neg_rw, lr=0.0, apply=False):
        """loss (and gradient / Adam step) on GIVEN window rows [n, context]: Node2Vec.loss(pos_rw, neg_rw)"""
        p = np.ascontiguousarray(pos_rw, dtype=np.int64); n = np.ascontiguousarray(neg_rw, dtype=np.int64); loss = C.c_float()
        assert p.shape[1] == n.shape[1]
        self._ck(lib().ntf_n2v_train_batch(self._h, None, 0, max(p.shape[1], 2), p.shape[1], 1, 1, float(lr), _ptr(p), len(p), _ptr(n), len(n), int(apply), C.byref(loss)))
        return loss.value

    def weight(self):
        out = np.empty((self.num_nodes, self.d), dtype=np.float32)
        self._ck(lib().ntf_n2v_get(self._h, 0, _ptr(out))); return out

    def grad(self):
        out = np.empty((self.num_nodes, self.d), dtype=np.float32)
        self._ck(lib().ntf_n2v_get(self._h, 1, _ptr(out))); return out

    def edge_bce(self, src, dst):
        s = np.ascontiguousarray(src, dtype=np.int64); t = np.ascontiguousarray(dst, dtype=np.int64); out = C.c_float()
        self._ck(lib().ntf_n2v_edge_bce(self._h, _ptr(s), _ptr(t), len(s), C.byref(out))); return out.value


class Doc2Vec:
    """gensim.models.Doc2Vec's tables (doc vectors, word vectors, syn1neg) resident on one MI355X and its negative-sampling trainer (include/opentf_amd.h ntf_d2v_*).
    The documents come as CSR over vocabulary indices together with what build_vocab prepares (sample_int, cum_table) and the initial vectors."""

    DV, WV, SYN1NEG = 0, 1, 2

    def __init__(self, doc_ptr, words, sample_int, cum_table, init_wv, init_dv, seed=0, device=0):
        self.doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64); self.words = np.ascontiguousarray(words, dtype=np.int32)
        si = np.ascontiguousarray(sample_int, dtype=np.uint32); ct = np.ascontiguousarray(cum_table, dtype=np.uint32)
        wv, dv = _f32(init_wv), _f32(init_dv)
        self.n_docs, self.d = dv.shape
        self.n_vocab = wv.shape[0]
        if len(self.doc_ptr) != self.n_docs + 1 or len(si) != self.n_vocab or len(ct) != self.n_vocab or wv.shape[1] != self.d:
            raise NtfError("doc_ptr / sample_int / cum_table / initial vectors do not fit together")
        self._h = C.c_void_p()
        rc = lib().ntf_d2v_create(int(device), self.n_docs, self.n_vocab, self.d, _ptr(self.doc_ptr), _ptr(self.words), _ptr(si), _ptr(ct), _ptr(wv), _ptr(dv),
                                  int(seed) & (2**64 - 1), C.byref(self._h))
        if rc != 0:
            msg = lib().ntf_d2v_last_error(None); self._h = None
            raise NtfError(f"ntf_d2v_create failed ({rc}): {msg.decode() if msg else ''}")

    def _ck(self, rc):
        if rc != 0:
            raise NtfError(f"libopentf_amd d2v error {rc}: {lib().ntf_d2v_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            lib().ntf_d2v_destroy(self._h); self._h = None

    def __del__(self):
        try: self.close()
        except Exception: pass

    def train_epoch(self, dm, window, alpha_start, alpha_end, epoch, negative=5, serial=False, order=None, progress=None, want_loss=False, want_ms=False):
        """one pass over the documents (gensim train(epochs=1)); returns (mean loss | None, device ms | None)"""
        o = None if order is None else np.ascontiguousarray(order, dtype=np.int64)
        pr = None if progress is None else np.ascontiguousarray(progress, dtype=np.float64)
        loss, ms = C.c_double(), C.c_double()
        self._ck(lib().ntf_d2v_train_epoch(self._h, int(dm), int(window), int(negative), float(alpha_start), float(alpha_end), int(epoch), int(bool(serial)),
                                           _ptr(o) if o is not None else None, _ptr(pr) if pr is not None else None, C.byref(loss) if want_loss else None,
                                           C.byref(ms) if want_ms else None))
        return (loss.value if want_loss else None), (ms.value if want_ms else None)

    def vectors(self, what=0):
        out = np.empty((self.n_docs if what == 0 else self.n_vocab, self.d), dtype=np.float32)
        self._ck(lib().ntf_d2v_get(self._h, int(what), _ptr(out))); return out

    def set_vectors(self, what, values):
        v = _f32(values)
        assert v.shape == ((self.n_docs if what == 0 else self.n_vocab), self.d)
        self._ck(lib().ntf_d2v_set(self._h, int(what), _ptr(v)))
