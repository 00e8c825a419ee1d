#!/usr/bin/env python3
"""teams/sec (train) of the Bnn `bnn_emb d=128` minibatch step on dblp-shaped data, on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one minibatch (B = 1000 teams per GPU) through the whole hot path of src/mdl/fnn.py:118-140:
CSR mean-pool gather of the skill embeddings, Flipout MLP forward, sparse-label weighted BCE with uniform
negative sampling, KL term, backward, (gradient all-reduce,) Adam.  Inputs (CSR matrices, embedding table,
weights) are resident in HBM before the timed region; nothing is skipped inside it.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant kernel (HIP
events recorded on the engine's stream during the timed region; the other output-layer kernel under
`roofline_other`) and `cpu_baseline` (the oracle's reference-shaped dense step, timed on this box's host cores,
N=1 only).  Timed regions of --steps steps are repeated until 2 s have been timed: `ms_per_step` is the median region / steps,
`ms_per_step_spread` = (max - min) / median over the regions.

N > 1 (`--parallel auto`, the default) times, in the same processes, the three ways a node shares a step:
  headline `value`  data parallel, b = 1000 teams per GPU (weak scaling) - rows split over the GPUs, gradients reduce-scattered / parameters
                    all-gathered over RCCL (opentf_amd/dp.py): what BASELINE.json's north_star names;
  `ep_weak`         the output layer split along the expert axis, every GPU steps the global minibatch of 1000 N teams (opentf_amd/ep.py);
  `strong_b1000`    what the plugin runs under torchrun: the reference's global minibatch of cfg.b = 1000 teams (src/mdl/__config__.yaml),
                    shared in the form opentf_amd/mdl/fnn.py::_parallel_mode picks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6 # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 / _f16 = 1024 FLOP/clk/SIMD = 16 x the f32 MFMA, dense (no sparsity)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dataset", default="dblp")
    ap.add_argument("--batch", type=int, default=1000, help="teams per GPU per step (cfg.b)")
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--model", default="bnn", choices=["bnn", "fnn"])
    ap.add_argument("--nsd", default="uniform")
    ap.add_argument("--input", default="meanpool", choices=["meanpool", "multihot"],
                    help="meanpool: team2vec table rows averaged over the team's skills (config 2); multihot: the 0/1 skill row itself, D=S (config 3)")
    ap.add_argument("--rows", type=int, default=0, help="override the number of teams (debug)")
    ap.add_argument("--experts", type=int, default=0, help="override the number of experts (debug)")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--fuse-adam", type=int, default=1, help="N=1 only. 1 (default): the output layer's Adam runs in the dW kernel's epilogue (52 instead of 76 B of HBM traffic per mu/rho pair, no gradient round trip); 0: one flat Adam kernel after backward; 2: dW in chunks, Adam of a finished chunk on a side stream")
    ap.add_argument("--mfma", default="default", choices=["default", "f32", "fp16x3"], help="arithmetic of the fused output-layer products (include/opentf_amd.h ntf_mfma)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-bench", action="store_true", help="(default at N=1) also time the whole-dataset gather (get_dense_vecs)")
    ap.add_argument("--no-gather-bench", action="store_true")
    ap.add_argument("--gather-only", action="store_true", help="run only the whole-dataset gather launches (the PMC passes of the gather roofline)")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the short exact-f32 (--mfma f32) measurement printed beside the default line")
    ap.add_argument("--force-dist", action="store_true", help="init RCCL and all-reduce the gradient buffer even at world_size 1 (validation)")
    ap.add_argument("--validate-on-one-gpu", action="store_true",
                    help="N > 1 launch whose ranks ALL drive cuda:0 and talk over gloo (RCCL refuses two ranks on one device): exercises the multi-rank code path of this "
                         "script on a 1-GPU box; the number it prints is not a measurement")
    ap.add_argument("--parallel", default="auto", choices=["auto", "ep", "dp"],
                    help="N > 1: ep = expert-sharded output layer (every GPU steps the whole global minibatch on its 1/N of the experts; the only exchange is "
                         "d(hidden), opentf_amd/ep.py); dp = rows split over GPUs, gradients reduce-scattered (opentf_amd/dp.py); auto = ep when the model shards")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the short dblp_full (unfiltered matrix, M = 5 022 955) measurement printed under extra_configs at N = 1")
    ap.add_argument("--ep-emulate", type=int, default=0, metavar="G",
                    help="N = 1 only: run what ONE rank of G runs under --parallel ep (its 1/G of the experts, a global minibatch of G * --batch teams, two-phase "
                         "step, no exchange) - the per-rank compute time behind the scaling projection in DESIGN.md; not the headline")
    ap.add_argument("--dp-emulate", type=int, default=0, metavar="G",
                    help="N = 1 only: run what ONE rank of G runs under --parallel dp (north_star's form): its --batch rows of a global minibatch of G * --batch teams through the "
                         "deferred, chunked dW path (no Adam in the dW epilogue, stand-alone operand producer), Adam on the 1/G shard it would own - WITHOUT the reduce-scatter / "
                         "all-gather (their bytes are reported): the per-rank compute time of the data-parallel form; not the headline")
    return ap.parse_args()


def cpu_baseline(ds, dims, bayesian, cfg, sample_rows=1000, steps=10):
    """The oracle's reference-shaped step (dense [B, M] labels, rand_like+topk negatives, autograd, Adam) on the host."""
    import scipy.sparse
    import torch
    from oracle import ntf_oracle as O
    from opentf_amd.synth import init_params
    from collections import OrderedDict
    ncpu = os.cpu_count() or 1
    sd = OrderedDict((k, torch.from_numpy(v.copy())) for k, v in init_params(dims, bayesian, 0).items())
    opt = O.Adam(sd, cfg["lr"])
    m_ip, m_ix = ds["member"]
    member = scipy.sparse.csr_matrix((np.ones(len(m_ix), np.uint8), m_ix, m_ip), shape=(ds["N"], ds["M"]))
    s_ip, s_ix = ds["skill"]
    rng = np.random.default_rng(1)
    # torch's intra-op pool does not scale to every hardware thread on these elementwise-heavy [B, M] ops: pick the
    # fastest of a few thread counts on a short probe, then time the sample with it (the count used is reported as `cores`)
    best, cores = None, 1
    for n in sorted({min(ncpu, t) for t in (16, 32, 64)}):      # (round 5: three candidates instead of six - 16-32 won on every box of the pool; the probe is host time in front of the JSON line)
        torch.set_num_threads(n)
        O.reference_shaped_step(sd, opt, ds["table"], s_ip, s_ix, rng.integers(0, ds["N"], 32), member, cfg)  # warm
        t0 = time.perf_counter()
        O.reference_shaped_step(sd, opt, ds["table"], s_ip, s_ix, rng.integers(0, ds["N"], 64), member, cfg)
        dt = time.perf_counter() - t0
        if best is None or dt < best: best, cores = dt, n
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.reference_shaped_step(sd, opt, ds["table"], s_ip, s_ix, rng.integers(0, ds["N"], sample_rows), member, cfg)
    dt = time.perf_counter() - t0
    model, phys = "", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model: model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"): pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"): cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None: phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return {"value": steps * sample_rows / dt, "unit": "teams/s", "cores": cores, "threads_used": cores, "physical_cores": len(phys) or None, "logical_cpus": ncpu, "cpu_model": model, "kind": "port",
            "sample": f"{steps} steps of B={sample_rows} at full M={ds['M']} (oracle/ntf_oracle.py reference_shaped_step, torch {torch.__version__} CPU)",
            "note": "cores = threads_used = the torch thread count a short probe found fastest; physical_cores = distinct (physical id, core id) pairs of /proc/cpuinfo, logical_cpus = os.cpu_count().  Calibration (tests/golden/calibrate_cpu_port.py, build container, 8 threads, "
                    "Fnn, M = 20 000, B = 1000): this port runs 1.62 x the rate of the imported reference Fnn.learn (1696 teams/s) - it skips the per-team NtfDataset.__getitem__ + collate "
                    "of src/mdl/ntf.py:22-24; BASELINE.md section 3"}


def launched_kernel(family, a):
    """name (prefix) of the output-layer kernel a default-shaped run launches for a timed family - what a committed PMC file must be about before its bytes are quoted"""
    if a.no_fused or a.hidden != 128: return None
    if a.mfma == "f32": return {"out_fused_fwd_loss_dh": "k_out_fwd<", "out_fused_dw_adam": "k_out_dw<"}.get(family)
    fwd = {None: "k_out_fwd_h3p", "5": "k_out_fwd_h3p", "0": "k_out_fwd_b6"}.get(os.environ.get("NTF_FWD_KERNEL"))
    dw = {None: "k_out_dw_q", "1": "k_out_dw_q"}.get(os.environ.get("NTF_DW_KERNEL"))
    return {"out_fused_fwd_loss_dh": fwd, "out_fused_dw_adam": dw}.get(family)


def pmc_traffic(family, a, ds):
    """HBM bytes per launch of an output-layer kernel from the committed rocprofv3 PMC passes (profiles/, collected and corrected
    as MI355X_MICROARCH.md prescribes: FETCH_SIZE doubled, separate passes); only when the run is the profiled configuration AND the
    file's kernel is the one this run launches (a PMC file goes stale silently when a kernel is replaced: then traffic is null)."""
    names = {"f32": ["r1_c_pmc_traffic_and_sq.json"]}.get(a.mfma, ["r6_pmc_traffic_and_sq.json", "r5_pmc_traffic_and_sq.json"])
    path = next((os.path.join(ROOT, "profiles", n) for n in names if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
    if not (path and a.dataset == "dblp" and a.model == "bnn" and a.batch == 1000 and a.d == 128 and a.hidden == 128
            and a.input == "meanpool" and not a.rows and not a.experts):
        return None, None
    key = launched_kernel(family, a)
    if not key:
        return None, None
    best = None
    for name, v in json.load(open(path))["kernels"].items():
        if key in name and "hbm_bytes" in v and (best is None or v["hbm_bytes"] > best): best = v["hbm_bytes"]   # (the no-op range-fallback kernels share the prefix)
    return (best, os.path.basename(path)) if best is not None else (None, None)


def workload_label(a, ds, bayesian, multihot):
    names = {"dblp": "dblp mt10.ts2 shapes", "dblp_full": "dblp UNFILTERED shapes (output/dblp/dblp.v12.json/prep.teamsvecs.log:18)", "uspt": "uspt mt10.ts2 shapes",
             "uspt_full": "uspt UNFILTERED shapes (output/uspt/patent.tsv/prep.teamsvecs.log:34)", "gith": "gith UNFILTERED shapes", "imdb": "imdb shapes"}
    return (f"{names.get(a.dataset, a.dataset)} N={ds['N']} S={ds['S']} M={ds['M']}; {a.model}{' (Flipout)' if bayesian else ''} on " +
            (f"multi-hot skill rows D={ds['S']}, " if multihot else f"mean-pooled skill table d={a.d}, ") +
            f"h=[{a.hidden}], b={a.batch}/GPU, ns=5 {a.nsd}, tpw 10 tnw 1, Adam lr 1e-3")


class Cfg(dict):
    """attribute-dict standing in for the omegaconf DictConfig the reference passes (module level: the plugin pickles it into its checkpoints)"""
    def __getattr__(self, k):
        if k.startswith("__"): raise AttributeError(k)
        return self.get(k)


class NoWriter:
    def __init__(self, log_dir=None): pass
    def add_scalar(self, **k): pass
    def close(self): pass


def plugin_epoch(ds, a, device, headline_value):
    """The DROP-IN timed, not only the engine (VERDICT r4 missing #4): one fold-epoch of `opentf_amd.mdl.bnn.Bnn.learn` - the replacement of src/mdl/fnn.py:78-170 that
    src/main.py:155-193 calls - at the headline's shapes, from lil `teamsvecs` as the reference's pipeline hands them over: ingestion (lil -> CSR, uploads), the train phase
    (loader order, staging, `b`-row steps), the validation phase and one checkpoint.  A sample of the teams (200 train + 100 validation batches): the rate does not
    depend on the number of batches, and building 2 M-row lil matrices would only be host time."""
    import shutil, tempfile, time as _t
    import scipy.sparse
    from opentf_amd import libntf
    from opentf_amd.mdl.bnn import Bnn

    b = a.batch
    n_tr, n_va = min(200 * b, ds["N"] * 2 // 3), min(100 * b, ds["N"] // 3)
    n = n_tr + n_va
    (s_ip, s_ix), (m_ip, m_ix) = ds["skill"], ds["member"]
    t0 = _t.perf_counter()
    skill = scipy.sparse.csr_matrix((np.ones(int(s_ip[n]), np.uint8), s_ix[: int(s_ip[n])], s_ip[: n + 1]), shape=(n, ds["S"])).tolil()
    member = scipy.sparse.csr_matrix((np.ones(int(m_ip[n]), np.uint8), m_ix[: int(m_ip[n])], m_ip[: n + 1]), shape=(n, ds["M"])).tolil()
    lil_s = _t.perf_counter() - t0
    tv = {"skill": skill, "member": member, "loc": None, "skill_table": ds["table"]}
    splits = {"test": np.empty(0, np.int64), "folds": {0: {"train": np.arange(n_tr), "valid": np.arange(n_tr, n)}}}
    cfg = Cfg(b=b, e=1, ns=5, nsd=a.nsd, lr=0.001, es=5, h=[a.hidden], spe=None, l="bce", tpw=10, tnw=1, nmc=10)
    out = tempfile.mkdtemp(prefix="ntf_plugin_epoch_")
    spans = {}
    from opentf_amd.mdl import fnn as fnn_mod
    real = {"tr": libntf.Engine.train_epoch, "ev": libntf.Engine.eval_epoch, "new": Bnn._new_engine, "save": Bnn._save, "order": fnn_mod.index_order,
            "init": Bnn.init, "load": libntf.Engine.load_state_dict, "close": libntf.Engine.close}

    def timed(key, fn):
        def w(*args, **kw):
            t = _t.perf_counter(); r = fn(*args, **kw); spans[key] = spans.get(key, 0.0) + _t.perf_counter() - t
            return r
        return w
    try:
        libntf.Engine.train_epoch = timed("train", real["tr"]); libntf.Engine.eval_epoch = timed("valid", real["ev"])
        Bnn._new_engine = timed("ingest", real["new"]); Bnn._save = timed("checkpoint", real["save"]); fnn_mod.index_order = timed("order", real["order"])
        Bnn.init = timed("init", real["init"]); libntf.Engine.load_state_dict = timed("load", real["load"]); libntf.Engine.close = timed("close", real["close"])
        m = Bnn(out, f"cuda:{device}", 0, cfg); m.writer = NoWriter
        t0 = _t.perf_counter(); m.learn(tv, splits, None); wall = _t.perf_counter() - t0
    finally:
        libntf.Engine.train_epoch, libntf.Engine.eval_epoch, Bnn._new_engine, Bnn._save, fnn_mod.index_order = real["tr"], real["ev"], real["new"], real["save"], real["order"]
        Bnn.init, libntf.Engine.load_state_dict, libntf.Engine.close = real["init"], real["load"], real["close"]
        shutil.rmtree(out, ignore_errors=True)
    tr_rate = n_tr / (spans["train"] + spans.get("order", 0.0))      # the loader's order (both phases') counted with the train phase
    return {"workload": f"opentf_amd.mdl.bnn.Bnn.learn, one fold x one epoch: {n_tr} train + {n_va} validation teams of the headline's dataset as lil teamsvecs + skill_table, b={b}, nsd={a.nsd}",
            "learn_wall_s": wall, "ingest_s": spans.get("ingest"), "train_phase_s": spans["train"], "loader_order_s": spans.get("order"), "valid_phase_s": spans.get("valid"), "checkpoint_s": spans.get("checkpoint"),
            "value_is": "train teams / (train_phase_s + loader_order_s): torch's DataLoader draws for the shuffled order, ntf_stage_order, ceil(n / b) ntf_step_staged calls, the epoch-loss read-back",
            "value": tr_rate, "unit": "teams/s", "ms_per_step": spans["train"] / (-(-n_tr // b)) * 1e3, "ratio_to_headline": tr_rate / headline_value if headline_value else None,
            "param_init_on_host_s": spans.get("init"), "param_upload_s": spans.get("load"), "engine_release_s": spans.get("close"),   # init: the reference's own torch CPU draws (same seed -> same weights)
            "rest_of_learn_s": wall - sum(v for v in spans.values()),
            "lil_build_s_not_plugin_time": lil_s}


class LegSkipped(RuntimeError):
    """an extra leg of an N > 1 run that the ranks agreed not to enter (some rank could not build its engine)"""


def rooflines(times, a, bayesian, eB, H, Mloc, ds, ep, evt_steps=None, adam_in_dw=True):
    """roofline objects of the two output-layer kernels from their HIP-event times in the timed region: (dominant, other).  A data-parallel rank launches the forward
    kernel range by range and the dW kernel chunk by chunk (`launches_per_step` > 1): the figures below are then a STEP's launches together - the whole layer's work over the
    sum of their times - and say so."""
    gemm = 2.0 * eB * H * Mloc  # one [rows,H]x[H,experts]-sized product of a launch
    k = 2 if bayesian else 1
    # per timed scope: the unfused families launch one GEMM per Flipout half (k launches), the fused ones a single kernel
    flops_per_launch = {"out_fwd_gemm": k * gemm, "out_bwd_dw_gemm": k * gemm, "out_bwd_da_gemm": k * gemm,
                        "out_fused_fwd_loss_dh": 2 * k * gemm, "out_fused_dw_adam": k * gemm}
    cand = {f: times[f] for f in flops_per_launch if f in times and times[f][1] > 0}
    out = []
    for fam in sorted(cand, key=lambda f: -cand[f][0]):
        ms, calls = cand[fam]
        lps = max(1, int(round(calls / evt_steps[fam]))) if (evt_steps and evt_steps.get(fam)) else 1
        t = (ms / evt_steps[fam] if lps > 1 else ms / calls) * 1e-3
        part = {"launches_per_step": lps, "ms_per_step_all_launches": t * 1e3,
                "note": f"{lps} launches of this family a step (ranges / chunks of the layer, and the conditional exact-f32 launch where there is one): achieved = the layer's work / the sum of their times; avg_ms is one launch"} if lps > 1 else {}
        traffic, src = (None, None) if ep else pmc_traffic(fam, a, ds)
        fused_adam = fam == "out_fused_dw_adam" and a.fuse_adam == 1 and adam_in_dw and not a.no_fused and a.mfma != "f32"
        if fused_adam:
            # dW + Adam (+ the next step's operands) in one kernel: HBM-bound.  Algorithmic bytes: the packed dz read once (4 B per row and expert of
            # the padded [256-expert tile, 128-row block] grid) + per weight 24 B read (mu, rho, four Adam moments) + 24 B written (mu, rho, moments) + 8 B of
            # next-step operands (the fp16 planes of sigma*eps' and of mu') for Flipout; 12 + 12 B for Fnn.  (Round 3: + 4 B read and 4 B written for an f32 copy of sigma*eps.)
            Bpad = (eB + 127) // 128 * 128; Mpad = (Mloc + 255) // 256 * 256
            lean = os.environ.get("NTF_LEAN", "1") != "0"      # round 4: no f32 copy of sigma * eps is written or read (eps drawn again in the epilogue)
            per_w = (56 if lean else 64) if bayesian else 24
            nbytes = 4.0 * Bpad * Mpad + per_w * H * Mloc
            out.append({"bound": "hbm", "kernel": fam, "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": src and f"per-launch mean of the separate rocprofv3 --pmc passes committed as profiles/{src} (FETCH_SIZE doubled per MI355X_MICROARCH.md), not counted in this run",
                        "avg_ms": ms / calls, "launches": calls, "bytes_per_launch": nbytes, "bytes_def": f"4 B x {Bpad} x {Mpad} (packed dz) + {per_w} B x {H} x {Mloc} (Adam in place + next-step operands)",
                        "mfma_tflops_algorithmic": flops_per_launch[fam] / t / 1e12, **part})
            continue
        ach = flops_per_launch[fam] / t / 1e12
        # arithmetic of the kernel: fp16x3 = every f32 operand (times an exact power of two) split into 2 fp16 values, a product = 3 fp16 MFMA products
        # accumulated in f32.  The roof for ALGORITHMIC flops is then the dense fp16 MFMA peak / 3.
        split = a.mfma != "f32" and not a.no_fused and ((fam == "out_fused_dw_adam") or (fam == "out_fused_fwd_loss_dh" and a.hidden == 128))
        nprod = 3 if split else 1
        peak = BF16_MFMA_PEAK_TFLOPS / nprod if split else F32_MFMA_PEAK_TFLOPS
        arith = {1: "f32 MFMA (v_mfma_f32_32x32x2_f32)",
                 3: "fp16x3: operands * 2^k split into 2 fp16 values (22 bits), 3 fp16 MFMA products per f32 product, f32 accumulate; peak = 2516.6 / 3"}[nprod]
        out.append({"bound": "mfma", "kernel": fam, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                    "traffic_source": src and f"per-launch mean of the separate rocprofv3 --pmc passes committed as profiles/{src} (FETCH_SIZE doubled per MI355X_MICROARCH.md), not counted in this run",
                    "avg_ms": ms / calls, "launches": calls, "flops_per_launch": flops_per_launch[fam], "arithmetic": arith, "hw_mfma_tflops": ach * nprod,
                    "power_note": "peak is the nominal 2.4 GHz figure; at the 1400 W package limit a dense fp16 MFMA stream on changing random operands sustains 1.595 GHz = 0.66 of it (profiles/r3_power_and_clocks.md, measured once, not in this run)" if nprod > 1 else None, **part})
    return (out[0] if out else None), (out[1] if len(out) > 1 else None)


def validation_step(ev, a, bayesian, eB, H, M):
    """the evaluation step against the MFMA roof: one (Fnn) or two (Flipout) [rows, H] x [H, experts] products, forward only"""
    if not ev: return None
    flop = (2 if bayesian else 1) * 2.0 * eB * H * M
    split = a.mfma != "f32" and not a.no_fused and a.hidden == 128
    peak = BF16_MFMA_PEAK_TFLOPS / 3 if split else F32_MFMA_PEAK_TFLOPS
    t = ev["ms_per_step"] * 1e-3
    return {**ev, "flop_per_step": flop, "tflops": flop / t / 1e12, "mfma_peak": peak, "mfma_frac": flop / t / 1e12 / peak,
            "note": "whole evaluation step (operand producer, head, forward + loss kernel, fix-up), timed over an eval_epoch call; kernel: k_out_fwd_h3e unless NTF_EVAL_KERNEL=0"}


def step_roofline(a, bayesian, head, ds, multihot, step_s):
    """The WHOLE step against both roofs (the per-kernel `roofline` objects leave out what sits between the kernels): SURVEY 8d's algorithmic FLOPs per team
    (Bnn 12 H M + 8 D H, Fnn 6 (D H + H M); D = the dense input width, the multi-hot first layer counted as its gather: 2 nnz H per product) x the rows one
    engine steps, and the step's compulsory HBM bytes (split planes read + packed dz written and read + Adam in place + next-step operands), each over the
    measured ms_per_step."""
    eB, H, M = head["eB"], a.hidden, head["Mloc"]
    D = (ds["skill"][0][-1] / ds["N"]) if multihot else a.d
    flop = eB * ((12.0 * H * M + 8.0 * D * H) if bayesian else 6.0 * (D * H + H * M))
    split = a.mfma != "f32" and not a.no_fused and a.hidden == 128
    peak = BF16_MFMA_PEAK_TFLOPS / 3 if split else F32_MFMA_PEAK_TFLOPS
    Bpad = (eB + 127) // 128 * 128; Mpad = (M + 255) // 256 * 256
    k = 2 if bayesian else 1
    fwd_b = 4.0 * k * H * M + 4.0 * Bpad * Mpad                        # two fp16 planes of mu (and of sigma * eps) read once; the packed dz written
    dw_b = 4.0 * Bpad * Mpad + ((56 if os.environ.get("NTF_LEAN", "1") != "0" else 64) if bayesian else 24) * H * M
    nbytes = fwd_b + dw_b
    return {"flop_per_step": flop, "tflops": flop / step_s / 1e12, "mfma_peak": peak, "mfma_frac": flop / step_s / 1e12 / peak,
            "bytes_per_step": nbytes, "gbs": nbytes / step_s / 1e9, "hbm_peak": HBM_PEAK_GBS, "hbm_frac": nbytes / step_s / 1e9 / HBM_PEAK_GBS,
            "note": "whole step: algorithmic FLOPs (SURVEY 8d) and the output layer's compulsory HBM bytes over ms_per_step; the hidden layer's and the head's bytes (< 1 %) are left out"}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # stdout must carry exactly ONE JSON line: RCCL writes a version banner to the C-level stdout (flushed when the process exits, i.e. after
    # the JSON line), so everything but that line is sent to stderr at the file-descriptor level
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush(); os.dup2(2, 1)
    if a.validate_on_one_gpu: local = 0
    torch.cuda.set_device(local)
    if world > 1 or a.force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ["NTF_DP_FORCE_ALLREDUCE"] = "1"; os.environ["NTF_EP_FORCE_EXCHANGE"] = "1"
        import datetime
        nccl_to = datetime.timedelta(seconds=float(os.environ.get("NTF_BENCH_COLLECTIVE_TIMEOUT_S", "180")))
        if a.validate_on_one_gpu: dist.init_process_group("gloo", timeout=nccl_to)
        else: dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"), timeout=nccl_to)
        os.environ.setdefault("NTF_COLLECTIVE_TIMEOUT_S", "120")      # opentf_amd/dp.py, ep.py: bounded waits that name the collective they are stuck behind
    # control plane of the extra legs (N > 1): a gloo group of its own - "did every rank get through this leg" must stay answerable when RCCL is not
    ctl = None
    if world > 1:
        import datetime
        ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=300))

    def all_ok(ok):
        """AND over the ranks of `ok`; a control plane that cannot be reached counts as a failure"""
        if world == 1: return bool(ok)
        t = torch.tensor([0 if ok else 1], dtype=torch.int32)
        try:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
            return int(t.item()) == 0
        except Exception:
            return False

    def inject_failure(leg, stage):
        """NTF_BENCH_FAIL_LEG=<leg>:<build|run>[:rank] (tests): a Python exception inside that leg"""
        spec = os.environ.get("NTF_BENCH_FAIL_LEG", "").split(":")
        if len(spec) >= 2 and spec[0] == leg and spec[1] == stage and (len(spec) < 3 or int(spec[2]) == rank):
            raise RuntimeError(f"injected failure in leg {leg} ({stage}) on rank {rank}")

    from opentf_amd import libntf
    from opentf_amd.dp import DataParallel
    from opentf_amd.ep import ExpertParallel, expert_shards, can_shard
    from opentf_amd.synth import make_dataset, init_params

    bayesian = a.model == "bnn"
    ds = make_dataset(a.dataset, d=a.d, seed=0, n_rows=a.rows or None, n_experts=a.experts or None)
    multihot = a.input == "multihot"
    dims = [ds["S"] if multihot else a.d, a.hidden, ds["M"]]
    cfg = {"ns": 5, "nsd": a.nsd, "tpw": 10.0, "tnw": 1.0, "lr": 1e-3}
    if (a.ep_emulate or a.dp_emulate) and world > 1: raise SystemExit("--ep-emulate / --dp-emulate are single-GPU measurements")
    if a.ep_emulate and a.dp_emulate: raise SystemExit("--ep-emulate and --dp-emulate exclude each other")
    G = a.ep_emulate or a.dp_emulate or world
    shardable = (not a.no_fused) and can_shard(dims, G)
    if a.parallel == "ep" and (G > 1 or a.force_dist) and not shardable:
        raise SystemExit(f"--parallel ep: {dims} does not shard over {G} GPUs (needs h[-1] in 32/64/128 and >= {G} tiles of 256 experts)")
    stream = torch.cuda.Stream()
    sd0 = init_params(dims, bayesian, 0)
    n_params = int(sum(v.size for v in sd0.values()))
    rng = np.random.default_rng(7)

    def run_mode(par, gB, steps, warmup, reps_allowed=True, breakdown=True, data=ds, model_dims=dims, params=sd0, leg=None, agree=None):
        """build an engine for this way of sharing a step (par = dp | ep, gB = the global minibatch), warm up, time `steps` steps between barriers; returns a dict.
        agree (extra legs at N > 1): called with this rank's "my engine is built" - the leg goes on only when every rank's is (no rank enters a collective alone)"""
        ep = par == "ep"
        shard = expert_shards(model_dims[-1], G)[0 if a.ep_emulate else rank] if ep else None
        eB = gB if ep else -(-gB // (a.dp_emulate or (1 if a.ep_emulate else world)))     # rows one engine steps: under ep every rank steps the whole global minibatch
        with torch.cuda.stream(stream):
            e, built_err = None, None
            try:
                if leg: inject_failure(leg, "build")
                e = libntf.Engine(model_dims, bayesian=bayesian, input_mode=libntf.INPUT_MULTIHOT if multihot else libntf.INPUT_MEANPOOL, max_batch=eB, ns=5, nsd=a.nsd, tpw=10.0, tnw=1.0,
                                  lr=1e-3, seed=1234, device=local, stream=stream.cuda_stream, fused=not a.no_fused,
                                  fuse_adam=a.fuse_adam if ((world == 1 and not a.dp_emulate) or ep) else 0, mfma=a.mfma, expert_shard=shard, ep_world=G if ep else 1)
                if not multihot: e.set_skill_table(data["table"])
                e.set_skill_csr(data["skill"]); e.set_member(data["member"])
                e.load_state_dict(params)
                if a.nsd == "unigram":   # expert frequency over the training rows (src/mdl/fnn.py:97)
                    e.set_unigram(np.bincount(data["member"][1], minlength=data["M"]) / data["N"])
                dp = ExpertParallel(e, two_phase=bool(a.ep_emulate)) if ep else DataParallel(e, emulate_world=a.dp_emulate)
            except Exception as ex:
                if agree is None: raise
                built_err = f"{type(ex).__name__}: {ex}"
            if agree is not None and not agree(built_err is None):
                if e is not None: e.close()
                raise LegSkipped(built_err or "another rank could not build its engine")
            if leg: inject_failure(leg, "run")
            # timed regions of `steps` steps are repeated until >= 2 s of GPU time have been timed (the driver's --steps 20 is a 30 ms region: 5 of them were 0.15 s
            # of a 90 s process - too short for an outside sampler to see the GPU busy); ms_per_step = the median region / steps
            reps_max = 96 if reps_allowed else 1
            order = rng.integers(0, data["N"], (warmup + steps * reps_max) * gB).astype(np.int64)   # the loader's shuffled row order
            if warmup: dp.train_epoch(order[: warmup * gB], gB)
            # HIP events inside the timed regions, around the roofline's kernels only - and around ONE of the two per region, alternating (an event pair costs the step
            # ~8 us: with both pairs in every region the headline carried 17 us = 1.2 % of measurement overhead; each kernel is still timed live, in every second region)
            no_events = bool(os.environ.get("NTF_BENCH_NO_EVENTS"))
            times = {}

            evt_steps = {}      # steps whose launches of a family carried events: a chunked / ranged step (data parallel) launches a kernel several times per step

            def collect(next_mode, nsteps=0):
                for fam, (ms, calls) in e.kernel_times(enable=next_mode).items():
                    if calls > 0:
                        times[fam] = (times.get(fam, (0.0, 0))[0] + ms, times.get(fam, (0.0, 0))[1] + calls)
                        evt_steps[fam] = evt_steps.get(fam, 0) + nsteps
            e.kernel_times(enable=0 if no_events else 3)
            regions, mean_loss, off, losses = [], None, warmup * gB, []
            while len(regions) < reps_max:
                e.synchronize(); torch.cuda.synchronize()
                if world > 1: dist.barrier()
                t0 = time.perf_counter()
                mean_loss = dp.train_epoch(order[off: off + steps * gB], gB)
                e.synchronize(); torch.cuda.synchronize()
                if world > 1: dist.barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
                if world > 1: dist.all_reduce(t, op=dist.ReduceOp.MAX)      # every rank sees the same region time, hence takes the same decision below
                regions.append(float(t.item())); off += steps * gB; losses.append(float(mean_loss))
                collect(0 if no_events else (4 if len(regions) % 2 else 3), steps)      # (a single region: the forward kernel only - its partner comes from the breakdown pass below)
                if sum(regions) >= float(os.environ.get("NTF_BENCH_MIN_TIMED_S", "2.0")): break                                # every rank sees the same (max-reduced) times, hence takes the same decision
            collect(0)
            bd, k3 = None, 0
            if breakdown:   # per-family breakdown from a SEPARATE short pass (events around every family perturb the step by a few per cent)
                k3 = max(5, min(10, steps))
                e.kernel_times(enable=True)
                dp.train_epoch(order[: k3 * gB], gB); e.synchronize()
                full = e.kernel_times(enable=False)
                bd = {f: round(v[0] / k3, 4) for f, v in full.items() if v[1] > 0}
                for fam in ("out_fused_fwd_loss_dh", "out_fused_dw_adam", "out_fwd_gemm", "out_bwd_dw_gemm"):      # a run of ONE timed region saw one of the two kernels only
                    if fam not in times and fam in full and full[fam][1] > 0: times[fam] = full[fam]; evt_steps[fam] = k3
            # N > 1 (or --force-dist): what the collectives cost this rank, from a SEPARATE short pass with an event pair around every stream-ordered wait (dp.CollectiveTrace.
            # exposed_waits: the stream does nothing between the pair but wait) - exposed wait per collective class and, by difference, the rank's own compute per step
            waits = None
            tr = getattr(dp, "trace", None)
            if (world > 1 or a.force_dist) and tr is not None and getattr(tr, "stream_ordered", False):
                kw = max(5, min(20, steps))
                tr.measure_waits = True
                e.synchronize(); torch.cuda.synchronize()
                if world > 1: dist.barrier()
                t0 = time.perf_counter(); dp.train_epoch(order[: kw * gB], gB); e.synchronize(); torch.cuda.synchronize()
                tw = (time.perf_counter() - t0) / kw * 1e3
                ex = {c: v / kw for c, v in tr.exposed_waits().items()}
                tr.measure_waits = False
                waits = {"steps": kw, "ms_per_step_of_this_pass": tw, "exposed_wait_ms_per_step": ex, "rank_compute_ms_per_step": tw - sum(ex.values()),
                         "note": "rank 0's figures; the pass carries two events per collective wait (not the timed regions)"}
            dt = float(np.median(regions))
            # the validation phase's step (src/mdl/fnn.py:143-151: forward + loss on fresh eps, no backward): 566 of a fold-epoch's 1 697 batches at config 2
            ev = None
            if breakdown and world == 1 and not a.ep_emulate and not a.dp_emulate:
                kev = max(10, min(40, steps))
                ev_run = e.eval_epoch if type(dp).__name__ == "DataParallel" else dp.eval_epoch      # (one GPU: ntf_eval_epoch, what the plugin's validation phase calls - its steps share the KL term and prefetch each other's operands)
                ev_run(order[: 3 * gB], gB); e.synchronize()
                t0 = time.perf_counter(); ev_loss = ev_run(order[: kev * gB], gB); e.synchronize()
                ev = {"ms_per_step": (time.perf_counter() - t0) / kev * 1e3, "steps": kev, "mean_loss": ev_loss}
            res = {"par": par, "eval": ev, "collective_waits": waits, "ep": ep, "gB": gB, "eB": eB, "dt": dt, "regions": regions, "mean_loss": mean_loss, "region_losses": losses, "times": times, "evt_steps": evt_steps, "adam_in_dw": bool((world == 1 and not a.dp_emulate) or ep), "breakdown": bd, "k3": k3,
                   "Mloc": (shard[1] - shard[0]) if ep else model_dims[-1], "engine": e,
                   "rccl_payload_bytes_per_step": (4 * gB * a.hidden) if ep else (8 * n_params if world > 1 else 0),
                   "emulated_bytes": getattr(dp, "emulated_bytes", None)}
            return res

    # ---- the headline mode
    if world > 1 and a.parallel == "auto": head_par = "dp"                 # north_star: data parallel, gradients exchanged over RCCL
    elif a.dp_emulate: head_par = "dp"
    elif (G > 1 or a.force_dist) and a.parallel != "dp" and shardable: head_par = "ep"
    else: head_par = "dp"
    gB = a.batch * G                                       # weak scaling: B teams per GPU
    head = run_mode(head_par, gB, a.steps, a.warmup)
    e = head["engine"]
    if a.gather_only:
        with torch.cuda.stream(stream):
            for _ in range(5): e.gather_meanpool(n=ds["N"], to_host=False)
            e.synchronize()
        print(json.dumps({"gather_only": True, "teams": ds["N"]}), file=real_stdout, flush=True)
        return

    gather = None
    with torch.cuda.stream(stream):
        if (a.gather_bench or (world == 1 and not a.no_gather_bench and not a.ep_emulate and not a.dp_emulate)) and rank == 0 and not multihot:
            e.kernel_times(enable=True)
            n = ds["N"]
            e.gather_meanpool(n=n, to_host=False); e.kernel_times(enable=True)
            for _ in range(3): e.gather_meanpool(n=n, to_host=False)
            ms, calls = e.kernel_times(enable=False)["gather"]
            nnz = ds["skill"][0][-1] / n
            bytes_per_team = nnz * (4 * a.d + 4) + 8 + 4 * a.d
            t = ms / calls * 1e-3
            gather = {"bound": "hbm", "achieved_algorithmic": bytes_per_team * n / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "bytes_per_team": bytes_per_team, "teams": n, "ms": ms / calls}
            pm = next((os.path.join(ROOT, "profiles", f) for f in ("r5_pmc_gather.json", "r4_pmc_gather.json", "r3_pmc_gather.json", "r2_pmc_gather.json") if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
            if pm:
                for name, v in json.load(open(pm))["kernels"].items():
                    if "k_gather_pool" in name and "hbm_bytes" in v: gather["traffic"] = v["hbm_bytes"]; gather["traffic_source"] = "profiles/" + os.path.basename(pm)
            # the algorithmic bytes count every gathered table row, but the 46 MB table is served on-die (Infinity Cache / L2): what crosses the HBM interface is the
            # counter's figure (PMC) - the fraction of the HBM peak is stated on THAT; the compulsory traffic is the output rows + the CSR (+ the table once)
            comp = n * (4 * a.d + 8 + nnz * 4) + ds["S"] * a.d * 4
            gather["hbm_compulsory_gbs"] = comp / t / 1e9
            if "traffic" in gather:
                gather["achieved"] = gather["traffic"] / t / 1e9; gather["frac"] = gather["achieved"] / HBM_PEAK_GBS
            else:
                gather["achieved"] = gather["hbm_compulsory_gbs"]; gather["frac"] = gather["achieved"] / HBM_PEAK_GBS
            gather["note"] = ("achieved = HBM bytes of the launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, committed pass) / time; achieved_algorithmic = SURVEY 8d's nnz*(4d+4)+8+4d per team / time "
                              "exceeds the HBM peak because ~80 % of the table-row reads are served by the XCDs' L2s and the Infinity Cache (MI355X_MICROARCH.md, Indexed rows: 8.6 TB/s on-die)")
        e.close()

    # ---- the JSON line (rank 0), assembled from the headline BEFORE anything else runs: whatever happens in a later leg, this much is printed
    extra_modes = {}
    state = {"exact_f32": None, "extra_configs": None, "cpu_baseline": None, "printed": False, "loss_finite": True}
    import threading
    emit_lock = threading.Lock()      # the watchdog thread and the main thread may both reach emit(): exactly one line is printed

    def emit(final):
        if rank != 0: return
        with emit_lock:
            if state["printed"]: return
            state["printed"] = True
        _emit(final)

    def _emit(final):
        B, H, M = a.batch, a.hidden, ds["M"]
        ep = head["ep"]; dt = head["dt"]
        roof, roof_other = rooflines(head["times"], a, bayesian, head["eB"], H, head["Mloc"], ds, ep, head.get("evt_steps"), head.get("adam_in_dw", True))
        spread = (max(head["regions"]) - min(head["regions"])) / dt if len(head["regions"]) > 1 else None
        state["loss_finite"] = bool(np.all(np.isfinite(head["region_losses"]))) and all(
            np.isfinite(v.get("mean_loss", 0.0) or 0.0) for v in list((state["extra_configs"] or {}).values()) + list(extra_modes.values()) if isinstance(v, dict))
        devices = [torch.cuda.get_device_name(local)]
        out = {
            "metric": "teams/sec (train) bnn_emb d=128 on DBLP", "value": a.steps * gB / dt, "unit": "teams/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "ms_per_step_spread": spread, "timed_regions": len(head["regions"]),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f32": "f32"}.get(a.mfma, "f32 (fp16x3 split products, f32 accumulate)"), "data": "synthetic",
            "config": {"workload": workload_label(a, ds, bayesian, multihot), "global_batch": gB,
                       "parallelism": (f"ep{world}: expert-sharded output layer, every GPU steps the global minibatch on 1/{world} of the experts, d(hidden) all-reduced" if ep else
                                       f"dp{world}" + (": rows split over the GPUs, gradients reduce-scattered / parameters all-gathered over RCCL, Adam on the owned 1/N shard" if world > 1 else ""))},
            "validation_step": validation_step(head.get("eval"), a, bayesian, head["eB"], H, head["Mloc"]),
            "roofline": roof, "roofline_other": roof_other, "step_roofline": step_roofline(a, bayesian, head, ds, multihot, dt / a.steps), "cpu_baseline": state["cpu_baseline"], "exact_f32_mfma": state["exact_f32"], "mean_loss": head["mean_loss"],
            "mean_loss_finite_in_every_timed_region": state["loss_finite"],
            "kernel_ms_per_step": head["breakdown"], "kernel_ms_note": "separate pass of %d steps with events around every kernel family (side-stream families overlap the big kernels: the column does not sum to the step); the timed region carries events around the two output-layer kernels only" % head["k3"],
            "rccl_ranks": world if world > 1 else 1, "rank0_device": f"cuda:{local} {devices[0]}",
            "rccl_payload_bytes_per_step": head["rccl_payload_bytes_per_step"],
            "collective_waits": head.get("collective_waits"),
        }
        out.update(extra_modes)
        if not final: out["cut_off"] = "printed by rank 0's watchdog: an extra leg did not return"
        if state["extra_configs"]: out["extra_configs"] = state["extra_configs"]
        if gather: out["roofline_gather"] = gather
        if a.validate_on_one_gpu: out["validation_only"] = "all ranks on cuda:0 over gloo: code-path check, not a measurement"
        if a.ep_emulate:
            # one rank of G: it processed the whole global minibatch on 1/G of the experts, i.e. 1/G of the job
            shard = expert_shards(dims[-1], G)[0]
            out["metric"] += f" [one rank of {G} under --parallel ep, emulated on one GPU without the exchange]"
            out["value"] = a.steps * a.batch / dt; out["n_gpus"] = 1
            out["ep_emulation"] = {"G": G, "experts": [int(shard[0]), int(shard[1])], "rows_per_step": gB, "ms_per_step": dt / a.steps * 1e3,
                                   "projected_teams_per_s_at_G_gpus": a.steps * gB / dt, "note": "projection = G * this rank's rate; excludes the 4*B*h[-1]-byte all-reduce per step"}
        if a.dp_emulate:
            eb = head["emulated_bytes"] or {}; ns_ = max(eb.get("steps", 0), 1)
            rs, ar, ag = eb.get("reduce_scatter_in", 0) / ns_, eb.get("all_reduce", 0) / ns_, eb.get("all_gather_out", 0) / ns_
            fam = head["breakdown"] or {}
            out["metric"] += f" [one rank of {G} under --parallel dp, emulated on one GPU without the exchange]"
            out["value"] = a.steps * a.batch / dt; out["n_gpus"] = 1
            # ring reduce-scatter / all-gather over G ranks: every rank sends and receives (G - 1) / G of the buffer; the remainders and the replicated segments are all-reduced (2 (G - 1) / G)
            wire = ((rs + ag) * (G - 1) / G + ar * 2 * (G - 1) / G)
            out["dp_emulation"] = {"G": G, "rows_per_rank": a.batch, "global_batch": gB, "ms_per_step": dt / a.steps * 1e3, "projected_teams_per_s_at_G_gpus_before_exchange": a.steps * gB / dt,
                                   "collective_bytes_per_step": {"reduce_scatter_input": rs, "all_gather_output": ag, "all_reduce": ar, "sent_and_received_per_rank": wire},
                                   "exchange_ms_at_350_GBps_per_rank": wire / 350e9 * 1e3,
                                   "overlap_window_ms": {"reduce_scatter_behind": "the dW chunks that follow a chunk's own kernel", "dw_chunks_total": fam.get("out_fused_dw_adam"),
                                                         "all_gather_behind": ("the next step's head, range by range: the forward kernel of range j runs while the chunks of range j + 1 arrive (3 of 4 ranges; DESIGN.md 6.5)"
                                                                               if os.environ.get("NTF_DP_RANGES", "1") != "0" else "nothing (NTF_DP_RANGES=0: every gather waited for before the head, as in round 4)")},
                                   "note": "compute only: deferred dW in expert chunks (fuse_adam = 0), stand-alone operand producer, Adam on the 1/G shard; the collectives' bytes are tallied, not moved. "
                                           "350 GB/s = what a rank receives over its seven xGMI links in a ring (MI355X_MICROARCH.md: 7 x ~153 GB/s point to point, a ring uses one link each way)"}
        sys.stdout.flush(); sys.stderr.flush()
        print(json.dumps(out), file=real_stdout, flush=True)

    # ---- N > 1, --parallel auto: the other two ways of sharing a step, same processes
    if world > 1 and a.parallel == "auto" and not a.ep_emulate:
        def brief(r, scaling):
            return {"parallelism": r["par"], "scaling": scaling, "global_batch": r["gB"], "rows_per_rank": r["eB"], "ms_per_step": r["dt"] / a.steps * 1e3,
                    "value": a.steps * r["gB"] / r["dt"], "unit": "teams/s", "rccl_payload_bytes_per_step": r["rccl_payload_bytes_per_step"], "mean_loss": r["mean_loss"], "collective_waits": r.get("collective_waits")}
        # The headline is measured; nothing below may lose it.  Each extra leg: (1) every rank builds its engine, the ranks agree over the gloo control group, and the
        # leg is skipped everywhere unless all succeeded; (2) a Python exception inside the leg becomes {"error": ...} under its key on all ranks (agreed the same way);
        # (3) once a leg has failed while running, the collectives' state is unknown: the remaining legs are skipped; (4) a leg that does not come back at all is cut
        # off by rank 0's watchdog, which prints the line with what there is and leaves (a fresh exit, never a re-exec).
        broken = [None]

        def guarded(name, par, gB_leg, scaling, what):
            if broken[0]:
                extra_modes[name] = {"error": f"skipped: an earlier leg failed while running ({broken[0]})"}
                return
            err, r = None, None
            try:
                r = run_mode(par, gB_leg, a.steps, a.warmup, breakdown=False, leg=name, agree=all_ok)
                r["engine"].close()
            except LegSkipped as ex:
                extra_modes[name] = {"error": f"skipped before any collective: {ex}"}
                return
            except Exception as ex:
                err = f"{type(ex).__name__}: {ex}"
            if not all_ok(err is None):
                broken[0] = name
                extra_modes[name] = {"error": err or "another rank failed inside this leg"}
                return
            extra_modes[name] = brief(r, scaling)
            extra_modes[name]["what"] = what

        watchdog = None
        if rank == 0:
            budget = float(os.environ.get("NTF_BENCH_LEG_BUDGET_S", "150")) * 2

            def cut_off():
                for k in ("ep_weak", "strong_b1000"):
                    extra_modes.setdefault(k, {"error": f"no result within {budget:.0f} s of the extra legs (a hang?): cut off by rank 0's watchdog"})
                emit(final=False)
                os._exit(3)      # the headline is printed, but a leg hung: the launcher must see a failure (dp.CollectiveTimeout's contract)
            watchdog = threading.Timer(budget, cut_off); watchdog.daemon = True; watchdog.start()
        if shardable:
            guarded("ep_weak", "ep", a.batch * world, "weak",
                    "output layer split along the expert axis; every GPU steps the global minibatch of b x N teams on 1/N of the experts; only d(hidden) is all-reduced")
        spar = "ep" if can_shard(dims, world) and not a.no_fused else "dp"      # opentf_amd/mdl/fnn.py::_parallel_mode
        guarded("strong_b1000", spar, a.batch, "strong", f"the plugin under torchrun: the reference's global minibatch of cfg.b = {a.batch} teams, shared as _parallel_mode picks ({spar})")
        if watchdog is not None: watchdog.cancel()

    exact_f32 = None
    if world == 1 and a.mfma == "default" and not a.no_f32_line and not a.no_fused and not a.ep_emulate and not a.dp_emulate:
        # the same workload on the exact-f32 MFMA kernels (v_mfma_f32_32x32x2_f32, a bit-exact f32 fma chain): quoted beside the fp16x3 headline
        with torch.cuda.stream(stream):
            e2 = libntf.Engine(dims, bayesian=bayesian, input_mode=libntf.INPUT_MULTIHOT if multihot else libntf.INPUT_MEANPOOL, max_batch=a.batch, ns=5, nsd=a.nsd,
                               tpw=10.0, tnw=1.0, lr=1e-3, seed=1234, device=local, stream=stream.cuda_stream, mfma="f32")
            if not multihot: e2.set_skill_table(ds["table"])
            e2.set_skill_csr(ds["skill"]); e2.set_member(ds["member"]); e2.load_state_dict(sd0)
            if a.nsd == "unigram": e2.set_unigram(np.bincount(ds["member"][1], minlength=ds["M"]) / ds["N"])
            k2 = max(5, min(20, a.steps))
            order = rng.integers(0, ds["N"], (3 + k2) * gB).astype(np.int64)
            e2.train_epoch(order[: 3 * gB], gB); e2.synchronize()
            t1 = time.perf_counter(); e2.train_epoch(order[3 * gB: (3 + k2) * gB], gB); e2.synchronize()
            dt2 = time.perf_counter() - t1
            exact_f32 = {"value": k2 * gB / dt2, "unit": "teams/s", "ms_per_step": dt2 / k2 * 1e3, "steps": k2, "arithmetic": "--mfma f32: v_mfma_f32_32x32x2_f32 kernels"}
            e2.close()

    extra_configs = None
    if world == 1 and rank == 0 and a.dataset == "dblp" and not (a.rows or a.experts or a.no_extra_configs or a.ep_emulate or a.dp_emulate or a.no_fused or multihot) and a.mfma == "default":
        # the "full DBLP" reading of north_star: the UNFILTERED matrix's expert count (M = 5 022 955, 1.29 G parameters, ~75 GB resident); 10 steps.  The step does not
        # depend on the number of teams, so a 200 000-team sample of the 4 877 383 is staged.
        saved = (a.dataset,)
        dsf = make_dataset("dblp_full", d=a.d, seed=0, n_rows=200_000)
        dimsf = [a.d, a.hidden, dsf["M"]]
        a.dataset = "dblp_full"
        r = run_mode("dp", a.batch, 10, 2, reps_allowed=False, breakdown=True, data=dsf, model_dims=dimsf, params=init_params(dimsf, bayesian, 0)); r["engine"].close()
        rf, rfo = rooflines(r["times"], a, bayesian, r["eB"], a.hidden, r["Mloc"], dsf, False, r.get("evt_steps"), r.get("adam_in_dw", True))      # (traffic: no PMC pass at this size - null)
        extra_configs = {"dblp_full": {"workload": workload_label(a, dsf, bayesian, False), "steps": 10, "ms_per_step": r["dt"] / 10 * 1e3, "value": 10 * a.batch / r["dt"], "unit": "teams/s",
                                       "mean_loss": r["mean_loss"], "roofline": rf, "roofline_other": rfo}}
        a.dataset = saved[0]
        # the team-vector producer in front of this path when the input is doc2vec (src/mdl/emb/d2v.py:69-84): one PV-DM pass (d = 128, window 5, 5 negatives, gensim's
        # defaults as the reference leaves them) over THIS dataset's teams as documents.  The reference's own log of that stage on dblp mt10.ts2
        # (output/dblp/dblp.v12.json.mt10.ts2/prep.d2v.skill.log: gensim 4.3.3, 224 workers): 276 s per pass over 19 073 021 words, 100 passes = 7.7 h.
        from opentf_amd.mdl.emb import d2v as d2v_host
        d_ptr, d_idx = ds["skill"][0], ds["skill"][1]
        keys, count, si, cum, wi = d2v_host.build_vocab(d_idx)
        wv0, dv0 = d2v_host.initial_vectors(len(d_ptr) - 1, len(keys), a.d, 0)
        net = libntf.Doc2Vec(d_ptr, wi, si, cum, wv0, dv0, seed=0, device=local)
        prog = d2v_host.job_progress(d_ptr)
        net.train_epoch(1, 5, 0.025, 0.001, 0, progress=prog)
        ms = [net.train_epoch(1, 5, 0.025, 0.001, 1 + k, progress=prog, want_ms=True)[1] for k in range(3)]
        net.close()
        extra_configs["d2v_epoch"] = {"workload": f"doc2vec PV-DM pass (src/mdl/emb/d2v.py:76-84) over {len(d_ptr) - 1} teams as documents of their skills: {len(d_idx)} words, {len(keys)} distinct, d={a.d}, window 5, negative 5",
                                      "ms_per_pass": float(np.median(ms)), "value": len(d_idx) / (float(np.median(ms)) * 1e-3), "unit": "words/s",
                                      "reference_log": {"file": "output/dblp/dblp.v12.json.mt10.ts2/prep.d2v.skill.log", "s_per_pass": 276.4, "raw_words_per_s": 19073021 / 276.4, "workers": 224, "words": 19073021}}

    leg_failed = any("error" in v and not str(v["error"]).startswith("skipped before any collective") for v in extra_modes.values() if isinstance(v, dict))
    if extra_configs is not None and not os.environ.get("NTF_BENCH_NO_PLUGIN_EPOCH"):
        try:
            extra_configs["plugin_epoch"] = plugin_epoch(ds, a, local, a.steps * gB / head["dt"])
        except Exception as ex:      # (the headline must not depend on it)
            extra_configs["plugin_epoch"] = {"error": f"{type(ex).__name__}: {ex}"}

    if rank != 0:
        if world > 1 and not any("error" in v for v in extra_modes.values() if isinstance(v, dict)): dist.destroy_process_group()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(4 if leg_failed else 0) if world > 1 else None      # (N > 1: no interpreter teardown over process groups whose state an abandoned leg may have left undefined; a leg that failed while running: non-zero)
        return
    if a.force_dist and world == 1: dist.destroy_process_group()
    state["exact_f32"], state["extra_configs"] = exact_f32, extra_configs
    if world == 1 and not a.no_cpu_baseline and not multihot and not a.ep_emulate and not a.dp_emulate:   # the CPU leg times the headline (mean-pool) configuration only
        state["cpu_baseline"] = cpu_baseline(ds, dims, bayesian, cfg)
    clean = not any("error" in v for v in extra_modes.values() if isinstance(v, dict))
    if world > 1 and clean: dist.destroy_process_group()   # before the JSON line: RCCL prints its version banner when the group goes away
    emit(final=True)
    if not state["loss_finite"]:      # the line is printed (it says which); a benchmark whose loss is not a number has measured nothing
        print("bench.py: a timed region's mean loss is not finite", file=sys.stderr, flush=True)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(5)
    if world > 1:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(4 if leg_failed else 0)      # the line (with the leg's {"error": ...}) is printed; the exit code says a leg failed while running


if __name__ == "__main__":
    main()
