#!/usr/bin/env python3
"""teams/sec (train) of the Bnn `bnn_emb d=128` minibatch step on dblp-shaped data, on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one minibatch (B = 1000 teams per GPU) through the whole hot path of src/mdl/fnn.py:118-140:
CSR mean-pool gather of the skill embeddings, Flipout MLP forward, sparse-label weighted BCE with uniform
negative sampling, KL term, backward, (gradient all-reduce,) Adam.  Inputs (CSR matrices, embedding table,
weights) are resident in HBM before the timed region; nothing is skipped inside it.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant kernel (HIP
events recorded on the engine's stream during the timed region) and `cpu_baseline` (the oracle's
reference-shaped dense step, timed on this box's host cores, N=1 only).
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
    ap.add_argument("--mfma", default="default", choices=["default", "f32", "bf16x6", "fp16x3"], help="arithmetic of the fused output-layer products (include/opentf_amd.h ntf_mfma)")
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
    ap.add_argument("--ep-emulate", type=int, default=0, metavar="G",
                    help="N = 1 only: run what ONE rank of G runs under --parallel ep (its 1/G of the experts, a global minibatch of G * --batch teams, two-phase "
                         "step, no exchange) - the per-rank compute time behind the scaling projection in DESIGN.md; not the headline")
    return ap.parse_args()


def cpu_baseline(ds, dims, bayesian, cfg, sample_rows=1000, steps=3):
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
    for n in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128, ncpu)}):
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
    return {"value": steps * sample_rows / dt, "unit": "teams/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of B={sample_rows} at full M={ds['M']} (oracle/ntf_oracle.py reference_shaped_step, torch {torch.__version__} CPU)"}


def pmc_traffic(family, a, ds):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/, collected and corrected
    as MI355X_MICROARCH.md prescribes: FETCH_SIZE doubled, separate passes); only when the run is the profiled configuration."""
    path = os.path.join(ROOT, "profiles", {"f32": "r1_c_pmc_traffic_and_sq.json", "bf16x6": "r1_d_pmc_traffic_and_sq.json"}.get(a.mfma, "r2_pmc_traffic_and_sq.json"))
    if not os.path.exists(path): path = os.path.join(ROOT, "profiles", "r1_e_pmc_traffic_and_sq.json")
    if not (os.path.exists(path) and a.dataset == "dblp" and a.model == "bnn" and a.batch == 1000 and a.d == 128 and a.hidden == 128
            and a.input == "meanpool" and not a.rows and not a.experts):
        return None
    key = {"out_fused_fwd_loss_dh": "k_out_fwd", "out_fused_dw_adam": "k_out_dw"}.get(family)
    if not key:
        return None
    for name, v in json.load(open(path))["kernels"].items():
        if key in name:
            return v["hbm_bytes"]
    return None


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
        if a.validate_on_one_gpu: dist.init_process_group("gloo")
        else: dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))

    from opentf_amd import libntf
    from opentf_amd.dp import DataParallel
    from opentf_amd.ep import ExpertParallel, expert_shards, can_shard
    from opentf_amd.synth import make_dataset, init_params

    bayesian = a.model == "bnn"
    ds = make_dataset(a.dataset, d=a.d, seed=0, n_rows=a.rows or None, n_experts=a.experts or None)
    multihot = a.input == "multihot"
    dims = [ds["S"] if multihot else a.d, a.hidden, ds["M"]]
    cfg = {"ns": 5, "nsd": a.nsd, "tpw": 10.0, "tnw": 1.0, "lr": 1e-3}
    if a.ep_emulate and world > 1: raise SystemExit("--ep-emulate is a single-GPU measurement")
    G = a.ep_emulate if a.ep_emulate else world
    par = "dp"
    if (G > 1 or (a.force_dist and a.parallel == "ep")) and not a.no_fused and a.parallel != "dp":
        if can_shard(dims, G): par = "ep"
        elif a.parallel == "ep": raise SystemExit(f"--parallel ep: {dims} does not shard over {G} GPUs (needs h[-1] in 32/64/128 and >= {G} tiles of 256 experts)")
    ep = par == "ep"
    shard = expert_shards(dims[-1], G)[0 if a.ep_emulate else rank] if ep else None
    eB = a.batch * G if ep else a.batch                     # rows one engine steps: under ep every rank steps the whole global minibatch
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        e = libntf.Engine(dims, bayesian=bayesian, input_mode=libntf.INPUT_MULTIHOT if multihot else libntf.INPUT_MEANPOOL, max_batch=eB, ns=5, nsd=a.nsd, tpw=10.0, tnw=1.0,
                          lr=1e-3, seed=1234, device=local, stream=stream.cuda_stream, fused=not a.no_fused,
                          fuse_adam=a.fuse_adam if (world == 1 or ep) else 0, mfma=a.mfma, expert_shard=shard, ep_world=G if ep else 1)
        if not multihot: e.set_skill_table(ds["table"])
        e.set_skill_csr(ds["skill"]); e.set_member(ds["member"])
        e.load_state_dict(init_params(dims, bayesian, 0))
        if a.nsd == "unigram":   # expert frequency over the training rows (src/mdl/fnn.py:97)
            e.set_unigram(np.bincount(ds["member"][1], minlength=ds["M"]) / ds["N"])
        if a.gather_only:
            for _ in range(5): e.gather_meanpool(n=ds["N"], to_host=False)
            e.synchronize()
            print(json.dumps({"gather_only": True, "teams": ds["N"]}), file=real_stdout, flush=True)
            return
        dp = ExpertParallel(e, two_phase=bool(a.ep_emulate)) if ep else DataParallel(e)
        gB = a.batch * G                                       # weak scaling: B teams per GPU
        rng = np.random.default_rng(7)
        total_steps = a.warmup + a.steps
        order = rng.integers(0, ds["N"], total_steps * gB).astype(np.int64)   # the loader's shuffled row order
        if a.warmup:
            dp.train_epoch(order[: a.warmup * gB], gB)
        e.kernel_times(enable=2)    # HIP events around the two output-layer kernels only inside the timed region (the roofline's kernels)
        e.synchronize(); torch.cuda.synchronize()
        if world > 1: dist.barrier()
        t0 = time.perf_counter()
        mean_loss = dp.train_epoch(order[a.warmup * gB:], gB)
        e.synchronize(); torch.cuda.synchronize()
        if world > 1: dist.barrier()
        dt = time.perf_counter() - t0
        times = e.kernel_times(enable=False)
        # per-family breakdown from a SEPARATE short pass (events around every family perturb the step by a few per cent)
        k3 = max(5, min(10, a.steps))
        e.kernel_times(enable=True)
        dp.train_epoch(order[: k3 * gB], gB); e.synchronize()
        breakdown = {f: round(v[0] / k3, 4) for f, v in e.kernel_times(enable=False).items() if v[1] > 0}
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1: dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

        gather = None
        if (a.gather_bench or (world == 1 and not a.no_gather_bench and not a.ep_emulate)) and rank == 0 and not multihot:
            e.kernel_times(enable=True)
            n = ds["N"]
            e.gather_meanpool(n=n, to_host=False); e.kernel_times(enable=True)
            for _ in range(3): e.gather_meanpool(n=n, to_host=False)
            ms, calls = e.kernel_times(enable=False)["gather"]
            nnz = ds["skill"][0][-1] / n
            bytes_per_team = nnz * (4 * a.d + 4) + 8 + 4 * a.d
            gather = {"bound": "hbm", "achieved": bytes_per_team * n / (ms / calls * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "bytes_per_team": bytes_per_team, "teams": n, "ms": ms / calls}
            # the roof of THIS kernel: its table rows are random reads of a 46 MB table, served on-die (MI355X_MICROARCH.md, Indexed rows: 8.6 TB/s for
            # uniformly random rows of a 38 MB table out of the Infinity Cache); only the CSR and the output rows are compulsory HBM traffic (8 TB/s)
            t_roof = n * nnz * 4 * a.d / 8.6e12 + (n * (4 * a.d + 8 + nnz * 4) + ds["S"] * a.d * 4) / (HBM_PEAK_GBS * 1e9)
            gather["peak"] = bytes_per_team * n / t_roof / 1e9
            gather["peak_def"] = "algorithmic bytes / (gathered table bytes / 8.6 TB/s on-die + (output + CSR + table once) / 8 TB/s HBM)"
            gather["frac"] = gather["achieved"] / gather["peak"]
            pm = os.path.join(ROOT, "profiles", "r2_pmc_gather.json")
            if os.path.exists(pm):
                for name, v in json.load(open(pm))["kernels"].items():
                    if "k_gather_pool" in name and "hbm_bytes" in v: gather["traffic"] = v["hbm_bytes"]
            # the algorithmic bytes count every gathered table row; the 46 MB table itself stays on-die (Infinity Cache / L2), so the compulsory
            # HBM traffic is the output rows + the CSR (+ the table once)
            comp = n * (4 * a.d + 8 + nnz * 4) + ds["S"] * a.d * 4
            gather["hbm_compulsory_gbs"] = comp / (ms / calls * 1e-3) / 1e9
            gather["note"] = "achieved = algorithmic bytes (SURVEY 8d: nnz*(4d+4)+8+4d per team) / time; the table rows are served on-die, so the roof is not the HBM peak (peak_def)"

    exact_f32 = None
    if world == 1 and a.mfma == "default" and not a.no_f32_line and not a.no_fused and not a.ep_emulate:
        # the same workload on the exact-f32 MFMA kernels (v_mfma_f32_32x32x2_f32, a bit-exact f32 fma chain): quoted beside the fp16x3 headline
        with torch.cuda.stream(stream):
            e.close()
            e2 = libntf.Engine(dims, bayesian=bayesian, input_mode=libntf.INPUT_MULTIHOT if multihot else libntf.INPUT_MEANPOOL, max_batch=a.batch, ns=5, nsd=a.nsd,
                               tpw=10.0, tnw=1.0, lr=1e-3, seed=1234, device=local, stream=stream.cuda_stream, mfma="f32")
            if not multihot: e2.set_skill_table(ds["table"])
            e2.set_skill_csr(ds["skill"]); e2.set_member(ds["member"]); e2.load_state_dict(init_params(dims, bayesian, 0))
            if a.nsd == "unigram": e2.set_unigram(np.bincount(ds["member"][1], minlength=ds["M"]) / ds["N"])
            k2 = max(5, min(20, a.steps))
            e2.train_epoch(order[: 3 * gB], gB); e2.synchronize()
            t1 = time.perf_counter(); e2.train_epoch(order[3 * gB: (3 + k2) * gB], gB); e2.synchronize()
            dt2 = time.perf_counter() - t1
            exact_f32 = {"value": k2 * gB / dt2, "unit": "teams/s", "ms_per_step": dt2 / k2 * 1e3, "steps": k2, "arithmetic": "--mfma f32: v_mfma_f32_32x32x2_f32 kernels"}
            e2.close()

    if rank != 0:
        if world > 1: dist.destroy_process_group()
        return
    if a.force_dist and world == 1: dist.destroy_process_group()
    B, H, M = a.batch, a.hidden, ds["M"]
    Mloc = (shard[1] - shard[0]) if ep else M                # experts this GPU's kernels cover, over eB rows
    gemm = 2.0 * eB * H * Mloc  # one [rows,H]x[H,experts]-sized product of a launch
    k = 2 if bayesian else 1
    # per timed scope: the unfused families launch one GEMM per Flipout half (k launches), the fused ones a single kernel
    flops_per_launch = {"out_fwd_gemm": k * gemm, "out_bwd_dw_gemm": k * gemm, "out_bwd_da_gemm": k * gemm,
                        "out_fused_fwd_loss_dh": 2 * k * gemm, "out_fused_dw_adam": k * gemm}
    cand = {f: times[f] for f in flops_per_launch if f in times and times[f][1] > 0}
    dom = max(cand, key=lambda f: cand[f][0]) if cand else None
    roof = None
    if dom:
        ms, calls = cand[dom]
        ach = flops_per_launch[dom] / (ms / calls * 1e-3) / 1e12
        # arithmetic of the dominant kernel: "bf16x6" = every f32 operand split exactly into 3 bf16 values, a product = 6 bf16 MFMA products
        # accumulated in f32 (f32-accurate).  The roof for ALGORITHMIC flops is then the dense bf16 MFMA peak / 6.
        split = a.mfma != "f32" and not a.no_fused and ((dom == "out_fused_dw_adam") or (dom == "out_fused_fwd_loss_dh" and a.hidden == 128))
        nprod = (6 if a.mfma == "bf16x6" else 3) if split else 1
        peak = BF16_MFMA_PEAK_TFLOPS / nprod if split else F32_MFMA_PEAK_TFLOPS
        arith = {1: "f32 MFMA (v_mfma_f32_32x32x2_f32)",
                 6: "bf16x6: operands split exactly into 3 bf16 values, 6 bf16 MFMA products per f32 product, f32 accumulate; peak = 2516.6 / 6",
                 3: "fp16x3: operands * 2^k split into 2 fp16 values (22 bits), 3 fp16 MFMA products per f32 product, f32 accumulate; peak = 2516.6 / 3"}[nprod]
        roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": None if ep else pmc_traffic(dom, a, ds),
                "traffic_source": "per-launch mean of the separate rocprofv3 --pmc passes of this command committed under profiles/ (collect_r2.sh), not counted in this run",
                "avg_ms": ms / calls, "launches": calls, "flops_per_launch": flops_per_launch[dom],
                "arithmetic": arith, "hw_mfma_tflops": ach * nprod}
    out = {
        "metric": "teams/sec (train) bnn_emb d=128 on DBLP", "value": a.steps * gB / dt, "unit": "teams/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"f32": "f32", "bf16x6": "f32 (bf16x6 split products, f32 accumulate)"}.get(a.mfma, "f32 (fp16x3 split products, f32 accumulate)"), "data": "synthetic",
        "config": {"workload": f"{a.dataset} mt10.ts2 shapes N={ds['N']} S={ds['S']} M={M}; {a.model}{' (Flipout)' if bayesian else ''} on " +
                               (f"multi-hot skill rows D={ds['S']}, " if multihot else f"mean-pooled skill table d={a.d}, ") +
                               f"h=[{H}], b={B}/GPU, ns=5 {a.nsd}, tpw 10 tnw 1, Adam lr 1e-3", "global_batch": gB,
                   "parallelism": (f"ep{world}: expert-sharded output layer, every GPU steps the global minibatch on 1/{world} of the experts, d(hidden) all-reduced" if ep else f"dp{world}")},
        "roofline": roof, "cpu_baseline": None, "exact_f32_mfma": exact_f32, "mean_loss": mean_loss,
        "kernel_ms_per_step": breakdown, "kernel_ms_note": "separate pass of %d steps with events around every kernel family; the timed region carries events around the two output-layer kernels only" % k3,
    }
    if gather: out["roofline_gather"] = gather
    if a.validate_on_one_gpu: out["validation_only"] = "all ranks on cuda:0 over gloo: code-path check, not a measurement"
    if a.ep_emulate:
        # one rank of G: it processed the whole global minibatch on 1/G of the experts, i.e. 1/G of the job
        out["metric"] += f" [one rank of {G} under --parallel ep, emulated on one GPU without the exchange]"
        out["value"] = a.steps * a.batch / dt; out["n_gpus"] = 1
        out["ep_emulation"] = {"G": G, "experts": [int(shard[0]), int(shard[1])], "rows_per_step": gB, "ms_per_step": dt / a.steps * 1e3,
                               "projected_teams_per_s_at_G_gpus": a.steps * gB / dt, "note": "projection = G * this rank's rate; excludes the 4*B*h[-1]-byte all-reduce per step"}
    if world == 1 and not a.no_cpu_baseline and not multihot and not a.ep_emulate:   # the CPU leg times the headline (mean-pool) configuration only
        out["cpu_baseline"] = cpu_baseline(ds, dims, bayesian, cfg)
    if world > 1: dist.destroy_process_group()   # before the JSON line: RCCL prints its version banner when the group goes away
    sys.stdout.flush(); sys.stderr.flush()
    print(json.dumps(out), file=real_stdout, flush=True)


if __name__ == "__main__":
    main()
