"""Expert-sharded output layer (opentf_amd/ep.py, ntf_step_staged_ep): G engines, each owning a contiguous range of experts, emulated on ONE
GPU (the exchange - the sum of d(hidden) over the shards - is done with torch here, by RCCL in production).  The contract under test: with
the NATIVE generators (keyed by global expert ids) the shards together compute the step one engine holding the whole layer computes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from opentf_amd.ep import expert_shards                      # noqa: E402
from opentf_amd.synth import make_dataset, init_params       # noqa: E402


def _mk(ds, dims, bayesian, B, nsd, shard=None, world=1, fuse_adam=1, multihot=False, seed=5, mfma=None):
    from opentf_amd import libntf
    mode = libntf.INPUT_MULTIHOT if multihot else libntf.INPUT_MEANPOOL
    e = libntf.Engine(dims, bayesian=bayesian, input_mode=mode, max_batch=B, ns=5, nsd=nsd, tpw=10.0, tnw=1.0, lr=1e-3, seed=seed,
                      fuse_adam=fuse_adam, expert_shard=shard, ep_world=world, mfma=mfma)
    if not multihot: e.set_skill_table(ds["table"])
    e.set_skill_csr(ds["skill"]); e.set_member(ds["member"])
    e.load_state_dict(init_params(dims, bayesian, 0))
    if nsd == "unigram":
        e.set_unigram(np.bincount(ds["member"][1], minlength=ds["M"]) / ds["N"])
    return e


def _ep_epoch(engines, order, B, train=True):
    import torch
    for e in engines:
        e.stage_order(order); e.epoch_loss()
    H = engines[0].dims[-2]
    for off in range(0, len(order), B):
        b = min(B, len(order) - off)
        if not train:
            for e in engines: e.step_staged(off, b, train=False, apply=False)
            continue
        for e in engines:
            e.step_staged_ep(off, b, 1); e.synchronize()
        dh = [e.dh_tensor() for e in engines]
        if dh[0] is not None:
            tot = torch.stack([d[: b * H] for d in dh]).sum(0)
            for d in dh: d[: b * H].copy_(tot)
            torch.cuda.synchronize()
        for e in engines:
            e.step_staged_ep(off, b, 2); e.step_staged_ep(off, b, 3)
    return sum(e.epoch_loss()[0] for e in engines)


def _full_epoch(e, order, B, train=True):
    e.stage_order(order); e.epoch_loss()
    for off in range(0, len(order), B):
        b = min(B, len(order) - off)
        e.step_staged(off, b, train=train, apply=train)
    return e.epoch_loss()[0]


def _gathered(engines, exact_replicas=True):
    sds = [e.state_dict() for e in engines]
    last = f"layers.{engines[0].L - 1}."
    out = {}
    for k in sds[0]:
        if k.startswith(last): out[k] = np.concatenate([sd[k] for sd in sds])
        else:
            for sd in sds[1:]:
                if exact_replicas: assert np.array_equal(sd[k], sds[0][k]), f"replicated {k} differs between shards"
                else: np.testing.assert_allclose(sd[k], sds[0][k], rtol=1e-5, atol=1e-6, err_msg=k)
            out[k] = sds[0][k]
    return out


CASES = {
    # name: (bayesian, dims builder, nsd, G, B, multihot)
    "bnn_uniform_g3": (True, lambda ds: [128, 128, ds["M"]], "uniform", 3, 256, False),
    "fnn_uniform_g2": (False, lambda ds: [128, 128, ds["M"]], "uniform", 2, 256, False),
    "bnn_unigram_g4": (True, lambda ds: [128, 128, ds["M"]], "unigram", 4, 200, False),
    "bnn_unigram_b_g2": (True, lambda ds: [128, 128, ds["M"]], "unigram_b", 2, 200, False),
    "bnn_two_hidden_h64_g2": (True, lambda ds: [128, 96, 64, ds["M"]], "uniform", 2, 200, False),
    "bnn_multihot_g2": (True, lambda ds: [ds["S"], 128, ds["M"]], "uniform", 2, 200, True),
    "bnn_no_hidden_g2": (True, lambda ds: [128, ds["M"]], "uniform", 2, 200, False),
    # where Adam of the output layer runs: 0 = one flat kernel at the end of phase 3, 2 = per expert chunk on a side stream (1, the default above: in the dW epilogue)
    "bnn_flat_adam_g2": (True, lambda ds: [128, 128, ds["M"]], "uniform", 2, 256, False, 0),
    "bnn_chunked_adam_g2": (True, lambda ds: [128, 128, ds["M"]], "uniform", 2, 256, False, 2),
}


@pytest.mark.parametrize("case", sorted(CASES) + ["bnn_uniform_g3+head_prefetch", "bnn_unigram_b_g2+head_prefetch", "bnn_uniform_g3+head_prefetch_late"])
def test_expert_shards_compute_the_single_engine_step(case, monkeypatch):
    case, _, variant = case.partition("+")
    ep_hp = variant.startswith("head_prefetch")      # a shard's phase 3 issuing the next batch's head (by default only where the dW kernel outlasts the backward: DESIGN.md section 6.3)
    monkeypatch.setenv("NTF_EP_HEAD_PREFETCH", {"head_prefetch": "1", "head_prefetch_late": "2"}.get(variant, "0"))      # (unset = by shape: these shards are wide enough for it; both ends are pinned here)
    bayesian, mkdims, nsd, G, B, multihot = CASES[case][:6]
    fuse_adam = CASES[case][6] if len(CASES[case]) > 6 else 1
    ds = make_dataset("dblp", d=128, seed=3, n_rows=1500, n_experts=3000)
    dims = mkdims(ds)
    order = np.random.default_rng(4).permutation(ds["N"])[: 2 * B + 77].astype(np.int64)    # two full batches and a ragged one
    shards = expert_shards(ds["M"], G)
    assert shards[0][0] == 0 and shards[-1][1] == ds["M"] and all(a[1] == b[0] and a[1] % 256 == 0 for a, b in zip(shards, shards[1:]))
    full = _mk(ds, dims, bayesian, B, nsd, multihot=multihot, fuse_adam=fuse_adam)
    eng = [_mk(ds, dims, bayesian, B, nsd, shard=s, world=G, multihot=multihot, fuse_adam=fuse_adam) for s in shards]

    # --- first step: the output layer sees bit-identical operands on both sides
    l_full = _full_epoch(full, order[:B], B); l_ep = _ep_epoch(eng, order[:B], B)
    assert abs(l_ep - l_full) <= 2e-6 * abs(l_full), (l_ep, l_full)
    # replicas of the hidden layers stay bit-identical (same summed d(hidden), deterministic kernels) - except behind the multi-hot first
    # layer, whose weight gradient is a scatter-add by float atomics (ExpertParallel re-broadcasts the replicated parameters every epoch)
    a, b = _gathered(eng, exact_replicas=not multihot), full.state_dict()
    last = f"layers.{full.L - 1}."
    for k in b:
        if k.startswith(last): assert np.array_equal(a[k], b[k]), f"{k}: the shards' first update differs from the single engine's"
        else: np.testing.assert_allclose(a[k], b[k], rtol=1e-5, atol=2e-6, err_msg=k)   # d(hidden) is summed in another order

    # --- a short epoch on top, then an evaluation phase
    hits0 = [e.head_prefetch_hits() for e in eng]
    l_full = _full_epoch(full, order, B); l_ep = _ep_epoch(eng, order, B)
    assert abs(l_ep - l_full) <= 1e-5 * abs(l_full), (l_ep, l_full)
    if ep_hp and bayesian and len(dims) == 3 and not multihot and fuse_adam == 1:
        # (NTF_EP_HEAD_PREFETCH=1: a shard's phase 3 issues the next batch's head beside its dW kernel, as the single engine's side stream does: batches 2 and 3 of the epoch took it)
        assert all(e.head_prefetch_hits() - h == 2 for e, h in zip(eng, hits0)), [e.head_prefetch_hits() - h for e, h in zip(eng, hits0)]
    else:
        assert all(e.head_prefetch_hits() == h for e, h in zip(eng, hits0))
    a, b = _gathered(eng, exact_replicas=not multihot), full.state_dict()
    for k in b:
        # Adam normalises every gradient element by its own running magnitude: where a gradient is ~0, a last-bit difference of the summed d(hidden) (another
        # summation order over the shards) moves the update by a visible fraction of lr.  A handful of such elements per million may leave the band; none far.
        bad = ~np.isclose(a[k], b[k], rtol=1e-4, atol=2e-5)
        assert bad.sum() <= max(1, a[k].size // 50_000), (k, int(bad.sum()), a[k].size)
        np.testing.assert_allclose(a[k], b[k], rtol=1e-4, atol=3 * 1e-3 * 0.1, err_msg=k)     # <= 10 % of lr per step over the three steps
    v_full = _full_epoch(full, order[:300], B, train=False); v_ep = _ep_epoch(eng, order[:300], B, train=False)
    assert abs(v_ep - v_full) <= 1e-5 * abs(v_full), (v_ep, v_full)
    for e in eng + [full]: e.close()


def test_one_shard_of_world_one_is_the_plain_engine():
    """expert_shard = the whole layer, ep_world = 1: both the ordinary step and the two-phase step are the plain engine's step, bit for bit"""
    ds = make_dataset("dblp", d=128, seed=1, n_rows=600, n_experts=1000)
    dims = [128, 128, ds["M"]]
    order = np.arange(512, dtype=np.int64)
    plain, one, two = _mk(ds, dims, True, 256, "uniform"), _mk(ds, dims, True, 256, "uniform", shard=(0, ds["M"])), _mk(ds, dims, True, 256, "uniform", shard=(0, ds["M"]))
    lp, lo, lt = _full_epoch(plain, order, 256), _full_epoch(one, order, 256), _ep_epoch([two], order, 256)
    assert lp == lo == lt
    sp, so, st = plain.state_dict(), one.state_dict(), two.state_dict()
    for k in sp: assert np.array_equal(sp[k], so[k]) and np.array_equal(sp[k], st[k]), k


def test_expert_shard_contract_errors():
    from opentf_amd import libntf
    ds = make_dataset("dblp", d=128, seed=1, n_rows=300, n_experts=1000)
    with pytest.raises(libntf.NtfError, match="multiple of 256"):
        _mk(ds, [128, 128, ds["M"]], True, 64, "uniform", shard=(100, 1000), world=2)
    with pytest.raises(libntf.NtfError, match="fused output-layer path"):
        _mk(ds, [128, 48, ds["M"]], True, 64, "uniform", shard=(0, 512), world=2)
    e = _mk(ds, [128, 128, ds["M"]], True, 64, "uniform", shard=(0, 512), world=2)
    e.stage_order(np.arange(64, dtype=np.int64))
    with pytest.raises(libntf.NtfError, match="ntf_step_staged_ep"):
        e.step_staged(0, 64, train=True, apply=True)           # a train step of one shard alone would use a partial d(hidden)
    with pytest.raises(libntf.NtfError, match="phase 2 without"):
        e.step_staged_ep(0, 64, 2)
    with pytest.raises(libntf.NtfError, match="phase 3 without"):
        e.step_staged_ep(0, 64, 3)
    e.step_staged(0, 64, train=False, apply=False)              # evaluation needs no exchange
    plain = _mk(ds, [128, 128, ds["M"]], True, 64, "uniform")
    plain.stage_order(np.arange(64, dtype=np.int64))
    with pytest.raises(libntf.NtfError, match="not created as an expert shard"):
        plain.step_staged_ep(0, 64, 1)


def test_wide_minibatch_on_a_narrow_shard():
    """what one rank of eight runs at BASELINE config 2 in miniature: many row blocks (B = 2000) against few expert tiles; fused == generic path"""
    ds = make_dataset("dblp", d=128, seed=2, n_rows=4000, n_experts=2048)
    dims = [128, 128, ds["M"]]
    order = np.random.default_rng(0).permutation(ds["N"])[:2000].astype(np.int64)
    shards = expert_shards(ds["M"], 4)
    full = _mk(ds, dims, True, 2000, "uniform")
    eng = [_mk(ds, dims, True, 2000, "uniform", shard=s, world=4) for s in shards]
    l_full, l_ep = _full_epoch(full, order, 2000), _ep_epoch(eng, order, 2000)
    assert abs(l_ep - l_full) <= 2e-6 * abs(l_full)
    a, b = _gathered(eng), full.state_dict()
    for k in b:
        if k.startswith("layers.1."): assert np.array_equal(a[k], b[k]), k
        else: np.testing.assert_allclose(a[k], b[k], rtol=1e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("bayesian", [True, False])
def test_split_k_weight_gradient_kernel_equals_the_unsplit_one(bayesian, monkeypatch):
    """few expert tiles: the dW kernel's K (batch) range is split over workgroups and k_out_dw_finish runs the epilogue (gradient finalisation, fused Adam);
    same sums in another order -> same update up to rounding, with and without the fused Adam, and the gradients themselves (half-tile workgroups,
    k_out_dw_q<.., SPLIT>; the one-workgroup-per-CU split form k_out_dw_p2 was retired in round 6)"""
    ds = make_dataset("dblp", d=128, seed=6, n_rows=3000, n_experts=1300)     # 6 tiles of 256 experts, the last one ragged
    dims = [128, 128, ds["M"]]
    order = np.random.default_rng(1).permutation(ds["N"])[:1500].astype(np.int64)
    def run(ks, fuse_adam):
        monkeypatch.setenv("NTF_DW_KSPLIT", str(ks))
        e = _mk(ds, dims, bayesian, 1000, "uniform", fuse_adam=fuse_adam)
        e.stage_order(order)
        e.step_staged(0, 1000, train=True, apply=False); g = e.grads() if not fuse_adam else None
        if not fuse_adam: e.apply()
        loss = _full_epoch(e, order, 1000)
        sd = e.state_dict(); e.close()
        return loss, sd, g
    for fuse_adam in (0, 1):
        (l1, p1, g1), (l5, p5, g5) = run(1, fuse_adam), run(5, fuse_adam)
        assert abs(l1 - l5) <= 1e-6 * abs(l1)
        for k in p1: np.testing.assert_allclose(p5[k], p1[k], rtol=1e-4, atol=2e-5, err_msg=k)
        if g1 is not None:
            for k in g1:
                scale = float(np.abs(g1[k]).max())
                assert float(np.abs(g5[k] - g1[k]).max()) <= 2e-6 * scale, k


def test_wide_minibatch_hidden_bias_gradients():
    """B >= 1024 takes the row-chunked bias-gradient kernels: the gradients of one 1536-row step equal the sum of two 768-row shards' (one-wave-per-column kernel)"""
    ds = make_dataset("dblp", d=128, seed=8, n_rows=2000, n_experts=1500)
    dims = [128, 96, 128, ds["M"]]
    order = np.random.default_rng(3).permutation(ds["N"])[:1536].astype(np.int64)
    full, a, b = (_mk(ds, dims, True, 1536, "uniform", fuse_adam=0) for _ in range(3))
    for e in (full, a, b): e.stage_order(order)
    full.step_staged(0, 1536, 0, 1536, train=True, apply=False)
    a.step_staged(0, 768, 0, 1536, train=True, apply=False); b.step_staged(768, 768, 0, 1536, train=True, apply=False)
    gf, ga, gb = full.grads(), a.grads(), b.grads()
    for k in gf:
        if "bias" not in k: continue
        s = ga[k] + gb[k]
        assert float(np.abs(s - gf[k]).max()) <= 2e-5 * float(np.abs(gf[k]).max()) + 1e-9, k


@pytest.mark.parametrize("bayesian", [True, False])
def test_two_stream_step_equals_the_one_stream_step(bayesian, monkeypatch):
    """NTF_SIDE_BWD (default 1) moves the operand producer and the hidden layers' backward to a side stream beside the big kernels: the same kernels on the
    same data - parameters bit-identical to the one-stream step's, losses equal up to the order of the KL atomics"""
    ds = make_dataset("dblp", d=128, seed=9, n_rows=3000, n_experts=70_000)      # 274 expert tiles: several rounds of the dW kernel, no split-K
    dims = [128, 64, 128, ds["M"]]
    order = np.random.default_rng(2).permutation(ds["N"])[:2500].astype(np.int64)
    out = []
    for side in ("0", "1"):
        monkeypatch.setenv("NTF_SIDE_BWD", side)
        e = _mk(ds, dims, bayesian, 1000, "uniform")
        loss = _full_epoch(e, order, 1000)
        v = _full_epoch(e, order[:1000], 1000, train=False)
        out.append((loss, v, e.state_dict())); e.close()
    (l0, v0, p0), (l1, v1, p1) = out
    assert abs(l0 - l1) <= 1e-9 * abs(l0) and abs(v0 - v1) <= 1e-9 * abs(v0)
    for k in p0: assert np.array_equal(p0[k], p1[k]), k


def test_two_processes_one_gpu(tmp_path):
    """world size 2 for real: two processes, each with its own engine on this GPU, ExpertParallel over gloo (which moves CUDA tensors; RCCL refuses two ranks
    on one device).  Exercises what the emulations above cannot: the asynchronous all-reduce between the phases, the loss all-reduce, the gather of the
    state_dict, the epoch-end broadcast of the replicated layers - against the single-engine run."""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ep_two_process_check.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(port), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    got = np.load(tmp_path / "ep2.npz")
    ds = make_dataset("dblp", d=128, seed=3, n_rows=1500, n_experts=3000)
    dims = [128, 128, ds["M"]]
    order = np.random.default_rng(4).permutation(ds["N"])[:577].astype(np.int64)
    full = _mk(ds, dims, True, 256, "uniform")
    t_loss = _full_epoch(full, order, 256) / 3
    v_loss = _full_epoch(full, order[:300], 256, train=False) / 2
    assert abs(float(got["t_loss"]) - t_loss) <= 1e-5 * abs(t_loss) and abs(float(got["v_loss"]) - v_loss) <= 1e-5 * abs(v_loss)
    ref = full.state_dict()
    for k in ref:
        assert got[k].shape == ref[k].shape, k
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-4, atol=2e-5, err_msg=k)


def test_inference_on_expert_shards_gives_the_whole_models_columns():
    """the generators being keyed by global expert ids, an expert shard's inference (MC mean of nmc stochastic forwards, logits) equals the corresponding
    columns of the whole model's - a test() spread over the GPUs would only have to merge per-shard top-K lists"""
    ds = make_dataset("dblp", d=128, seed=12, n_rows=800, n_experts=2000)
    dims = [128, 128, ds["M"]]
    rows = np.arange(300, dtype=np.int64)
    full = _mk(ds, dims, True, 300, "uniform")
    p_full = full.forward(rows, nmc=3); z_full = full.logits(rows)
    shards = expert_shards(ds["M"], 3)
    got_p, got_z = [], []
    for s in shards:
        e = _mk(ds, dims, True, 300, "uniform", shard=s, world=3)
        got_p.append(e.forward(rows, nmc=3)); got_z.append(e.logits(rows)); e.close()
    p, z = np.concatenate(got_p, axis=1), np.concatenate(got_z, axis=1)
    assert p.shape == p_full.shape
    assert np.array_equal(z, z_full) and np.array_equal(p, p_full)


def test_expert_shards_at_config2_full_size_equal_the_whole_model():
    """BASELINE config 2's own shapes ([128, 128, 233 629], B = 1000) cut into 8 expert shards - what `--parallel ep` runs on an 8-GPU node, here on one GPU:
    the first update of the output layer equals the whole-model engine's (whose step is checked against the oracle at this size in test_gpu_round3.py) to 1 ulp,
    the hidden layer's to rounding of the d(hidden) sum"""
    ds = make_dataset("dblp", d=128, seed=0, n_rows=20_000)
    dims = [128, 128, ds["M"]]
    assert ds["M"] == 233_629
    order = np.random.default_rng(3).permutation(ds["N"])[:2000].astype(np.int64)
    shards = expert_shards(ds["M"], 8)
    full = _mk(ds, dims, True, 1000, "uniform")
    eng = [_mk(ds, dims, True, 1000, "uniform", shard=s, world=8) for s in shards]
    l_full, l_ep = _full_epoch(full, order[:1000], 1000), _ep_epoch(eng, order[:1000], 1000)
    assert abs(l_ep - l_full) <= 2e-6 * abs(l_full)
    a, b = _gathered(eng), full.state_dict()
    for k in b:
        # a shard's 114 expert tiles do not fill the chip: its dW kernel splits the K (batch) range over two workgroups per tile - another summation order (1 ulp in 2 % of mu);
        # Adam's first step is lr * g / (|g| + eps): where |g| ~ eps a rounding difference of g moves the update by up to 2 lr
        d = np.abs(a[k] - b[k])
        assert float((d > 1e-5 * np.abs(b[k]) + 2e-6).mean()) <= 1e-3 and float(d.max()) <= 2.1e-3, (k, float((d > 1e-5 * np.abs(b[k]) + 2e-6).mean()), float(d.max()))
    l_full2, l_ep2 = _full_epoch(full, order[1000:], 1000), _ep_epoch(eng, order[1000:], 1000)       # a second step: on prefetched operands in every shard
    assert abs(l_ep2 - l_full2) <= 1e-5 * abs(l_full2)
    for e in eng + [full]: e.close()
