"""Round 3: the step's head.  The dW + Adam kernel of a fused train step also writes the NEXT step's Flipout operands from the parameters it has just
updated (FusedDw.produce, include/opentf_amd.h ntf_prefetched_steps); sampler / sign words / loss reduction run on side streams.  Same results as the
stand-alone producer at the head of every step (bayesian-torch LinearFlipout.forward + kl_loss, called at src/mdl/fnn.py:126,136)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from opentf_amd.synth import make_dataset, init_params       # noqa: E402
from test_gpu_ep import _mk, _full_epoch, _ep_epoch, _gathered  # noqa: E402
from opentf_amd.ep import expert_shards                      # noqa: E402


def _run(ds, dims, order, B, monkeypatch, prefetch, ep=0):
    monkeypatch.setenv("NTF_PREFETCH", prefetch)
    if ep:
        engines = [_mk(ds, dims, True, B, "uniform", shard=s, world=ep) for s in expert_shards(ds["M"], ep)]
        l1 = _ep_epoch(engines, order, B)
        v = _ep_epoch(engines, order[:B], B, train=False)
        l2 = _ep_epoch(engines, order[::-1].copy(), B)
        out = (l1, v, l2, _gathered(engines), sum(e.prefetched_steps() for e in engines))
        for e in engines: e.close()
        return out
    e = _mk(ds, dims, True, B, "uniform")
    l1 = _full_epoch(e, order, B)                           # train steps back to back: every step after the first starts on prefetched operands
    v = _full_epoch(e, order[:B], B, train=False)           # an eval step right behind a train step consumes them too (same step counter, same parameters)
    sd = e.state_dict(); e.load_state_dict(sd)              # parameters touched from outside: the prefetched operands are stale and must not be used
    l2 = _full_epoch(e, order[::-1].copy(), B)
    p = e.forward(order[:64], nmc=2)                        # inference overwrites the operand buffers ...
    l3 = _full_epoch(e, order[:B], B)                       # ... so this step makes its own
    out = (l1, v, l2, l3, p, e.state_dict(), e.prefetched_steps())
    e.close()
    return out


@pytest.mark.parametrize("M", [70_000, 3000])               # several rounds of the dW kernel / few tiles: its split-K form (the finish kernel produces)
def test_operands_written_by_the_adam_epilogue_equal_the_producers(M, monkeypatch):
    ds = make_dataset("dblp", d=128, seed=9, n_rows=3000, n_experts=M)
    dims = [128, 64, 128, ds["M"]]
    order = np.random.default_rng(2).permutation(ds["N"])[:2500].astype(np.int64)
    a = _run(ds, dims, order, 1000, monkeypatch, "0")
    b = _run(ds, dims, order, 1000, monkeypatch, "1")
    assert a[-1] == 0 and b[-1] == 2 + 1 + 2, (a[-1], b[-1])   # epoch 1: steps 2, 3; the eval step; epoch 2: steps 2, 3 (its first follows load_state_dict)
    for x, y in zip(a[:4], b[:4]): assert abs(x - y) <= 1e-9 * abs(x), (x, y)
    assert np.array_equal(a[4], b[4])
    for k in a[5]: assert np.array_equal(a[5][k], b[5][k]), k


def test_operands_written_by_the_adam_epilogue_on_expert_shards(monkeypatch):
    """an expert shard's dW + Adam kernel (phase 2 of ntf_step_staged_ep) produces its rows of the next step's operands, eps keyed by GLOBAL element ids"""
    ds = make_dataset("dblp", d=128, seed=3, n_rows=1500, n_experts=3000)
    dims = [128, 128, ds["M"]]
    order = np.random.default_rng(4).permutation(ds["N"])[:600].astype(np.int64)
    a = _run(ds, dims, order, 200, monkeypatch, "0", ep=3)
    b = _run(ds, dims, order, 200, monkeypatch, "1", ep=3)
    assert a[-1] == 0 and b[-1] > 0
    for x, y in zip(a[:3], b[:3]): assert abs(x - y) <= 1e-9 * abs(x), (x, y)
    for k in a[3]: assert np.array_equal(a[3][k], b[3][k]), k


# ------------------------------------------------------------------------------------------ a step of the fully prefetched path that falls back to the exact-f32 kernels
@pytest.mark.parametrize("what", ["sigma", "hidden", "in_range"])
def test_flagged_steps_on_the_prefetched_path_equal_the_f32_engine(what, monkeypatch):
    """The f32 copy of sigma * eps that only a range-fallback step reads: the lean dW epilogue writes none, so a step on prefetched operands makes it when its flag is
    raised - since round 5 as extra workgroups of the previous step's bias Adam launch (NTF_F32_COPY_MERGED=1, default: the flag of the NEXT step is complete there - dW
    epilogue and prefetched head are both behind it), before that as a conditional launch of its own in front of the step (0).  Staged train steps with prefetched
    operands AND prefetched head whose flag is raised by the dW epilogue's producer ('sigma': rho = 80) or by the prefetched head ('hidden': activations of 5 000): every
    step counted as a fallback, the two forms bit for bit, and both the step an mfma = 'f32' engine takes (to rounding: that engine's hidden layer is the chain of kernels,
    not k_head - another summation order).  In range nothing falls back."""
    ds = make_dataset("dblp", d=128, seed=11, n_rows=2000, n_experts=3000)
    dims = [128, 128, ds["M"]]
    B = 256
    order = np.random.default_rng(3).permutation(ds["N"])[:4 * B].astype(np.int64)
    sd = init_params(dims, True, 0)
    if what == "sigma": sd["layers.1.rho_weight"][:] = 80.0
    if what == "hidden": sd["layers.0.mu_bias"][:] = 5000.0
    res = {}
    for name, mfma, merged in (("merged", None, "1"), ("own_launch", None, "0"), ("f32", "f32", "1")):
        monkeypatch.setenv("NTF_F32_COPY_MERGED", merged)
        e = _mk(ds, dims, True, B, "uniform", mfma=mfma)
        e.load_state_dict(sd)
        loss = _full_epoch(e, order, B)                 # four staged train steps: steps 2-4 on the previous epilogue's operands with the head that ran beside it
        res[name] = (loss, e.state_dict(), e.range_fallbacks(), e.prefetched_steps(), e.head_prefetch_hits()); e.close()
    a, a0, b = res["merged"], res["own_launch"], res["f32"]
    assert a[3] == 3 and a[4] == 3 and a0[3] == 3 and a0[4] == 3, (a[3], a[4])
    assert a[2] == a0[2] == (0 if what == "in_range" else 4) and b[2] == 0, (a[2], a0[2], b[2])
    assert np.isfinite(a[0]) and a[0] == a0[0]
    for k in a[1]: assert np.array_equal(a[1][k], a0[1][k]), k
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]), (a[0], b[0])
    if what != "in_range":      # (a flagged step IS the f32 engine's step but for the hidden layer's summation order)
        for k in a[1]:
            bad = ~np.isclose(a[1][k], b[1][k], rtol=1e-4, atol=2e-5)
            assert bad.mean() <= 2e-4, (k, float(np.abs(a[1][k] - b[1][k]).max()))


# ------------------------------------------------------------------------------------------ the evaluation steps of one ntf_eval_epoch call
@pytest.mark.parametrize("what", ["in_range", "mu", "sigma"])
def test_evaluation_steps_of_one_epoch_call_share_the_output_layers_kl_and_mu_planes(what):
    """Round 5: inside ntf_eval_epoch the steps run back to back on unchanged parameters - the first producer launch keeps the output layer's KL term and the range verdict
    on the planes of mu, the later ones draw fresh eps only (no planes of mu, no KL pass; src/mdl/fnn.py:118-140's validation loop draws eps per batch and adds the same
    kl / b to every batch loss).  The epoch loss equals the mean of the same steps made one by one (which take no such shortcut) and an mfma = 'f32' engine's; a mu outside
    the fp16 window ('mu': the verdict is the first step's) or a sigma * eps outside it ('sigma': every step's own) sends EVERY step to the exact-f32 kernels."""
    ds = make_dataset("dblp", d=128, seed=12, n_rows=2000, n_experts=3000)
    dims = [128, 128, ds["M"]]
    B = 256
    order = np.random.default_rng(5).permutation(ds["N"])[:3 * B + 100].astype(np.int64)
    sd = init_params(dims, True, 0)
    if what == "mu": sd["layers.1.mu_weight"][17, 5] = 300.0
    if what == "sigma": sd["layers.1.rho_weight"][:] = 80.0
    e = _mk(ds, dims, True, B, "uniform"); e.load_state_dict(sd)
    l_epoch = e.eval_epoch(order, B); fb_epoch = e.range_fallbacks()
    e.set_seed(5, 0); e.stage_order(order)
    one_by_one = [e.step_staged(o, min(B, len(order) - o), train=False, apply=False, want_loss=True) for o in range(0, len(order), B)]
    fb_steps = e.range_fallbacks() - fb_epoch; e.close()
    f = _mk(ds, dims, True, B, "uniform", mfma="f32"); f.load_state_dict(sd)
    l_f32 = f.eval_epoch(order, B); f.close()
    assert fb_epoch == fb_steps == (0 if what == "in_range" else 4), (fb_epoch, fb_steps)
    assert np.isfinite(l_epoch) and abs(l_epoch - float(np.mean(one_by_one))) <= 2e-7 * abs(l_epoch), (l_epoch, float(np.mean(one_by_one)))
    assert abs(l_epoch - l_f32) <= 2e-5 * abs(l_f32), (l_epoch, l_f32)


# ------------------------------------------------------------------------------------------ head prefetch (round 4)
@pytest.mark.parametrize("nsd,bayesian,multihot", [("uniform", True, False), ("unigram", True, False), ("uniform", False, False), ("unigram", True, True)])
def test_head_run_beside_the_previous_steps_dw_kernel_equals_the_head_in_its_own_step(nsd, bayesian, multihot, monkeypatch):
    """ntf_head_prefetch_hits: sampler, s_out words, gather -> hidden layer -> h images of batch t + 1 issued on the side stream of step t (behind its hidden-layer
    backward and Adam, into the other workspace set, KL terms and range flag in the next step's slots) against the same work at the head of step t + 1: the same
    kernels on the same inputs - parameters bit for bit, losses to the order of the KL sum's double atomics.  The sequence walks every way out of the fast path: a
    ragged last batch, an evaluation step behind a train step (head redone without its KL terms), parameters rewritten from outside, inference in between."""
    ds = make_dataset("dblp", d=128, seed=9, n_rows=3000, n_experts=70_000)
    dims = [ds["S"] if multihot else 128, 128, ds["M"]]      # (multihot, round 6: BASELINE config 3's input - the head is the chain head_launch_multihot issues, behind the first layer's sweep)
    order = np.random.default_rng(2).permutation(ds["N"])[:2500].astype(np.int64)
    out = []
    for hp in ("0", "1"):
        monkeypatch.setenv("NTF_HEAD_PREFETCH", hp)
        e = _mk(ds, dims, bayesian, 1000, nsd, multihot=multihot)
        l1 = _full_epoch(e, order, 1000)                        # 1000, 1000, 500 rows: steps 2 and 3 find their head done
        v = _full_epoch(e, order[:700], 1000, train=False)      # (the train step before it had no next batch: nothing was issued)
        l2 = _full_epoch(e, order[::-1].copy(), 1000)
        v2 = [e.eval_step(order[:300]), e.eval_step(order[300:900])]     # an eval step right behind a train step that DID issue a head? no: the epoch's last step issues none
        sd = e.state_dict(); e.load_state_dict(sd)              # parameters touched from outside: prefetched operands and head are stale
        l3 = _full_epoch(e, order[:2000], 1000)
        p = e.forward(order[:64], nmc=2) if bayesian else e.forward(order[:64])
        l4 = _full_epoch(e, order[500:], 1000)
        out.append(((l1, v, l2, *v2, l3, l4), p, e.state_dict(), e.head_prefetch_hits())); e.close()
    (la, pa, sa, ha), (lb, pb, sb, hb) = out
    assert ha == 0 and hb == 2 + 2 + 1 + 1, (ha, hb)      # (round 6: Fnn steps run the same pipeline - planes of the updated mu from the dW epilogue, hidden backward and the next head on the side stream)
    if multihot:      # (the first layer's gradient is a scatter of f32 atomic row adds: the order of its sums differs from run to run, with or without the prefetch)
        # ... and Adam turns a last-bit difference of a near-zero gradient into a +-lr step of that weight (tests/test_gpu_replay.py, the same band): a fraction of the elements, bounded
        for x, y in zip(la, lb): assert abs(x - y) <= 2e-5 * abs(x), (x, y)
        np.testing.assert_allclose(pa, pb, rtol=5e-3, atol=1e-5)
        for k in sa:
            bad = np.abs(sa[k] - sb[k]) > (1e-3 * np.abs(sb[k]) + 2e-5)
            assert float(bad.mean()) <= 1e-3, (k, float(bad.mean()))
        return
    for x, y in zip(la, lb): assert abs(x - y) <= 1e-9 * abs(x), (x, y)
    assert np.array_equal(pa, pb)
    for k in sa: assert np.array_equal(sa[k], sb[k]), k


def test_head_prefetch_is_dropped_when_the_next_call_is_another_batch(monkeypatch):
    """a train step issues the head of the batch that FOLLOWS in the staged order; the caller then steps something else (an evaluation batch, the same batch again):
    the prefetched head is not used, its KL terms are not counted twice, the results are those of a run without prefetch"""
    ds = make_dataset("dblp", d=128, seed=4, n_rows=2500, n_experts=20_000)
    dims = [128, 128, ds["M"]]
    order = np.random.default_rng(5).permutation(ds["N"])[:2000].astype(np.int64)
    out = []
    for hp in ("0", "1"):
        monkeypatch.setenv("NTF_HEAD_PREFETCH", hp)
        e = _mk(ds, dims, True, 500, "uniform")
        e.stage_order(order)
        ls = [e.step_staged(0, 500, train=True, apply=True, want_loss=True),         # issues the head of rows 500 .. 999
              e.step_staged(1000, 500, train=True, apply=True, want_loss=True),      # ... but the caller jumps: head redone, KL counted once; issues 1500 .. 1999
              e.step_staged(1500, 500, train=False, apply=False, want_loss=True),    # an evaluation step on the very batch that was prefetched for training: not taken (train only)
              e.step_staged(1500, 500, train=True, apply=True, want_loss=True)]      # (no next batch: nothing issued)
        # a new order staged into the same device buffer: the batch at the prefetched OFFSET is another batch now
        e.stage_order(order[:1500]); e.step_staged(0, 500, train=True, apply=True)           # issues the head of rows 500 .. 999 of this order
        e.stage_order(order[::-1].copy())
        ls.append(e.step_staged(500, 500, train=True, apply=True, want_loss=True))          # same offset, same size, other rows: must not be taken
        out.append((ls, e.state_dict(), e.head_prefetch_hits())); e.close()
    (la, sa, ha), (lb, sb, hb) = out
    assert ha == 0 and hb == 0
    for x, y in zip(la, lb): assert abs(x - y) <= 1e-9 * abs(x), (x, y)
    for k in sa: assert np.array_equal(sa[k], sb[k]), k


# ------------------------------------------------------------------------------------------ the one-kernel head (ntf_head.hip)
@pytest.mark.parametrize("bayesian", [True, False])
@pytest.mark.parametrize("d,dense,B", [(128, False, 1000), (64, False, 333), (256, False, 77), (128, True, 500)])
def test_one_kernel_head_equals_the_kernel_chain(bayesian, d, dense, B, monkeypatch):
    """gather -> hidden Flipout layer -> h images in ONE kernel (NTF_HEAD, default on) against the chain of kernels it replaces (k_gather_pool, k_flipout_perturb x 2,
    k_gemm x 2, k_prep_h, k_prep_planes_T): the same generators and arithmetic, another summation order in the 128 x d products - loss, every gradient and the
    mean-pooled input agree to rounding; ragged last row blocks, d = 64 / 128 / 256, dense input rows"""
    from opentf_amd import libntf
    ds = make_dataset("dblp", d=d, seed=21, n_rows=1200, n_experts=3000)
    dims = [d, 128, ds["M"]]
    rows = np.random.default_rng(1).permutation(ds["N"])[:B].astype(np.int64)
    X = None
    if dense:
        from oracle import ntf_oracle as O
        X = O.gather_meanpool_fast(ds["skill"][0], ds["skill"][1], ds["table"])
    out = []
    for head in ("0", "1"):
        monkeypatch.setenv("NTF_HEAD", head)
        e = libntf.Engine(dims, bayesian=bayesian, input_mode=libntf.INPUT_DENSE if dense else libntf.INPUT_MEANPOOL, max_batch=1000, ns=5, nsd="uniform",
                          tpw=10.0, tnw=1.0, lr=1e-3, seed=5, fuse_adam=0)
        if dense: e.set_dense_input(X)
        else: e.set_skill_table(ds["table"]); e.set_skill_csr(ds["skill"])
        e.set_member(ds["member"]); e.load_state_dict(init_params(dims, bayesian, 0))
        ev = e.eval_step(rows)
        loss = e.backward(rows)
        g = e.grads()
        l2 = e.train_step(rows)          # and one more through the fused-Adam-free apply
        out.append((ev, loss, l2, g, e.state_dict())); e.close()
    a, b = out
    for x, y in zip(a[:3], b[:3]): assert abs(x - y) <= 2e-6 * abs(x), (x, y)
    for k in a[3]:
        tol = 2e-5 * float(np.abs(a[3][k]).max()) + 1e-12
        assert float(np.abs(a[3][k] - b[3][k]).max()) <= tol, (k, float(np.abs(a[3][k] - b[3][k]).max()), tol)


# ------------------------------------------------------------------------------------------ BASELINE config 2 at FULL size against the oracle
def _full_size_oracle_step(D, H, M, B, S, mean_s, mean_m, multihot=False, nsd="uniform", seed=11, bayesian=True):
    """one Bnn (bayesian = False: Fnn) train step of src/mdl/fnn.py:122-140 at full size, every random tensor injected (eps, s_in, s_out, negatives), against oracle/ntf_oracle.py (torch CPU,
    autograd): logits element-wise on a 64-row slice and in the max norm over all B x M (1e-4), loss (2e-5), every gradient (3e-4 of its max, with a budget of
    leaky_relu' kink flips on the output layer), and the parameters after the fused dW + Adam kernel.  multihot: the input is the teams' multi-hot skill rows
    (D = S, the first layer a CSR gather-sum of W0 columns on the device, a dense [B, S] product in the oracle)."""
    import torch
    from conftest import draw_noise
    from oracle import ntf_oracle as O
    from opentf_amd import libntf
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    sd = O.bnn_init(D, [H], M) if bayesian else O.fnn_init(D, [H], M)
    nnz = np.minimum(1 + rng.poisson(mean_s - 1, B), S)
    s_ip = np.concatenate([[0], np.cumsum(nnz)]).astype(np.int64)
    s_ix = np.concatenate([np.sort(rng.choice(S, k, replace=False)) for k in nnz]).astype(np.int32)
    if multihot:
        assert D == S
        table = None
        Xn = np.zeros((B, S), np.float32); Xn[np.repeat(np.arange(B), nnz), s_ix.astype(np.int64)] = 1.0
        X = torch.from_numpy(Xn); del Xn
    else:
        table = rng.standard_normal((S, D)).astype(np.float32)
        X = torch.from_numpy(O.gather_meanpool_fast(s_ip, s_ix, table))
    mn = 1 + rng.poisson(mean_m - 1, B)
    m_ip = np.concatenate([[0], np.cumsum(mn)]).astype(np.int64)
    m_ix = np.concatenate([np.sort(rng.choice(M, k, replace=False)) for k in mn]).astype(np.int32)
    y = torch.zeros(B, M)
    y[np.repeat(np.arange(B), mn), m_ix.astype(np.int64)] = 1.0
    noise = draw_noise(sd, B) if bayesian else None
    if nsd == "unigram":      # fnn.py:58-72 with a Zipf-like frequency table over ALL teams (the table itself is pinned by g3; here it only draws the injected indices)
        freq = 1.0 / (np.arange(M) + 10.0); freq = torch.tensor(freq / freq.sum()).reshape(1, M)
        neg = O.ns_unigram(y, freq, 5)
    else: neg = O.ns_uniform(y, 5)
    inj = {"neg_idx": neg.numpy()}
    if bayesian: inj.update({"eps_w": [n["eps_w"] for n in noise], "eps_b": [n["eps_b"] for n in noise], "s_in": [n["s_in"] for n in noise], "s_out": [n["s_out"] for n in noise]})
    rows = np.arange(B)

    def engine(fuse_adam):
        e = libntf.Engine([D, H, M], bayesian=bayesian, input_mode=libntf.INPUT_MULTIHOT if multihot else libntf.INPUT_MEANPOOL, max_batch=B, ns=5, nsd="uniform", tpw=10.0, tnw=1.0,
                          lr=1e-3, fuse_adam=fuse_adam)
        if not multihot: e.set_skill_table(table)
        e.set_skill_csr((s_ip, s_ix)); e.set_member((m_ip, m_ix)); e.load_state_dict(sd)
        return e

    e = engine(0)
    ref_logits = O.model_forward(sd, X, noise).detach().numpy()
    got = e.logits(rows, inject=inj)
    assert got.shape == ref_logits.shape == (B, M)
    sl = np.arange(7, B, 16)[:64]
    zmax = float(np.abs(ref_logits).max())
    np.testing.assert_allclose(got[sl], ref_logits[sl], rtol=1e-4, atol=max(2e-6, 1e-6 * zmax))     # (leaky_relu's negative side: |logit| ~ 1e-2 |z|)
    assert float(np.abs(got - ref_logits).max()) <= 1e-4 * zmax
    del got, ref_logits
    sd_ref = {k: v.clone() for k, v in sd.items()}
    ref_loss, ref_grads = O.train_step(sd_ref, O.Adam(sd_ref, 1e-3), X, y, neg, 10.0, 1.0, noise)     # sd_ref now holds the oracle's updated parameters
    loss = e.backward(rows, inject=inj)
    assert abs(loss - ref_loss) <= 2e-5 * abs(ref_loss), (loss, ref_loss)
    grads = e.grads()
    for k in sd:
        ref = ref_grads[k].numpy()
        d = np.abs(grads[k] - ref)
        tol = 3e-4 * float(np.abs(ref).max())
        if k.startswith("layers.1."):
            # |z| within rounding of 0 lands on either side of leaky_relu's kink in another summation order: one (row, expert) pair flips, moving that expert's
            # gradient row (128 elements of the weight tensors, one of the bias tensors) by up to 0.99 |dz| |h|; ~1e-7 of the B x M logits
            assert int((d > tol).sum()) <= 64 * (H if k.endswith("weight") else 1), (k, int((d > tol).sum()))
        else:
            assert float(d.max()) <= tol, (k, float(d.max()), tol)
    e.close()
    # the default path: dW + Adam fused (the update happens inside the kernel; gradients of the output layer are never written)
    e = engine(1)
    loss1 = e.train_step(rows, inject=inj)
    assert abs(loss1 - ref_loss) <= 2e-5 * abs(ref_loss)
    st = e.state_dict(); e.close()
    for k in sd:
        a, b = st[k], sd_ref[k].numpy()
        bad = np.abs(a - b) > (1e-3 * np.abs(b) + 2e-5)
        # Adam's first step is lr * g / (|g| + eps): where |g| ~ 1e-8 .. a rounding difference in g moves the update by up to 2 lr
        assert float(bad.mean()) <= 2e-4, (k, float(bad.mean()))


def test_config2_full_size_step_against_the_oracle():
    """VERDICT r2 missing #4: at [128, 128, 233 629], B = 1000 (dblp mt10.ts2, mean-pooled d = 128 table) the HIP path had only been compared with the repo's own generic path"""
    _full_size_oracle_step(D=128, H=128, M=233_629, B=1000, S=4000, mean_s=8.57, mean_m=3.06)


def test_config2_full_size_fnn_step_against_the_oracle():
    """the same shapes with the non-Bayesian model (src/mdl/fnn.py alone: one matrix, no Flipout operands) - the other half of the fnn / bnn path"""
    _full_size_oracle_step(D=128, H=128, M=233_629, B=1000, S=4000, mean_s=8.57, mean_m=3.06, seed=15, bayesian=False)


def test_config4_full_size_step_against_the_oracle():
    """BASELINE config 4 at its size: uspt mt10.ts2 (M = 394 187, 6.29 skills / 2.51 members per team), d = 256 table, layer 0 256 -> 128"""
    _full_size_oracle_step(D=256, H=128, M=394_187, B=1000, S=6000, mean_s=6.29, mean_m=2.51, seed=12)


def test_config5_full_expert_count_step_against_the_oracle():
    """BASELINE config 5's expert axis at its size: gith unfiltered (M = 1 369 895, 1.37 skills / 5.53 members per team, S = 486) - with B = 200 rows so that the
    oracle's dense [B, M] tensors stay at 1.1 GB each"""
    _full_size_oracle_step(D=128, H=128, M=1_369_895, B=200, S=486, mean_s=1.37, mean_m=5.53, seed=14)


def test_config3_full_size_step_against_the_oracle():
    """BASELINE config 3 at its size: dblp mt10.ts2 with the MULTI-HOT input (D = S = 90 671: the first layer is a CSR gather-sum of W0 columns), negatives drawn by
    the reference's unigram rule"""
    _full_size_oracle_step(D=90_671, H=128, M=233_629, B=1000, S=90_671, mean_s=8.57, mean_m=3.06, multihot=True, nsd="unigram", seed=13)


def _host_gib_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"): return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def test_dblp_unfiltered_expert_count_step_against_the_oracle():
    """VERDICT r3 missing #3: north_star says "the full DBLP sparse matrix" - M = 5 022 955 experts (output/dblp/dblp.v12.json/prep.teamsvecs.log:18).  At that size the HIP
    path had only been compared with the repo's own generic path (which shared a -335 ppm loss bug with it in round 2).  One injected-noise oracle step at B = 48: the
    oracle's dense [B, M] tensors are 0.96 GB each, its [M, H] tensors 2.6 GB each (~70 GB of host memory at the peak)."""
    if _host_gib_available() < 110: pytest.skip("needs ~70 GB of host memory for the oracle's dense tensors")
    _full_size_oracle_step(D=128, H=128, M=5_022_955, B=48, S=4000, mean_s=8.57, mean_m=3.06, seed=16)


def test_uspt_unfiltered_expert_count_step_against_the_oracle():
    """the same for uspt's unfiltered matrix: M = 3 508 807 (output/uspt/patent.tsv/prep.teamsvecs.log:34), d = 256 table, B = 64"""
    if _host_gib_available() < 90: pytest.skip("needs ~50 GB of host memory for the oracle's dense tensors")
    _full_size_oracle_step(D=256, H=128, M=3_508_807, B=64, S=6000, mean_s=6.29, mean_m=2.51, seed=17)


# ------------------------------------------------------------------------------------------ the evaluation-loss kernel (k_out_fwd_h3e, round 6)
@pytest.mark.parametrize("bayesian", [True, False])
@pytest.mark.parametrize("M,B", [(70_000, 1000), (3000, 333), (70_001, 129), (40, 70), (233_629, 257)])     # ragged / empty last sub-tile, a ragged and a half-empty 256-row block, one tile only
def test_eval_loss_kernel_equals_the_forward_only_kernel_of_round_5(bayesian, M, B, monkeypatch):
    """k_out_fwd_h3e (eight staggered logit waves on 256 rows) against k_out_fwd_b6<.., TRAIN = false> (NTF_EVAL_KERNEL=0): the loss of evaluation steps on the device's
    own draws - the same fp16x3 products and logit arithmetic, the row sums taken in another order (2e-6) - one by one, as an epoch call, and behind a train step
    (prefetched operands); then against the exact-f32 engine (2e-5)."""
    ds = make_dataset("dblp", d=128, seed=14, n_rows=2600, n_experts=M)
    dims = [128, 128, ds["M"]]
    order = np.random.default_rng(8).permutation(ds["N"])[:2 * B + B // 3].astype(np.int64)
    out = []
    for k, mfma in (("0", None), ("2", None), ("2", "f32")):      # (2: the kernel also for the non-Bayesian model, whose default stays k_out_fwd_b6 - it measured faster there)
        monkeypatch.setenv("NTF_EVAL_KERNEL", k)
        e = _mk(ds, dims, bayesian, B, "uniform", mfma=mfma)
        ls = [e.eval_step(order[:B]), e.eval_step(order[B:2 * B]), e.eval_epoch(order, B)]
        e.train_step(order[:B]); ls.append(e.eval_step(order[B:2 * B]))
        assert e.range_fallbacks() == 0
        out.append(ls); e.close()
    for x, y in zip(out[0], out[1]): assert np.isfinite(x) and abs(x - y) <= 2e-6 * abs(x), (x, y)
    for x, y in zip(out[1][:3], out[2][:3]): assert abs(x - y) <= 2e-5 * abs(y), (x, y)      # (behind the train step the two arithmetics' parameters differ)


# ------------------------------------------------------------------------------------------ inference (Fnn.test, src/mdl/fnn.py:172-219) at BASELINE config 2's expert count
@pytest.mark.parametrize("bayesian", [True, False])
def test_config2_full_size_inference_against_the_oracle(bayesian):
    """the test() path at M = 233 629 (256 teams per call so that the oracle's [nmc, B, M] tensor stays at 0.7 GB): MC-mean probabilities of the fused PROBS-mode kernel,
    predictive entropy and mutual information against oracle/ntf_oracle.py with every MC pass's noise injected; the device top-K (K = 100) against a stable sort of
    the engine's own probabilities for the deterministic model"""
    import torch
    from conftest import draw_noise
    from oracle import ntf_oracle as O
    from opentf_amd import libntf
    D, H, M, B, S, nmc = 128, 128, 233_629, 256, 4000, 3
    torch.manual_seed(21)
    rng = np.random.default_rng(21)
    sd = O.bnn_init(D, [H], M) if bayesian else O.fnn_init(D, [H], M)
    if not bayesian: nmc = 1
    table = rng.standard_normal((S, D)).astype(np.float32)
    nnz = 1 + rng.poisson(7.57, B)
    s_ip = np.concatenate([[0], np.cumsum(nnz)]).astype(np.int64)
    s_ix = np.concatenate([np.sort(rng.choice(S, k, replace=False)) for k in nnz]).astype(np.int32)
    X = torch.from_numpy(O.gather_meanpool_fast(s_ip, s_ix, table))
    m_ip = np.arange(B + 1, dtype=np.int64); m_ix = rng.integers(0, M, B).astype(np.int32)
    e = libntf.Engine([D, H, M], bayesian=bayesian, input_mode=libntf.INPUT_MEANPOOL, max_batch=B, ns=5, nsd="uniform", tpw=10.0, tnw=1.0, lr=1e-3)
    e.set_skill_table(table); e.set_skill_csr((s_ip, s_ix)); e.set_member((m_ip, m_ix)); e.load_state_dict(sd)
    noises = [draw_noise(sd, B) for _ in range(nmc)] if bayesian else None
    injs = [{"eps_w": [n["eps_w"] for n in nz], "eps_b": [n["eps_b"] for n in nz], "s_in": [n["s_in"] for n in nz], "s_out": [n["s_out"] for n in nz]} for nz in noises] if bayesian else None
    mc = O.predict(sd, X, nmc, noises).numpy()
    mc = mc if mc.ndim == 3 else mc[None]
    probs, pu, mu = e.forward(np.arange(B), nmc=nmc, injects=injs, uncertainty=True)
    np.testing.assert_allclose(probs, mc.mean(0), rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(pu, O.predictive_entropy(mc), rtol=1e-4, atol=1e-4)
    if bayesian: np.testing.assert_allclose(mu, O.mutual_information(mc), rtol=1e-3, atol=2e-4)
    else:
        vals, idx = e.forward_topk(np.arange(B), 100, nmc=1)
        order = np.argsort(-probs, axis=1, kind="stable")[:, :100]
        assert np.array_equal(idx, order) and np.array_equal(vals, np.take_along_axis(probs, order, axis=1))
    e.close()
