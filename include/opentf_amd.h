/* opentf_amd.h — C ABI of the MI355X-native engine for OpeNTF's fnn/bnn minibatch hot path.
 *
 * The reference (fani-lab/OpeNTF) is pure Python and has no FFI; the boundary it offers is the model
 * plugin API of src/mdl/ntf.py:5-31 (constructor, learn, test).  This library is what a C-ABI
 * replacement of the body of that API binds (last row).  Each entry point names the
 * reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative NTF_E* code on failure; ntf_last_error() gives
 *     the message (per engine; for a failed create, pass NULL).  No exceptions cross the ABI.
 *   - plain pointers and sizes only.  "host" pointers are ordinary host memory, copied in/out.
 *     "dev" pointers are HBM addresses owned by the engine (exposed for RCCL all-reduce).
 *   - an engine is bound to one GPU and one HIP stream; it is not thread-safe; no global state.
 *   - there is no CPU fallback: without a usable HIP device ntf_engine_create fails.
 */
#ifndef OPENTF_AMD_H
#define OPENTF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTF_ABI_VERSION 1
#define NTF_MAX_LAYERS 8

enum { NTF_OK = 0, NTF_EINVAL = -1, NTF_EHIP = -2, NTF_ESTATE = -3, NTF_ENOMEM = -4 };

/* how a minibatch's input rows X[B, D] are produced on the device */
enum ntf_input_mode {
    NTF_INPUT_DENSE = 0,   /* rows of a resident dense [N, D] f32 matrix (D2v vectors; src/mdl/ntf.py:24 else-branch) */
    NTF_INPUT_MEANPOOL = 1,/* mean of the team's skill-embedding rows: CSR(skill) @ E / nnz (src/mdl/emb/gnn.py:484-486) */
    NTF_INPUT_MULTIHOT = 2 /* multi-hot skill row (src/mdl/ntf.py:23); layer 0 becomes a CSR gather-sum of W0 columns */
};

/* negative sampling distribution, src/mdl/fnn.py:34-36 */
enum ntf_nsd { NTF_NSD_NONE = 0, NTF_NSD_UNIFORM = 1, NTF_NSD_UNIGRAM = 2, NTF_NSD_UNIGRAM_B = 3 };

/* parameter kinds of one layer; Fnn uses WEIGHT/BIAS, Bnn all four in the state_dict order
 * mu_weight, rho_weight, mu_bias, rho_bias (src/mdl/bnn.py:25 via bayesian-torch LinearFlipout) */
enum ntf_param_kind { NTF_P_WEIGHT = 0, NTF_P_BIAS = 1, NTF_P_RHO_WEIGHT = 2, NTF_P_RHO_BIAS = 3 };

enum ntf_mfma { NTF_MFMA_DEFAULT = 0, NTF_MFMA_F32 = 1, NTF_MFMA_BF16X6_RETIRED = 2 /* rejected by ntf_engine_create since round 6 */, NTF_MFMA_FP16X3 = 3 };

typedef struct ntf_config {
    int32_t abi_version;        /* NTF_ABI_VERSION */
    int32_t device;             /* HIP device ordinal ("cuda:N" of src/__config__.yaml:10) */
    void*   stream;             /* hipStream_t to run on, or NULL: the engine creates its own */
    int32_t n_layers;           /* number of Linear layers = len(h) + 1 (src/mdl/fnn.py:20-22) */
    int32_t dims[NTF_MAX_LAYERS + 1]; /* D, h[0], ..., h[-1], M */
    int32_t bayesian;           /* 0 = Fnn, 1 = Bnn/Flipout (src/mdl/bnn.py:17-27) */
    int32_t input_mode;         /* enum ntf_input_mode */
    int32_t max_batch;          /* cfg.b (src/mdl/__config__.yaml:1) */
    int32_t ns;                 /* cfg.ns */
    int32_t nsd;                /* enum ntf_nsd */
    float   tpw, tnw;           /* cfg.tpw, cfg.tnw (src/mdl/fnn.py:45) */
    float   lr;                 /* cfg.lr; Adam defaults beta 0.9/0.999 eps 1e-8 (src/mdl/fnn.py:104) */
    uint64_t seed;              /* seed of the device-side generators (negatives, Flipout eps and signs) */
    int32_t fused;              /* 1 = use the fused output-layer kernels when the shape allows, 0 = generic path */
    int32_t fuse_adam;          /* single-GPU train steps only.  0: one flat Adam kernel after backward.  1: the output layer's Adam runs inside
                                   the dW kernel's epilogue (its gradients are not materialised).  2: the dW kernel is launched in expert chunks
                                   and Adam of a finished chunk runs on a side stream beside the next chunk's dW */
    int32_t mfma;               /* arithmetic of the fused output-layer products: NTF_MFMA_DEFAULT (0) = NTF_MFMA_FP16X3 = each operand times an exact power of two
                                   split into two fp16 values (22 bits), three fp16 MFMA products per f32 product, f32 accumulate (error against f64 ~1.2x that of
                                   the f32 MFMA; operands outside the fp16 window send the step to the exact-f32 kernels, ntf_range_fallbacks);
                                   NTF_MFMA_F32 = v_mfma_f32_32x32x2_f32 (a bit-exact f32 fma chain).  2 (bf16x6, rounds 1-5) is rejected */
    /* Expert-sharded output layer (every field 0 = off).  This engine owns the experts [expert_lo, expert_lo + dims[n_layers]) of an
       output layer of `experts_global` experts that is split over `ep_world` engines (one per GPU); hidden layers are replicated.  Every engine
       steps the WHOLE minibatch: labels (member CSR) and sampled negatives keep global expert ids, the device generators are keyed by global
       ids, and the only exchange of a train step is the sum over engines of d(hidden) [B, h[-1]] between ntf_step_staged_ep phases 1 and 3.
       expert_lo must be a multiple of 256; needs the fused output-layer path (h[-1] in {32, 64, 128}). */
    int32_t expert_lo;
    int32_t experts_global;
    int32_t ep_world;
    int32_t reserved[2];
} ntf_config;

/* Random tensors of one step, injected instead of generated (parity tests).  Any pointer may be NULL
 * (= generate on device).  All are HOST pointers.  Layouts follow bayesian-torch LinearFlipout.forward:
 * eps_w[l] [out,in], eps_b[l] [out], s_in[l] [B,in] (+1/-1 as f32), s_out[l] [B,out] (+1/-1 as f32);
 * neg_idx [B, ns] int64 as returned by src/mdl/fnn.py:48-76. */
typedef struct ntf_inject {
    const int64_t* neg_idx;
    const float* eps_w[NTF_MAX_LAYERS];
    const float* eps_b[NTF_MAX_LAYERS];
    const float* s_in[NTF_MAX_LAYERS];
    const float* s_out[NTF_MAX_LAYERS];
} ntf_inject;

typedef struct ntf_engine ntf_engine;

/* ---- lifetime:  Fnn.init / Bnn.init + model.to(device)            src/mdl/fnn.py:15-30,100-102 */
int  ntf_engine_create(const ntf_config* cfg, ntf_engine** out);
void ntf_engine_destroy(ntf_engine* e);
const char* ntf_last_error(const ntf_engine* e);
int  ntf_abi_version(void);

/* ---- data residency (once per learn/test call): replaces NtfDataset's per-sample densify
 *      src/mdl/ntf.py:16-25.  CSR = int64 indptr [n_rows+1], int32 indices [nnz]; values are 1. */
int ntf_set_member_csr(ntf_engine* e, const int64_t* indptr, const int32_t* indices, int64_t n_rows);
int ntf_set_skill_csr(ntf_engine* e, const int64_t* indptr, const int32_t* indices, int64_t n_rows);
int ntf_set_skill_table(ntf_engine* e, const float* table, int64_t n_skills, int32_t d); /* E of gnn.py:485 */
int ntf_set_dense_input(ntf_engine* e, const float* X, int64_t n_rows, int32_t d);
int ntf_set_unigram(ntf_engine* e, const double* freq, int64_t n);                         /* src/mdl/fnn.py:82 */

/* ---- state:  state_dict() / load_state_dict()                     src/mdl/fnn.py:101,160,168,187 */
int ntf_set_param(ntf_engine* e, int layer, int kind, const float* host, int64_t count);
int ntf_get_param(ntf_engine* e, int layer, int kind, float* host, int64_t count);
int ntf_get_grad(ntf_engine* e, int layer, int kind, float* host, int64_t count); /* p.grad after backward, fnn.py:137.  Valid after ntf_backward (or a step with fuse_adam = 0):
                                                                                     a fused step (fuse_adam = 1, ntf_train_step) consumes the gradients of the tensors it updates in a kernel's
                                                                                     epilogue - the output layer's weight / rho_weight (never written), a multi-hot Flipout first layer's (read and
                                                                                     cleared by its one-pass update) - and this call then returns zeros or the previous backward's values for them */
/* d loss / d z [B, M] of the output layer (z = its pre-activation) as the last backward / train step left it: what autograd holds for
 * `y_` of src/mdl/fnn.py:132 before leaky_relu.  The fused kernels' only dense product, exposed so that it can be checked element-wise. */
int ntf_get_dlogits(ntf_engine* e, float* host, int64_t count);
/* The sampled negatives [B, ns] (GLOBAL expert ids) of the last step: `topk_indices` of src/mdl/fnn.py:48-76 as the device samplers drew them (or as they were injected).
 * Exposed so that the samplers can be checked draw by draw over a sequence of steps (batch support, distinctness, frequencies), not only through the loss. */
int ntf_get_negatives(ntf_engine* e, int64_t* host, int64_t count);
/* The device generators' own draws of ONE random tensor of bayesian-torch's LinearFlipout.forward (called from src/mdl/fnn.py:126,135 through mdl/bnn.py:17-27) for step
 * index `step` of this engine's seed - what a native, non-injected step with that index consumes: kind 0 eps_weight [out, in], 1 eps_bias [out], 2 sign_input
 * [rows, in], 3 sign_output [rows, out] (+1 / -1; row r = position r of the step's minibatch), in the reference's layouts.  Exposed so that a sequence of native
 * steps can be replayed through the oracle with exactly the tensors the kernels regenerate in place (forward sign words, the dW epilogue's re-drawn eps, the
 * operands the previous step's epilogue produced): tests/test_gpu_replay.py. */
int ntf_get_noise(ntf_engine* e, uint64_t step, int32_t layer, int32_t kind, int32_t rows, float* host, int64_t count);
int ntf_reset_optimizer(ntf_engine* e);                  /* fresh Adam per fold, src/mdl/fnn.py:104 */
int ntf_set_lr(ntf_engine* e, float lr);                 /* ReduceLROnPlateau result, fnn.py:105,163 */
int ntf_set_seed(ntf_engine* e, uint64_t seed, uint64_t step);
/* the device generators (Flipout eps, signs, sampled negatives) are keyed by (seed, step counter, global row position); every step call
 * advances the counter.  A data-parallel rank whose shard of a global minibatch is EMPTY makes no step call: it calls this instead, so
 * that all ranks keep drawing the same eps for the same global step. */
int ntf_skip_step(ntf_engine* e);
/* fp16x3 arithmetic (NTF_MFMA_FP16X3 / default) splits weights * 2^8 and activations * 2^4 into fp16 pairs: exact for |w| < 255.9,
 * |h| < 4094.  Operands are range-checked where they are split; a step (or inference call) in which one leaves that window runs on the
 * exact-f32 kernels instead - never on saturated values.  This reports how many steps / calls did so since the engine was created. */
int ntf_range_fallbacks(ntf_engine* e, int64_t* steps);
/* A fused train step's dW + Adam kernel holds the output layer's UPDATED mu / rho in its epilogue and writes, from them, the NEXT step's Flipout operands
 * (eps of step + 1, sigma * eps, the split planes, the layer's KL - bayesian-torch's LinearFlipout.forward / kl_loss, called at src/mdl/fnn.py:126,136) -
 * that step then starts without the operand producer's own pass over the layer.  Reports how many steps started that way (same results either way:
 * the arithmetic is the producer's; NTF_PREFETCH=0 in the environment turns it off). */
int ntf_prefetched_steps(ntf_engine* e, int64_t* steps);
/* ... and, behind its hidden-layer backward, its side stream runs what the NEXT batch of the staged order (ntf_step_staged walks it front to back, as the loader of
 * src/mdl/fnn.py:118 does) needs before its forward kernel: the negative sampler (fnn.py:48-72), the team2vec gather and the hidden layer (src/mdl/emb/gnn.py:485,
 * fnn.py:25) - beside the dW kernel instead of between two steps' big kernels.  Reports how many steps found their head done that way (same results either way;
 * NTF_HEAD_PREFETCH=0 turns it off; per-batch unigram_b tables, injected tensors, expert shards and data-parallel shards always run their head in their own step). */
int ntf_head_prefetch_hits(ntf_engine* e, int64_t* steps);
/* Multi-hot input with a Flipout first layer (BASELINE config 3: 90 671 x 128 mu / rho pairs, src/mdl/ntf.py:23 + src/mdl/fnn.py:25,136-139): in a fused train step the
 * layer's gradient finalisation (Flipout chain rule + KL), its Adam update and the NEXT step's sigma * eps + KL term are ONE pass over the layer beside the dW kernel
 * (launch_flipout_sweep) instead of three.  Reports how many steps started on a first-layer operand produced that way (same results either way; NTF_L0_SWEEP=0 turns it off). */
int ntf_first_layer_sweeps(ntf_engine* e, int64_t* steps);

/* ---- the step:  body of the hot loop                              src/mdl/fnn.py:118-151
 * rows = B global team ids (host).  loss_out may be NULL: then nothing is synchronised and the loss is
 * only added to the epoch accumulator (ntf_epoch_loss). */
int ntf_train_step(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, float* loss_out);
int ntf_eval_step(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, float* loss_out);
/* split form for data parallelism: backward leaves d(loss_sum_over_rows / global_B)/dparam in the flat
 * gradient buffer (sum over ranks = the single-process gradient); apply runs Adam. */
int ntf_backward(ntf_engine* e, const int64_t* rows, int32_t B, int32_t global_B, const ntf_inject* inj, float* loss_out);
int ntf_apply(ntf_engine* e);
/* one Adam step restricted to n [lo, hi) float ranges of the flat buffers (lo_hi = 2n offsets, lo a multiple of 4): a data-parallel rank
 * updates only the shard of the optimiser state it owns after a reduce-scatter of the gradients; parameters are then all-gathered */
int ntf_apply_ranges(ntf_engine* e, const int64_t* lo_hi, int32_t n);
/* whole phase without host round trips: `for batch in loader` of src/mdl/fnn.py:118 run natively.
 * order = n row ids in the order the loader yields them; mean of batch losses is returned (fnn.py:153). */
int ntf_train_epoch(ntf_engine* e, const int64_t* order, int64_t n, int32_t B, float* mean_loss);
int ntf_eval_epoch(ntf_engine* e, const int64_t* order, int64_t n, int32_t B, float* mean_loss);
int ntf_epoch_loss(ntf_engine* e, double* sum, int64_t* steps); /* reads and clears the accumulator */
/* the same without any per-step host copy: stage the epoch's row order once, then step by offset.  The local
 * shard is order[offset, offset+B); the global minibatch it belongs to is order[global_offset, +global_B)
 * (they coincide on one GPU).  apply != 0 runs Adam right after backward (single GPU). */
int ntf_stage_order(ntf_engine* e, const int64_t* order, int64_t n);
int ntf_step_staged(ntf_engine* e, int64_t offset, int32_t B, int64_t global_offset, int32_t global_B, int32_t train, int32_t apply, float* loss_out);

/* data-parallel pipelining.  ntf_step_staged_deferred = ntf_step_staged(train=1, apply=0) except that the output layer's
 * weight-gradient kernel is left pending; ntf_dw_chunk(k), k = 0..n-1 in order, then launches it for the k-th range of experts, so that
 * the host can all-reduce the gradients of chunk k while chunk k+1 is being computed.  ntf_dw_chunks: n for this model (0 when the
 * fused output-layer path does not apply: nothing is deferred then).  ntf_dw_chunk_range: where chunk k's gradients sit in the flat
 * gradient buffer (offsets in floats; off_rho = -1 for Fnn) - a pure function of the model shape, identical on every rank.
 * After the last chunk the whole gradient buffer is complete - on the engine's stream: the hidden layers' backward of a deferred step runs on a side stream beside the
 * chunks (round 5) and is joined behind the LAST ntf_dw_chunk, so work queued on the engine's stream after that call (the caller's collectives) sees every gradient;
 * ntf_get_grad / ntf_apply / ntf_apply_ranges / ntf_epoch_loss / ntf_synchronize / the next step join it themselves if the chunks were abandoned.
 * ntf_param_segment: where a parameter sits in the flat buffers. */
int ntf_step_staged_deferred(ntf_engine* e, int64_t offset, int32_t B, int64_t global_offset, int32_t global_B, float* loss_out);
int ntf_dw_chunks(ntf_engine* e, int32_t* n_chunks);
int ntf_dw_chunk_range(ntf_engine* e, int32_t k, int64_t* off_weight, int64_t* off_rho, int64_t* count);
int ntf_dw_chunk(ntf_engine* e, int32_t k);
/* ... and of the step's HEAD (round 5).  Under data parallelism the output layer's parameters arrive by all-gather in the ranges of ntf_dw_chunk_range; a rank need not
 * wait for all of them before its next step: ntf_step_staged_deferred_cb runs the Flipout operand producer and the forward kernel RANGE BY RANGE (ntf_fwd_ranges: up to
 * four ranges of whole dW chunks, k0_k1[2 j], k0_k1[2 j + 1] = the dW chunks [k0, k1) of range j; 0 ranges: the model / arithmetic does not allow it - use
 * ntf_step_staged_deferred) and calls `before_range(j, user)` on the calling thread in front of range j: the caller makes the engine's stream wait for the all-gathers of
 * that range's chunks there (e.g. torch's Work.wait()), so that RCCL moves range j + 1 while the forward kernel works on range j.  A non-zero return aborts the step
 * (NTF_ESTATE).  Everything else is ntf_step_staged_deferred.  No counterpart in the reference (src/__config__.yaml:10 "TODO: multiple gpus"). */
int ntf_fwd_ranges(ntf_engine* e, int32_t B, int32_t* n_ranges, int32_t* k0_k1);
int ntf_step_staged_deferred_cb(ntf_engine* e, int64_t offset, int32_t B, int64_t global_offset, int32_t global_B, float* loss_out,
                                int (*before_range)(int32_t range, void* user), void* user);
int ntf_param_segment(ntf_engine* e, int layer, int kind, int64_t* off, int64_t* count);

/* expert-sharded output layer (ntf_config.expert_lo ..): one train step on the whole minibatch order[offset, offset + B) in three phases.
 * phase 1: forward, loss, sparse fix-up - leaves this engine's PARTIAL d(hidden) in the buffer ntf_dh_buffer names ([B, h[-1]] floats, row-major);
 * (the host starts the sum of that buffer over the engines: one all-reduce of B * h[-1] floats, the only exchange of the step;)
 * phase 2: the output layer's backward on this engine's experts, its Adam included - independent of the exchange, which it hides;
 * (the host makes the engine's stream wait for the all-reduce;)
 * phase 3: backward through the replicated hidden layers from the summed d(hidden), Adam on them (identical on every engine).
 * The loss accumulated for ntf_epoch_loss is this engine's share: sum over engines = the single-engine loss.  Evaluation steps need no
 * exchange: ntf_step_staged(train = 0) as usual.  src/mdl/fnn.py:122-140 (the step this splits) */
int ntf_step_staged_ep(ntf_engine* e, int64_t offset, int32_t B, int32_t phase);
int ntf_dh_buffer(ntf_engine* e, void** dev_ptr, int64_t* n_floats);

/* ---- inference:  Fnn.test batch body                              src/mdl/fnn.py:200-211
 * probs_host [B, M] = sigmoid(forward) (Bnn: mean over nmc stochastic forwards);
 * pred_unc/model_unc [B] = predictive entropy / mutual information (may be NULL). */
int ntf_forward(ntf_engine* e, const int64_t* rows, int32_t B, int32_t nmc, const ntf_inject* inj_per_mc,
                float* probs_host, float* pred_unc, float* model_unc);
/* pre-sigmoid post-leaky_relu logits of ONE forward (the quantity the 1e-4 parity bar is stated on); computed by the same kernel as
 * ntf_forward (for H = 128: the split-product forward kernel), not by a separate reference path */
int ntf_logits(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, float* logits_host);
/* top-K per row of the probabilities, without moving [B, M] to the host   src/pkgmgr.py:125-134 */
int ntf_forward_topk(ntf_engine* e, const int64_t* rows, int32_t B, int32_t nmc, int32_t K,
                     float* values_host, int32_t* indices_host, float* pred_unc, float* model_unc);

/* ---- the team2vec gather on its own:  Gnn.get_dense_vecs           src/mdl/emb/gnn.py:484-486
 * out_host [n, d] (may be NULL to keep the result on the device only); rows NULL = 0..n-1. */
int ntf_gather_meanpool(ntf_engine* e, const int64_t* rows, int64_t n, float* out_host);

/* ---- raw device views for RCCL (torch.distributed) and measurement */
int ntf_grad_buffer(ntf_engine* e, void** dev_ptr, int64_t* n_floats);
int ntf_param_buffer(ntf_engine* e, void** dev_ptr, int64_t* n_floats);
/* A caller that WRITES parameters through the view above (an all-gather, a broadcast: torch.optim's `p.data.copy_`, src/mdl/fnn.py:104) says so before the next
 * step: operands a fused train step prepared for its successor from the old values (eps, sigma eps, split planes, KL) are dropped and made again.
 * ntf_param_buffer itself also drops what is pending at the time of the call; ntf_set_param / ntf_apply_ranges / ntf_set_seed do it on their own. */
int ntf_params_touched(ntf_engine* e);
/* Adam's first / second moments (torch.optim.Adam's exp_avg / exp_avg_sq, src/mdl/fnn.py:104), same flat layout as the parameters: an expert-sharded
 * run re-broadcasts the replicated hidden layers' parameters AND moments once per epoch (opentf_amd/ep.py) */
int ntf_moment_buffers(ntf_engine* e, void** dev_m1, void** dev_v2, int64_t* n_floats);
int ntf_synchronize(ntf_engine* e);
/* HIP-event timing of the kernels launched by the engine since the last reset, by kernel family:
 * names[i] (static strings), ms[i] total, calls[i].  Returns the number of families (<= cap).
 * enable: 0 off, 1 every family (two event records around each of ~15 scopes per step), 2 only the output layer's two MFMA kernels
 * (what a roofline needs; keeps the timed region of a benchmark free of the other families' event records), 3 / 4 only its forward / only its dW + Adam kernel
 * (an event pair costs the step ~8 us: a benchmark that alternates 3 and 4 between its timed regions measures both kernels live at half that price) */
int ntf_kernel_times(ntf_engine* e, int enable, const char** names, double* ms, int64_t* calls, int cap);

/* ---- stateless kernels on caller-owned device memory (used by tests and micro-benchmarks) */
int ntf_k_gemm_f32(void* stream, int m, int n, int k, const float* A, int64_t sam, int64_t sak,
                   const float* B, int64_t sbk, int64_t sbn, float* C, int64_t ldc);

/* ---- ranking metrics of the eval stage on the device (next row after the hot path)            src/evl/metric.py:5-35,44-73
 * topk_idx [n, K]: expert ids ranked by decreasing score (ntf_forward_topk order).  The truth / required-skill row of instance i is
 * rows[i] (NULL: i) of the CSR.  out_metrics [n, 5*n_cut] = P, recall, ndcg_cut, map_cut, success, each over the cutoffs (trec_eval
 * definitions, binary relevance);  out_cov [n, n_cut] = |skills of the top-k experts ∩ required| / |required|. */
int ntf_rank_metrics(int device, const int32_t* topk_idx, int64_t n, int32_t K, const int64_t* truth_indptr, const int32_t* truth_indices,
                     int64_t n_truth_rows, const int64_t* rows, const int32_t* cutoffs, int32_t n_cut, float* out_metrics);
int ntf_skill_coverage(int device, const int32_t* topk_idx, int64_t n, int32_t K, const int64_t* skill_indptr, const int32_t* skill_indices,
                       int64_t n_skill_rows, const int64_t* rows, const int64_t* cov_indptr, const int32_t* cov_indices, int64_t n_experts,
                       const int32_t* cutoffs, int32_t n_cut, float* out_cov);

/* ---- member-skill co-occurrence on the device                                                 src/cmn/team.py:302-337
 * `Team.gen_skill_coverage`: C = member^T . skill over the teams NOT listed in skip_rows (the reference empties the test teams' rows,
 * team.py:327-331), as scipy computes it on the two uint8 matrices: counts wrap mod 256, entries whose wrapped count is 0 are not stored.
 * CSR inputs are host pointers (column ids unique inside a row); the result [n_members, n_skills] stays on the device behind an opaque
 * handle: *nnz tells the caller how much to allocate, ntf_csr_result_fetch copies indptr [n_members+1] / indices / data (column ids
 * ascending inside a row = scipy's result after sort_indices()) and reports the device time of the build. */
typedef struct ntf_csr_result ntf_csr_result;
int ntf_skill_cooccurrence(int device, int64_t n_teams, int32_t n_members, int32_t n_skills, const int64_t* m_indptr, const int32_t* m_indices,
                           const int64_t* s_indptr, const int32_t* s_indices, const int64_t* skip_rows, int64_t n_skip,
                           ntf_csr_result** out, int64_t* nnz);
int ntf_csr_result_fetch(ntf_csr_result* r, int64_t* indptr, int32_t* indices, uint8_t* data, double* device_ms);
void ntf_csr_result_free(ntf_csr_result* r);

/* ---- node2vec skill-embedding producer on the device                    src/mdl/emb/gnn.py:153-168 (model), 401-453 (_train_rw)
 * torch_geometric.nn.Node2Vec with p = q = 1 as the reference configures it (src/mdl/emb/__config__.yaml:58-70): uniform random walks over
 * a homogeneous CSR graph (rowptr [num_nodes+1] int64, col int32), windows of `context` nodes, negatives = start node + uniformly random
 * nodes, loss = -mean log(sigmoid(<start, rest>) + 1e-15) - mean log(1 - sigmoid(.) + 1e-15), dense Adam on embedding.weight [num_nodes, d]
 * (1 <= d <= 256: device rows are padded to a multiple of 64 floats, the pad stays zero; init_weight = the nn.Embedding initial draw, supplied by the host so that a seed reproduces torch's).
 * ntf_n2v_train_batch: one loader batch (gnn.py:416-419).  inj_pos / inj_neg non-NULL: the window rows [n, context] are given instead of
 * generated (parity tests); apply = 0 leaves the gradient in place (ntf_n2v_get(what = 1)) and skips Adam.  walk_length counts NODES per
 * walk (= cfg.wl, as Node2Vec's constructor takes it).  ntf_n2v_edge_bce: v_loss of gnn.py:420-431 before its second division. */
typedef struct ntf_n2v ntf_n2v;
int  ntf_n2v_create(int device, int64_t num_nodes, int32_t d, const int64_t* rowptr, const int32_t* col, const float* init_weight, uint64_t seed, ntf_n2v** out);
void ntf_n2v_destroy(ntf_n2v* h);
const char* ntf_n2v_last_error(const ntf_n2v* h);
int  ntf_n2v_walks(ntf_n2v* h, const int64_t* start, int64_t n, int32_t walk_length, uint64_t step, int64_t* out_host /* [n, walk_length] */);
int  ntf_n2v_train_batch(ntf_n2v* h, const int64_t* batch, int32_t B, int32_t walk_length, int32_t context, int32_t walks_per_node, int32_t num_neg, float lr,
                         const int64_t* inj_pos, int64_t n_pos, const int64_t* inj_neg, int64_t n_neg, int32_t apply, float* loss_out);
int  ntf_n2v_get(ntf_n2v* h, int what /* 0 embedding.weight, 1 gradient */, float* host /* [num_nodes, d] */);
int  ntf_n2v_edge_bce(ntf_n2v* h, const int64_t* src, const int64_t* dst, int64_t n, float* mean_bce);

/* ---- doc2vec team-vector producer on the device                          src/mdl/emb/d2v.py:69-84 (gensim.models.Doc2Vec: build_vocab + train)
 * gensim 4.3.3's PV-DM (dm = 1, mean of doc vector + shrunk-window word vectors) and PV-DBOW with dbow_words = 1 (dm = 0), negative sampling from the
 * count^0.75 table, frequent-word subsampling, 1000-bin sigmoid table, |f| >= 6 skipped - as the reference's call leaves gensim's defaults (oracle/d2v_oracle.py
 * restates the algorithm and lists what is pinned).  The host prepares what gensim's build_vocab prepares: documents as CSR over VOCABULARY indices
 * (doc_ptr [n_docs+1] int64, words int32), sample_int [n_vocab] (keep a word iff sample_int >= a uniform uint32), cum_table [n_vocab] (uint32, last = 2^31 - 1),
 * and the initial vectors (numpy default_rng(seed) / (seed + 7919), so that a seed gives gensim's initial table); syn1neg starts at zero.
 * ntf_d2v_train_epoch = one pass over the documents (gensim train(epochs=1)), taken in `order` (nullable: 0..n_docs-1): the document of rank r trains at
 * alpha_start - (alpha_start - alpha_end) * progress[r] (nullable: r / n_docs) - gensim fixes alpha per job of <= 10 000 words, the host passes the fraction of
 * the pass at which the document's job was cut.  Random draws are Philox words keyed by (seed, epoch) and counted by (document, position, unit, slot).  serial != 0: one wave walks
 * all documents in order - the oracle's sequential pass (parity tests); otherwise one wave per document, Hogwild through f32 atomic adds as gensim's worker threads
 * are through plain stores.  mean_loss (nullable): mean -log sigmoid(+-f) over the pairs trained; device_ms (nullable): device time of the pass.
 * d: any vector size 1..256 (data.embedding.d is free in the reference, its CI trains d = 9): on the device a row is padded to a multiple of 64 floats, the pad
 * stays zero; the host side sees [rows, d] arrays only.
 * ntf_d2v_get / ntf_d2v_set: what = 0 doc vectors [n_docs, d] (= Doc2Vec.dv.vectors, row i = team i), 1 word vectors [n_vocab, d], 2 syn1neg [n_vocab, d]. */
typedef struct ntf_d2v ntf_d2v;
int  ntf_d2v_create(int device, int64_t n_docs, int64_t n_vocab, int32_t d, const int64_t* doc_ptr, const int32_t* words, const uint32_t* sample_int,
                    const uint32_t* cum_table, const float* init_wv, const float* init_dv, uint64_t seed, ntf_d2v** out);
void ntf_d2v_destroy(ntf_d2v* h);
const char* ntf_d2v_last_error(const ntf_d2v* h);
int  ntf_d2v_train_epoch(ntf_d2v* h, int32_t dm, int32_t window, int32_t negative, double alpha_start, double alpha_end, uint64_t epoch, int32_t serial,
                         const int64_t* order, const double* progress, double* mean_loss, double* device_ms);
int  ntf_d2v_get(ntf_d2v* h, int what, float* host);
int  ntf_d2v_set(ntf_d2v* h, int what, const float* host);

/* device generators behind Flipout's eps / signs (dev_out = device pointers), for statistical tests */
int ntf_k_fill_normal(void* stream, uint64_t seed, uint64_t step, int layer, int64_t n, float* dev_out);
int ntf_k_fill_sign(void* stream, uint64_t seed, uint64_t step, int layer, int rows, int cols, float* dev_out);

#ifdef __cplusplus
}
#endif
#endif /* OPENTF_AMD_H */
