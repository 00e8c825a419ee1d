// C-ABI engine (include/opentf_amd.h): owns the HBM-resident state of one Fnn/Bnn model on one MI355X
// and runs the minibatch hot loop of src/mdl/fnn.py:118-151 (reference) natively.
#include "../../include/opentf_amd.h"
#include "ntf_kernels.h"
#include "ntf_fused.h"
#include "ntf_head.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace ntf;

static thread_local std::string g_create_error;

struct LayerInfo {
    int in = 0, out = 0;
    int64_t off[4] = {0, 0, 0, 0};  // offsets (floats) of WEIGHT, BIAS, RHO_WEIGHT, RHO_BIAS in the flat buffers
    int64_t nw() const { return (int64_t)in * out; }
};

enum Fam { F_GATHER = 0, F_GEMM_HIDDEN, F_FLIPOUT_OPERAND, F_OUT_FWD, F_LOSS, F_OUT_BWD_DW, F_OUT_BWD_DA, F_BIAS_GRAD,
           F_FLIPOUT_FINAL, F_KL, F_ADAM, F_SAMPLER, F_INFER, F_OUT_FUSED_FWD, F_OUT_FUSED_DW, F_OUT_FUSED_AUX, F_MULTIHOT, F_COUNT };
static const char* kFamNames[F_COUNT] = {"gather", "gemm_hidden", "flipout_operand", "out_fwd_gemm", "loss", "out_bwd_dw_gemm",
                                         "out_bwd_da_gemm", "bias_grad", "flipout_grad_finalize", "kl", "adam", "sampler", "infer",
                                         "out_fused_fwd_loss_dh", "out_fused_dw_adam", "out_fused_prep_special", "multihot_layer0"};

struct TimeRec { int fam; hipEvent_t a, b; };
constexpr int64_t kGemmSlabFloats = 8 << 20;  // 32 MiB

struct StepCtx {
    const int64_t* rows_dev = nullptr;
    int B = 0; int global_B = 0;
    const ntf_inject* inj = nullptr;
    uint64_t step = 0;
    bool train = false;
    uint32_t row0 = 0;       // position of this shard's first row inside its global minibatch: device generators are keyed by the
                             // GLOBAL row position, so a sharded step draws exactly what the single-process step would draw
    bool defer_dw = false;   // leave the output layer's dW kernel to ntf_dw_chunk (data-parallel overlap with the all-reduce)
    bool fuse_adam = false;  // train step with immediate apply on one GPU: Adam of the output layer goes into the dW epilogue
    int (*chunk_cb)(int32_t, void*) = nullptr; void* chunk_user = nullptr;   // data-parallel pipelining: called in front of every forward range (ntf_step_staged_deferred_cb)
    int ub = 0;              // nsd = unigram_b: which of the two staged per-batch alias tables is THIS batch's
    bool chain = false;      // an evaluation step inside ntf_eval_epoch's loop (ntf_engine.eval_chain)
    int part = 0;            // expert-sharded step: 1 = forward + loss (leaves the partial d(hidden)), 2 = the output layer's backward, 3 = the hidden layers' backward (0 = whole step)
};

struct ntf_engine {
    ntf_config cfg{};
    hipStream_t st = nullptr;
    bool own_stream = false;
    std::string err;
    int L = 0;
    std::vector<LayerInfo> layers;
    int64_t n_params = 0;
    float *P = nullptr, *G = nullptr, *M1 = nullptr, *V2 = nullptr;
    int64_t adam_t = 0;
    float lr = 1e-3f;
    // resident data
    int64_t* m_indptr = nullptr; int32_t* m_indices = nullptr; int64_t m_rows = 0;
    std::vector<int64_t> h_m_indptr; std::vector<int32_t> h_m_indices;
    int64_t* s_indptr = nullptr; int32_t* s_indices = nullptr; int64_t s_rows = 0;
    float* table = nullptr; int64_t n_skills = 0; int table_d = 0; int32_t s_max_col = -1;
    float* Xall = nullptr; int64_t x_rows = 0;
    float* al_prob = nullptr; int32_t* al_alias = nullptr; double* al_weight = nullptr; double al_total = 0; int64_t al_n = 0;
    // per-step buffers
    int64_t* d_rows = nullptr; int64_t* d_order = nullptr; int64_t order_cap = 0; std::vector<int64_t> h_order;
    // sampled negatives [B, ns] of a step live in one of TWO buffers, by step parity (as the fused workspace does): a train step's head prefetch runs the NEXT batch's
    // sampler beside this step's dW kernel, and ntf_get_negatives / the sparse fix-up of the step in flight must keep reading their own draws
    int64_t* d_neg_set[2] = {nullptr, nullptr}; uint64_t neg_step = 0;
    std::vector<float*> act;          // act[0] = X [B, D], act[l] = leaky_relu output of layer l-1 (hidden)
    float *Zout = nullptr, *dZout = nullptr, *Pbuf = nullptr;
    float *Zh = nullptr;              // [B, max hidden] pre-activation scratch for hidden flipout layers
    float* dAct[2] = {nullptr, nullptr};
    std::vector<float*> Wp, bp;       // flipout perturbation operands per layer
    float *partial = nullptr, *row_fix = nullptr, *d_loss = nullptr, *ent_mc = nullptr, *ent_mean = nullptr;
    float *dh_slab = nullptr;         // fused path: partial d(hidden) slabs
    char* fws = nullptr;              // fused path: sign-bit images, h*s_in, loss partials - of the CURRENT step: one of the two sets below, by step parity
    char* fws_set[2] = {nullptr, nullptr};
    // Head prefetch (round 4): what a fused train step needs before its forward kernel and that depends on the batch's rows and the hidden layer only - the sampled
    // negatives, the transposed s_out words, gather -> hidden layer -> h images (k_head) - is issued for the NEXT batch of the staged order on the side stream, behind
    // this step's hidden-layer backward (+ its Adam), i.e. beside the dW kernel, into the other workspace set.  hp = what was issued; a step takes it when it is that batch.
    int head_prefetch = 1;            // NTF_HEAD_PREFETCH=0: everything at the head of its own step (A/B runs)
    struct { bool valid = false; uint64_t step = 0; const int64_t* rows = nullptr; int B = 0; int ub = -1; } hp;      // ub: the unigram_b table slot staged for that batch (-1: none)
    const int64_t* hp_next_rows = nullptr; int hp_next_B = 0;     // the batch that follows in the staged order (ntf_step_staged), 0: unknown
    const int64_t* hp_next_host = nullptr;                        // ... and its row ids on the host (unigram_b: the per-batch table is built from them)
    bool hidden_adam_done = false;    // this step's Adam of the hidden layers already ran on the side stream (in front of the prefetched head)
    int64_t hp_used = 0;
    float* gemm_slab = nullptr;       // split-K partial sums of the generic GEMM
    double* d_kl = nullptr; double* d_acc = nullptr; int64_t* d_acc_steps = nullptr;
    int fwd_kernel = -1;
#ifdef NTF_DIAG
    // NTF_COSCHED=ncg (a -DNTF_DIAG build; RESULTS ARE GARBAGE): the co-scheduling experiment of DESIGN.md section 4 - the forward kernel on NRB * ncg workgroups (one
    // per CU) while, on the side stream, the dW + Adam kernel of the PREVIOUS step's operands fills the CUs it leaves free; the step's own dW launch is skipped
    int cosched = 0; bool cosched_have = false; FusedDw cosched_dw; hipEvent_t ev_co0 = nullptr, ev_co1 = nullptr;
#endif
    int lean = 1;                     // NTF_LEAN=0: the dW epilogue also writes the f32 copy of the next step's sigma * eps (round 3's 64 B per pair; A/B runs)
    int dw_tail = 0;                  // NTF_DW_TAIL=1|2|3 (-DNTF_DIAG builds only; measured SLOWER): the last partial round of half-tiles as split-K launches (dw_launch_whole)
    int n_cu = 256;
    int dw_ksplit = 0;                // 0: automatic (few expert tiles -> split the dW kernel's K range), else forced (NTF_DW_KSPLIT)
    int32_t* d_range = nullptr;       // fp16x3 range guard (lives behind d_kl[0]): [0] raised for the current step, [1] steps that fell back to the f32 kernels
    int64_t range_fallbacks_host = 0; // inference calls redone on the generic path for the same reason
    float* tk_vals = nullptr; int32_t* tk_idx = nullptr; int64_t tk_cap = 0;
    // injection staging (device)
    std::vector<float*> inj_eps_w, inj_eps_b, inj_s_in, inj_s_out;
    uint64_t seed = 0, step = 0;
    int maxhid = 0;
    // timing
    int timing = 0;                   // 0 off, 1 every kernel family, 2 only the output layer's two MFMA kernels (the roofline's kernels), 3 / 4 only its forward / only its dW kernel
    std::vector<TimeRec> recs;
    std::vector<hipEvent_t> pool;
    double fam_ms[F_COUNT] = {0}; int64_t fam_calls[F_COUNT] = {0};
    int last_global_B = 0; int last_B = 0; float last_dz_packed_scale = 0.f;
    bool adam_in_dw = false;          // this step's output-layer Adam already ran inside / beside the dW kernel
    // the dW + Adam kernel of a train step also produced the NEXT step's output-layer operands (FusedDw.produce): valid for step pre_step as long as nothing else
    // touched the output layer's parameters or the operand buffers (Wp, split planes); d_kl[2] / d_range[4] hold that step's KL and range flag until it starts
    int f32_copy_merged = 1;          // NTF_F32_COPY_MERGED=0: the conditional f32 copy of sigma * eps as a launch of its own in front of every step, as in round 4 (A/B runs)
    uint64_t f32_copy_step = 0;       // step + 1 whose copy job rode in the previous step's Adam launch
    int merge_bias = 1;               // NTF_MERGE_BIAS=0: the next step's output-bias operand as k_head's bias workgroups in a launch of their own behind the bias Adam, as in round 4 (A/B runs)
    int dp_side_bwd = 1;              // NTF_DP_SIDE_BWD=0: a deferred-dW (data-parallel) step runs its hidden layers' backward on the main stream in front of the dW chunks, as in round 4 (A/B runs)
    int dp_ranges = 1;                // NTF_DP_RANGES=0: a data-parallel rank waits for every parameter all-gather before its step, as in round 4 (A/B runs, tests)
    int ep_head_prefetch = -1;        // NTF_EP_HEAD_PREFETCH: an expert shard's phase 3 issues the next batch's head behind its hidden backward, beside its dW kernel (phase 2).  -1 (default): when the
                                      // dW kernel outlasts the backward (B x 8 <= the shard's experts: ranks of 2-4 at config 2; 1.27 against 1.33 ms at a rank of 4, neutral at 8 where both end together);
                                      // 0 never; 2 always; 1 always, with the sampler / sign words on the auxiliary stream from the end of phase 1 (beside the whole dW kernel: slower, 1.34 against 1.30 at 8)
    int head = 1;                     // NTF_HEAD=0: the step's head as its chain of small kernels (A/B runs)
    bool pre_valid = false; uint64_t pre_step = 0; int prefetch = 1;   // NTF_PREFETCH=0: always the stand-alone producer (A/B runs)
    // Multi-hot Flipout first layer (BASELINE config 3: 90 671 x 128 mu / rho pairs) in a step that applies Adam (round 6): launch_flipout_sweep - finalize + Adam + the NEXT
    // step's sigma * eps and KL in one pass, on the side stream beside the dW kernel.  pre0_step: the step whose Wp[0] / KL the last sweep produced (used iff that step also
    // starts on the output layer's prefetched operands: one validity protocol, pre_valid / pre_step); l0_swept: this step's Adam of the layer's weights ran in the sweep
    // (apply_adam leaves them out); g0_clean: the layer's gradient rows are all zero (the sweep clears what it reads: no memset in front of the scatter)
    int eval_kernel = 1;              // NTF_EVAL_KERNEL=0 (A/B runs, tests): evaluation steps on k_out_fwd_b6 as in round 5 instead of k_out_fwd_h3e
    int mh_head = 1;                  // NTF_MH_HEAD=0 (A/B runs): a multi-hot step's head (first-layer operands, gather-sum, h images) inline in its own step as in rounds 1-5; 1 (round 6): as ONE
                                      // unit (head_launch's multi-hot form) that the previous step issues for the next staged batch beside its dW kernel, behind the first layer's sweep
    int fnn_pipe = 1;                 // NTF_FNN_PIPE=0 (A/B runs): the non-Bayesian step as in round 5 - mu planes split at the head of every step (k_split_planes), hidden backward
                                      // on the main stream, no head prefetch.  1 (round 6): the dW + Adam epilogue writes the NEXT step's planes of mu (FusedDw.produce without the
                                      // Flipout half), the hidden backward and the next batch's head run on the side stream beside the dW kernel - the Bnn step's pipeline
    int l0_sweep = 1;                 // NTF_L0_SWEEP=0: k_flipout_grad_finalize + the flat Adam + the stand-alone producer, as in round 5 (A/B runs)
    bool l0_swept = false, g0_clean = false, use_pre0 = false; uint64_t pre0_step = ~0ull; uint8_t* l0_touched = nullptr; int64_t pre0_used = 0;
    int64_t pre_used = 0;             // steps that started on prefetched operands (diagnostics / tests)
    // evaluation steps of one ntf_eval_epoch call (run_epoch) run back to back on unchanged parameters: eval_chain != 0 while that loop runs; the first producer launch of
    // the chain keeps the output layer's KL term and the range verdict on the planes of mu (d_chain: [double KL, int flag]), the later ones reuse them (PerturbChain)
    int eval_chain = 0; bool chain_valid = false; double* d_chain = nullptr;
    bool join_pending = false;        // a deferred-dW step (data parallel) left its hidden layers' backward running on the side stream: joined (join_side) behind the last dW chunk, or by whatever reads its results first
    bool pre_rotated = false;         // ... and whose KL / range-flag scalars the previous step's last kernel already moved into place
    bool fin_pend = false; NormalSpec fin_eps; float fin_klw = 0.f;   // the output layer's bias-gradient finalisation rides in the Adam launch that follows (fused-Adam steps)
    // data-parallel pipelining: the output layer's dW kernel deferred by ntf_step_staged_deferred, launched by ntf_dw_chunk
    int (*cb_fn)(int32_t, void*) = nullptr; void* cb_user = nullptr;   // ntf_step_staged_deferred_cb -> the step's StepCtx
    bool pend_valid = false; FusedDw pend; NormalSpec pend_eps_b; float pend_klw_b = 0.f; int pend_chunks = 0;
    // unigram_b staging (sparse per-batch alias table)
    std::vector<int32_t> ub_entries; void* ub_host[2] = {nullptr, nullptr}; void* ub_dev[2] = {nullptr, nullptr}; hipEvent_t ub_ev[2] = {nullptr, nullptr};
    bool ub_used[2] = {false, false}; size_t ub_cap = 0; int ub_slot = 0; int ub_nsup[2] = {0, 0}; double ub_total[2] = {0, 0};
    uint16_t* pl_mu = nullptr; uint16_t* pl_wp = nullptr;   // bf16 split planes of the output layer's mu / Wp (bf16x6 arithmetic)
    hipStream_t st2 = nullptr;        // side stream: Adam of finished expert chunks runs beside the dW kernel of the next chunk
    hipEvent_t ev_chunk = nullptr, ev_side = nullptr;
    hipStream_t st3 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the hidden layers' backward runs beside the output layer's dW kernel
    hipStream_t st4 = nullptr; hipEvent_t ev_aux = nullptr;   // auxiliary stream: what the step's head needs from the row ids / sign keys only (sampler, s_out words)
    int side_bwd = 1;                 // NTF_SIDE_BWD=0 keeps the whole step on one stream (A/B runs)
    // expert-sharded output layer (ntf_config.expert_lo ..): this engine owns experts [ep_lo, ep_lo + dims[L]) of Mg
    bool ep = false; int ep_lo = 0, Mg = 0, ep_world = 1;
    bool ep_side = false;                   // this step's phase 2 runs on the side stream
    int ep_open = 0; StepCtx ep_ctx;        // the ntf_step_staged_ep phase that ran last (0: none pending)
};

#define HIPCHK(e, call)                                                                                   \
    do {                                                                                                  \
        hipError_t _s = (call);                                                                           \
        if (_s != hipSuccess) {                                                                           \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_s);                                 \
            return NTF_EHIP;                                                                              \
        }                                                                                                 \
    } while (0)
#define FAIL(e, code, msg) do { (e)->err = (msg); return (code); } while (0)
static int join_side(ntf_engine* e);

struct Scope {
    ntf_engine* e; int fam; hipEvent_t a = nullptr, b = nullptr;
    bool on() const {
        const bool fwd = fam == F_OUT_FUSED_FWD || fam == F_OUT_FWD, dw = fam == F_OUT_FUSED_DW || fam == F_OUT_BWD_DW;
        return e->timing == 1 || (e->timing == 2 && (fwd || dw)) || (e->timing == 3 && fwd) || (e->timing == 4 && dw);
    }
    Scope(ntf_engine* e_, int f) : e(e_), fam(f) {
        if (!on()) return;
        auto get = [&]() { hipEvent_t ev; if (!e->pool.empty()) { ev = e->pool.back(); e->pool.pop_back(); } else hipEventCreate(&ev); return ev; };
        a = get(); b = get();
        hipEventRecord(a, e->st);
    }
    ~Scope() {
        if (!on()) return;
        hipEventRecord(b, e->st);
        e->recs.push_back({fam, a, b});
    }
};

static void drain_times(ntf_engine* e) {
    if (e->recs.empty()) return;
    hipStreamSynchronize(e->st);
    for (auto& r : e->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { e->fam_ms[r.fam] += ms; e->fam_calls[r.fam] += 1; }
        e->pool.push_back(r.a); e->pool.push_back(r.b);
    }
    e->recs.clear();
}

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// per-(seed, step, layer, tensor) key of the device generators
static void make_key(const ntf_engine* e, uint64_t step, int layer, int tensor, uint32_t& k0, uint32_t& k1) {
    uint64_t h = splitmix64(e->seed ^ splitmix64(step * 0x100 + (uint64_t)layer * 8 + (uint64_t)tensor));
    k0 = (uint32_t)h; k1 = (uint32_t)(h >> 32);
}
enum { T_EPS_W = 0, T_EPS_B = 1, T_S_IN = 2, T_S_OUT = 3, T_NEG = 4 };

template <typename T> static int dmalloc(ntf_engine* e, T** p, int64_t n) {
    *p = nullptr;
    if (n <= 0) return NTF_OK;
    hipError_t s = hipMalloc((void**)p, (size_t)n * sizeof(T));
    if (s != hipSuccess) { e->err = std::string("hipMalloc: ") + hipGetErrorString(s); return NTF_ENOMEM; }
    return NTF_OK;
}
#define DM(e, p, n) do { int _r = dmalloc(e, p, n); if (_r) return _r; } while (0)
template <typename T> static void dfree(T*& p) { if (p) { hipFree(p); p = nullptr; } }

extern "C" int ntf_abi_version(void) { return NTF_ABI_VERSION; }
extern "C" const char* ntf_last_error(const ntf_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

static bool fused_ok(const ntf_engine* e) {
    if (!e->cfg.fused) return false;
    return fused_supported(e->layers[e->L - 1].in);
}

extern "C" int ntf_engine_create(const ntf_config* cfg, ntf_engine** out) {
    if (!out) { g_create_error = "out is NULL"; return NTF_EINVAL; }
    *out = nullptr;
    if (!cfg || cfg->abi_version != NTF_ABI_VERSION) { g_create_error = "bad config / abi_version"; return NTF_EINVAL; }
    if (cfg->n_layers < 1 || cfg->n_layers > NTF_MAX_LAYERS || cfg->max_batch < 1 || cfg->ns < 0) { g_create_error = "bad n_layers/max_batch/ns"; return NTF_EINVAL; }
    for (int i = 0; i <= cfg->n_layers; ++i) if (cfg->dims[i] < 1) { g_create_error = "bad dims"; return NTF_EINVAL; }
    if (cfg->mfma == NTF_MFMA_BF16X6_RETIRED) { g_create_error = "mfma = 2 (bf16x6: three bf16 values per operand, six products) was retired in round 6: the default fp16x3 has its accuracy at half the matrix work; use NTF_MFMA_DEFAULT or NTF_MFMA_F32"; return NTF_EINVAL; }
    if (cfg->mfma < 0 || cfg->mfma > NTF_MFMA_FP16X3) { g_create_error = "bad mfma"; return NTF_EINVAL; }
    if (cfg->input_mode == NTF_INPUT_MULTIHOT && cfg->n_layers < 2) { g_create_error = "multi-hot input needs a hidden layer (src/mdl/fnn.py:17-19)"; return NTF_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "no HIP device: the engine has no CPU fallback"; return NTF_EHIP; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return NTF_EINVAL; }
    if (hipSetDevice(cfg->device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return NTF_EHIP; }
    ntf_engine* e = new ntf_engine();
    e->cfg = *cfg;
    e->L = cfg->n_layers;
    e->lr = cfg->lr;
    e->seed = cfg->seed;
    if (const char* fk = getenv("NTF_FWD_KERNEL")) {
        e->fwd_kernel = atoi(fk);
        if (e->fwd_kernel != 0 && e->fwd_kernel != 5) { g_create_error = "NTF_FWD_KERNEL: 5 (k_out_fwd_h3p, the default) or 0 (k_out_fwd_b6); the forms 1-4 of rounds 2-3 are retired"; delete e; return NTF_EINVAL; }
    }
    if (const char* ks = getenv("NTF_DW_KSPLIT")) e->dw_ksplit = atoi(ks);
    if (const char* ln = getenv("NTF_LEAN")) e->lean = atoi(ln);
#ifdef NTF_DIAG
    if (const char* dt = getenv("NTF_DW_TAIL")) e->dw_tail = atoi(dt);
#endif
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) e->n_cu = ncu; }
    if (const char* hp = getenv("NTF_HEAD_PREFETCH")) e->head_prefetch = atoi(hp);
#ifdef NTF_DIAG
    if (const char* co = getenv("NTF_COSCHED")) e->cosched = atoi(co);
#endif
    if (const char* pf = getenv("NTF_PREFETCH")) e->prefetch = atoi(pf);
    if (const char* hd = getenv("NTF_HEAD")) e->head = atoi(hd);
    if (const char* mb = getenv("NTF_MERGE_BIAS")) e->merge_bias = atoi(mb);
    if (const char* sw = getenv("NTF_L0_SWEEP")) e->l0_sweep = atoi(sw);
    if (const char* fp = getenv("NTF_FNN_PIPE")) e->fnn_pipe = atoi(fp);
    if (const char* mh = getenv("NTF_MH_HEAD")) e->mh_head = atoi(mh);
    if (const char* ek = getenv("NTF_EVAL_KERNEL")) e->eval_kernel = atoi(ek);
    if (const char* fc = getenv("NTF_F32_COPY_MERGED")) e->f32_copy_merged = atoi(fc);
    if (const char* eh = getenv("NTF_EP_HEAD_PREFETCH")) e->ep_head_prefetch = atoi(eh);
    if (const char* dr = getenv("NTF_DP_RANGES")) e->dp_ranges = atoi(dr);
    if (const char* ds = getenv("NTF_DP_SIDE_BWD")) e->dp_side_bwd = atoi(ds);
    if (const char* sb = getenv("NTF_SIDE_BWD")) e->side_bwd = atoi(sb);   // A/B runs and tests: 1 = never split the dW kernel's K range, n = force n, unset = by the tile count
    if (cfg->stream) e->st = (hipStream_t)cfg->stream;
    else { if (hipStreamCreate(&e->st) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete e; return NTF_EHIP; } e->own_stream = true; }
    int64_t off = 0;
    auto seg = [&](int64_t n) { int64_t o = off; off += (n + 63) / 64 * 64; return o; };
    e->layers.resize(e->L);
    for (int l = 0; l < e->L; ++l) {
        LayerInfo& li = e->layers[l];
        li.in = cfg->dims[l]; li.out = cfg->dims[l + 1];
        li.off[NTF_P_WEIGHT] = seg(li.nw());
        li.off[NTF_P_BIAS] = seg(li.out);
        if (cfg->bayesian) { li.off[NTF_P_RHO_WEIGHT] = seg(li.nw()); li.off[NTF_P_RHO_BIAS] = seg(li.out); }
        if (l < e->L - 1) e->maxhid = std::max(e->maxhid, li.out);
    }
    e->n_params = off;
    if (cfg->fused && !fused_supported(e->layers[e->L - 1].in) && (int64_t)cfg->dims[e->L] >= 4096 && !getenv("NTF_QUIET")) {
        // (a hidden width outside {32, 64, 128} fell to the generic chain - k_gemm + dense [B, M] loss - without a word: 59 k teams/s against ~740 k at config 2's size)
        static bool warned = false;
        if (!warned) fprintf(stderr, "opentf_amd: h[-1] = %d has no fused output-layer kernels (32, 64 and 128 have; 128 also the split-product ones): this model runs on the "
                                     "generic GEMM + dense-loss chain, an order of magnitude slower at %d experts (INTEGRATION.md, limits)\n", e->layers[e->L - 1].in, cfg->dims[e->L]);
        warned = true;
    }
    e->Mg = cfg->experts_global > 0 ? cfg->experts_global : cfg->dims[e->L];
    e->ep_lo = cfg->expert_lo; e->ep_world = std::max(1, cfg->ep_world);
    e->ep = cfg->experts_global > 0 || e->ep_lo != 0 || e->ep_world > 1;   // a shard may also be the whole layer (ep_world = 1)
    if (e->ep) {
        const char* bad = nullptr;
        if (!fused_ok(e)) bad = "an expert-sharded output layer needs the fused output-layer path (fused = 1, h[-1] in {32, 64, 128})";
        else if (e->ep_lo < 0 || (e->ep_lo & 255)) bad = "expert_lo must be a non-negative multiple of 256";
        else if ((int64_t)e->ep_lo + cfg->dims[e->L] > e->Mg) bad = "expert_lo + dims[n_layers] exceeds experts_global";
        if (bad) { g_create_error = bad; if (e->own_stream) hipStreamDestroy(e->st); delete e; return NTF_EINVAL; }
    }
    const int B = cfg->max_batch, M = cfg->dims[e->L];
    int rc = NTF_OK;
    auto A = [&](int r) { if (rc == NTF_OK) rc = r; };
    A(dmalloc(e, &e->P, off)); A(dmalloc(e, &e->G, off)); A(dmalloc(e, &e->M1, off)); A(dmalloc(e, &e->V2, off));
    A(dmalloc(e, &e->d_rows, B)); for (int k = 0; k < 2; ++k) A(dmalloc(e, &e->d_neg_set[k], (int64_t)B * std::max(1, cfg->ns)));
    e->act.assign(e->L, nullptr);
    for (int l = (cfg->input_mode == NTF_INPUT_MULTIHOT ? 1 : 0); l < e->L; ++l) A(dmalloc(e, &e->act[l], (int64_t)B * cfg->dims[l]));  // multi-hot X is never dense
    A(dmalloc(e, &e->Zout, (int64_t)B * M)); A(dmalloc(e, &e->dZout, (int64_t)((M + 255) / 256 * 256) * fused_ldb(B)));
    if (e->maxhid) { A(dmalloc(e, &e->Zh, (int64_t)B * e->maxhid)); A(dmalloc(e, &e->dAct[0], (int64_t)B * e->maxhid)); A(dmalloc(e, &e->dAct[1], (int64_t)B * e->maxhid)); }
    e->Wp.assign(e->L, nullptr); e->bp.assign(e->L, nullptr);
    if (cfg->bayesian) for (int l = 0; l < e->L; ++l) { A(dmalloc(e, &e->Wp[l], e->layers[l].nw())); A(dmalloc(e, &e->bp[l], e->layers[l].out)); }
    A(dmalloc(e, &e->partial, (int64_t)B * std::max(loss_dense_nchunk(M), fused_loss_slots(M)))); A(dmalloc(e, &e->row_fix, B));
    if (cfg->bayesian && cfg->input_mode == NTF_INPUT_MULTIHOT) A(dmalloc(e, &e->l0_touched, (int64_t)(cfg->dims[0] + 255) / 256 * 256));
    A(dmalloc(e, &e->d_loss, 4)); A(dmalloc(e, &e->d_kl, 4)); A(dmalloc(e, &e->d_chain, 2)); A(dmalloc(e, &e->d_acc, 2)); A(dmalloc(e, &e->d_acc_steps, 2));
    A(dmalloc(e, &e->ent_mc, B)); A(dmalloc(e, &e->ent_mean, B)); A(dmalloc(e, &e->gemm_slab, kGemmSlabFloats));
    if (rc == NTF_OK && fused_ok(e)) {
        A(dmalloc(e, &e->dh_slab, fused_dh_slab_floats(B, e->layers[e->L - 1].in, M)));
        for (int k = 0; k < 2; ++k) A(dmalloc(e, &e->fws_set[k], (int64_t)fused_workspace_bytes(B, e->layers[e->L - 1].in, M)));
        e->fws = e->fws_set[0];
        if (cfg->mfma != NTF_MFMA_F32 && e->layers[e->L - 1].in == 128) {
            A(dmalloc(e, &e->pl_mu, fused_planes_elems(M, 128)));
            if (cfg->bayesian) A(dmalloc(e, &e->pl_wp, fused_planes_elems(M, 128)));
            if (rc == NTF_OK) { hipMemsetAsync(e->pl_mu, 0, (size_t)fused_planes_elems(M, 128) * 2, e->st); if (e->pl_wp) hipMemsetAsync(e->pl_wp, 0, (size_t)fused_planes_elems(M, 128) * 2, e->st); }
        }
    }
    e->inj_eps_w.assign(e->L, nullptr); e->inj_eps_b.assign(e->L, nullptr); e->inj_s_in.assign(e->L, nullptr); e->inj_s_out.assign(e->L, nullptr);
    if (rc != NTF_OK) { g_create_error = e->err; ntf_engine_destroy(e); return rc; }
    hipMemsetAsync(e->P, 0, off * 4, e->st); hipMemsetAsync(e->G, 0, off * 4, e->st);
    hipMemsetAsync(e->M1, 0, off * 4, e->st); hipMemsetAsync(e->V2, 0, off * 4, e->st);
    hipMemsetAsync(e->d_acc, 0, 16, e->st); hipMemsetAsync(e->d_acc_steps, 0, 16, e->st);
    hipMemsetAsync(e->d_kl, 0, 32, e->st);
    e->d_range = reinterpret_cast<int32_t*>(e->d_kl + 1);
    if (hipStreamSynchronize(e->st) != hipSuccess) { g_create_error = "device initialisation failed"; ntf_engine_destroy(e); return NTF_EHIP; }
    *out = e;
    return NTF_OK;
}

extern "C" void ntf_engine_destroy(ntf_engine* e) {
    if (!e) return;
    hipSetDevice(e->cfg.device);
    if (e->st) hipStreamSynchronize(e->st);
    // the side streams are drained and destroyed BEFORE any buffer their kernels may still touch is freed
    if (e->st4) { hipStreamSynchronize(e->st4); hipStreamDestroy(e->st4); hipEventDestroy(e->ev_aux); }
    if (e->st3) { hipStreamSynchronize(e->st3); hipStreamDestroy(e->st3); hipEventDestroy(e->ev_fork); hipEventDestroy(e->ev_join); }
    if (e->st2) { hipStreamSynchronize(e->st2); hipStreamDestroy(e->st2); hipEventDestroy(e->ev_chunk); hipEventDestroy(e->ev_side); }
    dfree(e->P); dfree(e->G); dfree(e->M1); dfree(e->V2);
    dfree(e->m_indptr); dfree(e->m_indices); dfree(e->s_indptr); dfree(e->s_indices); dfree(e->table); dfree(e->Xall);
    dfree(e->al_prob); dfree(e->al_alias); dfree(e->al_weight);
    dfree(e->d_rows); dfree(e->d_order); dfree(e->d_neg_set[0]); dfree(e->d_neg_set[1]);
    for (auto& p : e->act) dfree(p);
    dfree(e->Zout); dfree(e->dZout); dfree(e->Pbuf); dfree(e->Zh); dfree(e->dAct[0]); dfree(e->dAct[1]);
    for (auto& p : e->Wp) dfree(p);
    for (auto& p : e->bp) dfree(p);
    dfree(e->l0_touched);
    dfree(e->partial); dfree(e->row_fix); dfree(e->d_loss); dfree(e->d_kl); dfree(e->d_chain); dfree(e->d_acc); dfree(e->d_acc_steps);
    dfree(e->ent_mc); dfree(e->ent_mean); dfree(e->dh_slab); dfree(e->fws_set[0]); dfree(e->fws_set[1]); dfree(e->pl_mu); dfree(e->pl_wp); dfree(e->gemm_slab); dfree(e->tk_vals); dfree(e->tk_idx);
    for (auto* v : {&e->inj_eps_w, &e->inj_eps_b, &e->inj_s_in, &e->inj_s_out}) for (auto& p : *v) dfree(p);
    for (auto& r : e->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto ev : e->pool) hipEventDestroy(ev);
    for (int k = 0; k < 2; ++k) { if (e->ub_host[k]) hipHostFree(e->ub_host[k]); if (e->ub_dev[k]) hipFree(e->ub_dev[k]); if (e->ub_ev[k]) hipEventDestroy(e->ub_ev[k]); }
    if (e->own_stream && e->st) hipStreamDestroy(e->st);
    delete e;
}

// ------------------------------------------------------------------------------------------ data
static int upload_csr(ntf_engine* e, const int64_t* indptr, const int32_t* indices, int64_t n_rows, int64_t** d_ip, int32_t** d_ix, int width) {
    if (!indptr || n_rows < 0) FAIL(e, NTF_EINVAL, "csr: bad arguments");
    const int64_t nnz = indptr[n_rows];
    if (nnz < 0 || (nnz > 0 && !indices)) FAIL(e, NTF_EINVAL, "csr: bad arguments");
    for (int64_t i = 0; i < n_rows; ++i) if (indptr[i + 1] < indptr[i]) FAIL(e, NTF_EINVAL, "csr: indptr not monotone");
    for (int64_t p = 0; p < nnz; ++p) if (indices[p] < 0 || indices[p] >= width) FAIL(e, NTF_EINVAL, "csr: column id out of range");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipStreamSynchronize(e->st));
    dfree(*d_ip); dfree(*d_ix);
    DM(e, d_ip, n_rows + 1); DM(e, d_ix, std::max<int64_t>(nnz, 1));
    HIPCHK(e, hipMemcpy(*d_ip, indptr, (n_rows + 1) * 8, hipMemcpyHostToDevice));
    if (nnz) HIPCHK(e, hipMemcpy(*d_ix, indices, nnz * 4, hipMemcpyHostToDevice));
    return NTF_OK;
}
extern "C" int ntf_set_member_csr(ntf_engine* e, const int64_t* indptr, const int32_t* indices, int64_t n_rows) {
    if (!e) return NTF_EINVAL;
    int r = upload_csr(e, indptr, indices, n_rows, &e->m_indptr, &e->m_indices, e->Mg);   // labels keep GLOBAL expert ids on an expert shard
    if (r) return r;
    e->m_rows = n_rows;
    e->h_m_indptr.assign(indptr, indptr + n_rows + 1);
    e->h_m_indices.assign(indices, indices + indptr[n_rows]);
    return NTF_OK;
}
extern "C" int ntf_set_skill_csr(ntf_engine* e, const int64_t* indptr, const int32_t* indices, int64_t n_rows) {
    if (!e) return NTF_EINVAL;
    const int width = e->cfg.input_mode == NTF_INPUT_MULTIHOT ? e->cfg.dims[0] : (e->n_skills ? (int)e->n_skills : INT32_MAX);
    int r = upload_csr(e, indptr, indices, n_rows, &e->s_indptr, &e->s_indices, width);
    if (r) return r;
    e->s_rows = n_rows;
    e->s_max_col = -1;   // remembered so that a table set LATER can be checked against the column ids already resident
    for (int64_t p = 0; p < indptr[n_rows]; ++p) e->s_max_col = std::max(e->s_max_col, indices[p]);
    return NTF_OK;
}
extern "C" int ntf_set_skill_table(ntf_engine* e, const float* table, int64_t n_skills, int32_t d) {
    if (!e || !table || n_skills < 1 || d < 1) return NTF_EINVAL;
    if (e->cfg.input_mode == NTF_INPUT_MEANPOOL && d != e->cfg.dims[0]) FAIL(e, NTF_EINVAL, "skill table width != dims[0]");
    if (e->s_indptr && e->s_max_col >= n_skills) FAIL(e, NTF_EINVAL, "skill table has fewer rows than the column ids of the resident skill CSR");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipStreamSynchronize(e->st));
    dfree(e->table);
    DM(e, &e->table, n_skills * d);
    HIPCHK(e, hipMemcpy(e->table, table, n_skills * d * 4, hipMemcpyHostToDevice));
    e->n_skills = n_skills; e->table_d = d;
    return NTF_OK;
}
extern "C" int ntf_set_dense_input(ntf_engine* e, const float* X, int64_t n_rows, int32_t d) {
    if (!e || !X || n_rows < 1) return NTF_EINVAL;
    if (d != e->cfg.dims[0]) FAIL(e, NTF_EINVAL, "dense input width != dims[0]");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipStreamSynchronize(e->st));
    dfree(e->Xall);
    DM(e, &e->Xall, n_rows * d);
    HIPCHK(e, hipMemcpy(e->Xall, X, n_rows * d * 4, hipMemcpyHostToDevice));
    e->x_rows = n_rows;
    return NTF_OK;
}

// Vose alias table over weights w[0..n)
static void build_alias(const double* w, int64_t n, std::vector<float>& prob, std::vector<int32_t>& alias, double& total) {
    total = 0; for (int64_t i = 0; i < n; ++i) total += w[i];
    prob.assign(n, 1.f); alias.resize(n);
    for (int64_t i = 0; i < n; ++i) alias[i] = (int32_t)i;
    if (!(total > 0)) return;
    std::vector<double> sc(n);
    std::vector<int64_t> small, large;
    for (int64_t i = 0; i < n; ++i) { sc[i] = w[i] * (double)n / total; (sc[i] < 1.0 ? small : large).push_back(i); }
    while (!small.empty() && !large.empty()) {
        int64_t s = small.back(); small.pop_back();
        int64_t l = large.back();
        prob[s] = (float)sc[s]; alias[s] = (int32_t)l;
        sc[l] = (sc[l] + sc[s]) - 1.0;
        if (sc[l] < 1.0) { large.pop_back(); small.push_back(l); }
    }
    for (int64_t i : large) prob[i] = 1.f;
    for (int64_t i : small) prob[i] = 1.f;
}
static int upload_alias(ntf_engine* e, const double* w, int64_t n) {
    std::vector<float> prob; std::vector<int32_t> alias; double total;
    build_alias(w, n, prob, alias, total);
    if (e->al_n != n) { dfree(e->al_prob); dfree(e->al_alias); dfree(e->al_weight); DM(e, &e->al_prob, n); DM(e, &e->al_alias, n); DM(e, &e->al_weight, n); e->al_n = n; }
    HIPCHK(e, hipStreamSynchronize(e->st));
    HIPCHK(e, hipMemcpy(e->al_prob, prob.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->al_alias, alias.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->al_weight, w, n * 8, hipMemcpyHostToDevice));
    e->al_total = total;
    return NTF_OK;
}
extern "C" int ntf_set_unigram(ntf_engine* e, const double* freq, int64_t n) {
    if (!e || !freq) return NTF_EINVAL;
    if (n != e->Mg) FAIL(e, NTF_EINVAL, "unigram length != number of experts (of the whole output layer)");
    for (int64_t i = 0; i < n; ++i) if (!(freq[i] >= 0)) FAIL(e, NTF_EINVAL, "unigram: negative or NaN weight");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    return upload_alias(e, freq, n);
}

// ------------------------------------------------------------------------------------------ state
// Multi-hot input: layer 0's weight-shaped segments (weight, rho_weight; parameters, gradients, Adam moments alike) live TRANSPOSED in HBM, [S, H]
// instead of the reference's [H, S] (k_multihot_fwd / _bwd read and scatter whole rows).  Everything elementwise (Adam, Flipout operands, KL, the
// all-reduce) is layout-blind; the host boundary transposes.
static bool stored_transposed(const ntf_engine* e, int layer, int kind) {
    return e->cfg.input_mode == NTF_INPUT_MULTIHOT && layer == 0 && (kind == NTF_P_WEIGHT || kind == NTF_P_RHO_WEIGHT);
}
static void transpose_host(const float* src, float* dst, int64_t rows, int64_t cols) {   // dst [cols, rows] = src [rows, cols]^T, blocked
    constexpr int64_t T = 64;
    for (int64_t r0 = 0; r0 < rows; r0 += T)
        for (int64_t c0 = 0; c0 < cols; c0 += T)
            for (int64_t r = r0; r < std::min(rows, r0 + T); ++r)
                for (int64_t c = c0; c < std::min(cols, c0 + T); ++c) dst[c * rows + r] = src[r * cols + c];
}

static int param_span(ntf_engine* e, int layer, int kind, int64_t& off, int64_t& n) {
    if (layer < 0 || layer >= e->L || kind < 0 || kind > 3) FAIL(e, NTF_EINVAL, "param: bad layer/kind");
    if (kind >= 2 && !e->cfg.bayesian) FAIL(e, NTF_EINVAL, "param: rho on a non-bayesian model");
    off = e->layers[layer].off[kind];
    n = (kind == NTF_P_WEIGHT || kind == NTF_P_RHO_WEIGHT) ? e->layers[layer].nw() : e->layers[layer].out;
    return NTF_OK;
}
extern "C" int ntf_set_param(ntf_engine* e, int layer, int kind, const float* host, int64_t count) {
    if (e) e->pre_valid = false;   // prefetched operands were made from the old parameters
    if (!e || !host) return NTF_EINVAL;
    int64_t off, n; int r = param_span(e, layer, kind, off, n); if (r) return r;
    if (count != n) FAIL(e, NTF_EINVAL, "param: element count mismatch");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipStreamSynchronize(e->st));
    if (stored_transposed(e, layer, kind)) {
        std::vector<float> t((size_t)n);
        transpose_host(host, t.data(), e->layers[0].out, e->layers[0].in);   // [H, S] -> [S, H]
        HIPCHK(e, hipMemcpy(e->P + off, t.data(), n * 4, hipMemcpyHostToDevice));
        return NTF_OK;
    }
    HIPCHK(e, hipMemcpy(e->P + off, host, n * 4, hipMemcpyHostToDevice));
    return NTF_OK;
}
static int fetch_segment(ntf_engine* e, const float* dev, int layer, int kind, float* host, int64_t n) {
    if (!stored_transposed(e, layer, kind)) { HIPCHK(e, hipMemcpy(host, dev, n * 4, hipMemcpyDeviceToHost)); return NTF_OK; }
    std::vector<float> t((size_t)n);
    HIPCHK(e, hipMemcpy(t.data(), dev, n * 4, hipMemcpyDeviceToHost));
    transpose_host(t.data(), host, e->layers[0].in, e->layers[0].out);       // [S, H] -> [H, S]
    return NTF_OK;
}
extern "C" int ntf_get_param(ntf_engine* e, int layer, int kind, float* host, int64_t count) {
    if (!e || !host) return NTF_EINVAL;
    int64_t off, n; int r = param_span(e, layer, kind, off, n); if (r) return r;
    if (count != n) FAIL(e, NTF_EINVAL, "param: element count mismatch");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return fetch_segment(e, e->P + off, layer, kind, host, n);
}
extern "C" int ntf_get_grad(ntf_engine* e, int layer, int kind, float* host, int64_t count) {
    if (!e || !host) return NTF_EINVAL;
    int64_t off, n; int r = param_span(e, layer, kind, off, n); if (r) return r;
    if (count != n) FAIL(e, NTF_EINVAL, "grad: element count mismatch");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if ((r = join_side(e))) return r;
    HIPCHK(e, hipStreamSynchronize(e->st));
    return fetch_segment(e, e->G + off, layer, kind, host, n);
}
extern "C" int ntf_reset_optimizer(ntf_engine* e) {
    if (!e) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipMemsetAsync(e->M1, 0, e->n_params * 4, e->st));
    HIPCHK(e, hipMemsetAsync(e->V2, 0, e->n_params * 4, e->st));
    e->adam_t = 0; e->lr = e->cfg.lr;
    return NTF_OK;
}
extern "C" int ntf_set_lr(ntf_engine* e, float lr) { if (!e) return NTF_EINVAL; e->lr = lr; return NTF_OK; }
extern "C" int ntf_set_seed(ntf_engine* e, uint64_t seed, uint64_t step) { if (!e) return NTF_EINVAL; e->seed = seed; e->step = step; e->pre_valid = false; e->pre0_step = ~0ull; e->hp.valid = false; return NTF_OK; }
extern "C" int ntf_skip_step(ntf_engine* e) { if (!e) return NTF_EINVAL; e->step += 1; return NTF_OK; }
extern "C" int ntf_prefetched_steps(ntf_engine* e, int64_t* steps) { if (!e || !steps) return NTF_EINVAL; *steps = e->pre_used; return NTF_OK; }
extern "C" int ntf_head_prefetch_hits(ntf_engine* e, int64_t* steps) { if (!e || !steps) return NTF_EINVAL; *steps = e->hp_used; return NTF_OK; }
extern "C" int ntf_first_layer_sweeps(ntf_engine* e, int64_t* steps) { if (!e || !steps) return NTF_EINVAL; *steps = e->pre0_used; return NTF_OK; }
extern "C" int ntf_range_fallbacks(ntf_engine* e, int64_t* steps) {
    if (!e || !steps) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    int32_t dev = 0;
    HIPCHK(e, hipMemcpyAsync(&dev, e->d_range + 1, 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    *steps = (int64_t)dev + e->range_fallbacks_host;
    return NTF_OK;
}

// ------------------------------------------------------------------------------------------ step

// team ids a step may name: below the row count of EVERY resident matrix it reads (input source and, when set, the member CSR)
static int64_t row_limit(const ntf_engine* e) {
    int64_t lim = INT64_MAX;
    if (e->m_indptr) lim = std::min(lim, e->m_rows);
    if (e->cfg.input_mode == NTF_INPUT_DENSE) { if (e->Xall) lim = std::min(lim, e->x_rows); }
    else if (e->s_indptr) lim = std::min(lim, e->s_rows);
    return lim == INT64_MAX ? 0 : lim;
}

static int stage_rows(ntf_engine* e, const int64_t* rows, int B, bool rows_on_device, const int64_t** dev) {
    if (!rows || B < 1 || B > e->cfg.max_batch) FAIL(e, NTF_EINVAL, "step: bad rows/B (B must be in [1, max_batch])");
    if (rows_on_device) { *dev = rows; return NTF_OK; }
    const int64_t limit = row_limit(e);
    for (int i = 0; i < B; ++i) if (rows[i] < 0 || rows[i] >= limit) FAIL(e, NTF_EINVAL, "step: row id out of range");
    HIPCHK(e, hipMemcpyAsync(e->d_rows, rows, (size_t)B * 8, hipMemcpyHostToDevice, e->st));
    // the host buffer may be reused by the caller right after return
    HIPCHK(e, hipStreamSynchronize(e->st));
    *dev = e->d_rows;
    return NTF_OK;
}

static int stage_inj(ntf_engine* e, float** slot, const float* host, int64_t n) {
    if (!*slot) DM(e, slot, n);
    HIPCHK(e, hipMemcpyAsync(*slot, host, (size_t)n * 4, hipMemcpyHostToDevice, e->st));
    return NTF_OK;
}

static SignSpec sign_spec(ntf_engine* e, const StepCtx& c, int layer, int tensor, int ld) {
    SignSpec s; s.enabled = 1; s.ld = ld;
    float* inj = tensor == T_S_IN ? e->inj_s_in[layer] : e->inj_s_out[layer];
    const float* h = c.inj ? (tensor == T_S_IN ? c.inj->s_in[layer] : c.inj->s_out[layer]) : nullptr;
    s.inj = h ? inj : nullptr;
    make_key(e, c.step, layer, tensor, s.k0, s.k1);
    s.k0 += c.row0 * 0x9E3779B1u;  // sign_word(k0 + row0*phi, k1, r, .) == sign_word(k0, k1, r + row0, .): shifts the row index, kernels unchanged
    if (e->ep && layer == e->L - 1 && tensor == T_S_OUT) s.k1 += (uint32_t)(e->ep_lo >> 5) * 0x85EBCA77u;   // likewise the column block: an expert shard draws the whole layer's signs
    return s;
}
static NormalSpec normal_spec(ntf_engine* e, const StepCtx& c, int layer, int tensor) {
    NormalSpec s;
    float* inj = tensor == T_EPS_W ? e->inj_eps_w[layer] : e->inj_eps_b[layer];
    const float* h = c.inj ? (tensor == T_EPS_W ? c.inj->eps_w[layer] : c.inj->eps_b[layer]) : nullptr;
    s.inj = h ? inj : nullptr;
    make_key(e, c.step, layer, tensor, s.k0, s.k1);
    s.tag = (uint32_t)(layer * 8 + tensor); s.step = (uint32_t)c.step;
    if (e->ep && layer == e->L - 1) s.qbase = tensor == T_EPS_W ? (int64_t)e->ep_lo * e->layers[layer].in / 4 : e->ep_lo / 4;
    return s;
}

// split-product arithmetic of the fused output layer: number of planes and the exact power-of-two scales of fp16x3
static inline int mfma_np(const ntf_engine*) { return 2; }   // fp16x3: two fp16 planes per operand (the three-plane bf16x6 arithmetic was retired in round 6)
constexpr float kW16Scale = 256.f, kH16Scale = 16.f;          // weights (|w| << 256), hidden activations (|h| << 4096)
static inline float dz_scale16(const ntf_engine* e, int global_B) {   // |dz| <= max(tpw, tnw) / B  ->  scaled below 2^14
    const float dzmax = std::max(std::max(e->cfg.tpw, e->cfg.tnw), 1e-30f) / (float)std::max(global_B, 1);
    return std::exp2(std::floor(std::log2(16384.f / dzmax)));
}

// the range flag of the fp16x3 arithmetic, or null when this engine never multiplies in fp16x3
static int32_t* range_ptr(const ntf_engine* e) {
    return (fused_ok(e) && (e->cfg.mfma == NTF_MFMA_DEFAULT || e->cfg.mfma == NTF_MFMA_FP16X3)) ? e->d_range : nullptr;
}

static int check_ready(ntf_engine* e, bool need_labels) {
    if (need_labels && !e->m_indptr) FAIL(e, NTF_ESTATE, "member CSR not set (ntf_set_member_csr)");
    switch (e->cfg.input_mode) {
        case NTF_INPUT_DENSE: if (!e->Xall) FAIL(e, NTF_ESTATE, "dense input not set (ntf_set_dense_input)"); break;
        case NTF_INPUT_MEANPOOL: if (!e->table || !e->s_indptr) FAIL(e, NTF_ESTATE, "skill CSR / table not set"); break;
        case NTF_INPUT_MULTIHOT: if (!e->s_indptr) FAIL(e, NTF_ESTATE, "skill CSR not set"); break;
        default: FAIL(e, NTF_EINVAL, "bad input_mode");
    }
    return NTF_OK;
}

static int stage_all_inj(ntf_engine* e, const StepCtx& c) {
    if (!c.inj) return NTF_OK;
    for (int l = 0; l < e->L; ++l) {
        const LayerInfo& li = e->layers[l];
        int r;
        if (c.inj->eps_w[l]) {
            if (stored_transposed(e, l, NTF_P_WEIGHT)) {   // the producer walks the stored [S, H] order: stage the injected [H, S] tensor transposed
                std::vector<float> t((size_t)li.nw());
                transpose_host(c.inj->eps_w[l], t.data(), li.out, li.in);
                if ((r = stage_inj(e, &e->inj_eps_w[l], t.data(), li.nw()))) return r;
                HIPCHK(e, hipStreamSynchronize(e->st));   // t goes out of scope
            } else if ((r = stage_inj(e, &e->inj_eps_w[l], c.inj->eps_w[l], li.nw()))) return r;
        }
        if (c.inj->eps_b[l] && (r = stage_inj(e, &e->inj_eps_b[l], c.inj->eps_b[l], li.out))) return r;
        if (c.inj->s_in[l]) { if (!e->inj_s_in[l]) DM(e, &e->inj_s_in[l], (int64_t)e->cfg.max_batch * li.in);
            HIPCHK(e, hipMemcpyAsync(e->inj_s_in[l], c.inj->s_in[l], (size_t)c.B * li.in * 4, hipMemcpyHostToDevice, e->st)); }
        if (c.inj->s_out[l]) { if (!e->inj_s_out[l]) DM(e, &e->inj_s_out[l], (int64_t)e->cfg.max_batch * li.out);
            HIPCHK(e, hipMemcpyAsync(e->inj_s_out[l], c.inj->s_out[l], (size_t)c.B * li.out * 4, hipMemcpyHostToDevice, e->st)); }
    }
    if (c.inj->neg_idx && e->cfg.ns > 0)
        HIPCHK(e, hipMemcpyAsync(e->d_neg_set[c.step & 1], c.inj->neg_idx, (size_t)c.B * e->cfg.ns * 8, hipMemcpyHostToDevice, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return NTF_OK;
}

// X = act[0] for the batch (A1/A2 of SURVEY.md §8a)
static int make_input(ntf_engine* e, const StepCtx& c) {
    if (e->cfg.input_mode == NTF_INPUT_MULTIHOT) return NTF_OK;  // layer 0 reads the skill CSR itself
    Scope t(e, F_GATHER);
    const int D = e->cfg.dims[0];
    if (e->cfg.input_mode == NTF_INPUT_DENSE) launch_gather_dense_rows(e->st, e->Xall, D, c.rows_dev, c.B, e->act[0]);
    else launch_gather_meanpool(e->st, e->s_indptr, e->s_indices, e->table, c.rows_dev, c.B, D, 1, e->act[0]);
    return NTF_OK;
}

// forward through every layer.  want_logits: the last layer's leaky_relu output goes to dZout (inference);
// otherwise its pre-activation goes to Zout (loss).  hidden_only stops before the last layer (fused path).
static int forward_layers(ntf_engine* e, const StepCtx& c, bool want_logits, bool hidden_only) {
    const int B = c.B;
    for (int l = 0; l < e->L; ++l) {
        const LayerInfo& li = e->layers[l];
        const bool last = (l == e->L - 1);
        if (last && hidden_only) break;
        const float* in = e->act[l];
        float* W = e->P + li.off[NTF_P_WEIGHT]; float* b = e->P + li.off[NTF_P_BIAS];
        if (e->cfg.bayesian) {
            Scope t(e, F_FLIPOUT_OPERAND);
            if (last) e->pre_valid = false;   // overwrites the output layer's Wp
            const bool kl = !want_logits;  // loss steps: this layer's KL rides on the producer's pass over rho (and mu)
            const double share = (e->ep && !last) ? 1.0 / (double)e->ep_world : 1.0;   // expert shards: a replicated layer's KL is counted once over the shards
            if (!(l == 0 && e->use_pre0))      // (use_pre0: sigma * eps of this step and its KL term came out of the previous step's launch_flipout_sweep)
            launch_flipout_perturb(e->st, e->P + li.off[NTF_P_RHO_WEIGHT], kl ? W : nullptr, li.nw(), normal_spec(e, c, l, T_EPS_W), e->Wp[l],
                                   share / (double)li.nw(), e->d_kl);
            launch_flipout_perturb(e->st, e->P + li.off[NTF_P_RHO_BIAS], kl ? b : nullptr, li.out, normal_spec(e, c, l, T_EPS_B), e->bp[l],
                                   share / (double)li.out, e->d_kl);
        }
        if (l == 0 && e->cfg.input_mode == NTF_INPUT_MULTIHOT) {
            Scope t(e, F_MULTIHOT);
            SignSpec si, so;
            if (e->cfg.bayesian) { si = sign_spec(e, c, 0, T_S_IN, li.in); so = sign_spec(e, c, 0, T_S_OUT, li.out); }
            launch_multihot_fwd(e->st, c.rows_dev, B, li.in, li.out, e->s_indptr, e->s_indices, W, b, e->cfg.bayesian ? e->Wp[0] : nullptr,
                                e->cfg.bayesian ? e->bp[0] : nullptr, si, so, e->act[1]);
            continue;
        }
        Scope t(e, last ? F_OUT_FWD : F_GEMM_HIDDEN);
        GemmArgs g;
        g.M = B; g.N = li.out; g.K = li.in;
        g.A = in; g.sam = li.in; g.sak = 1;
        g.B = W; g.sbk = 1; g.sbn = li.in;
        g.bias = b;
        float* zbuf = last ? e->Zout : e->Zh;
        float* actdst = last ? (want_logits ? e->dZout : nullptr) : e->act[l + 1];
        if (!e->cfg.bayesian) {
            g.C = last ? (want_logits ? nullptr : zbuf) : nullptr; g.ldc = li.out;
            g.Act = actdst; g.ldact = li.out;
            launch_gemm(e->st, g);
        } else {
            g.C = zbuf; g.ldc = li.out;
            launch_gemm(e->st, g);
            GemmArgs p = g;
            p.B = e->Wp[l]; p.bias = e->bp[l];
            p.sa = sign_spec(e, c, l, T_S_IN, li.in); p.sa_t = 0;
            p.sc = sign_spec(e, c, l, T_S_OUT, li.out);
            p.accumulate = 1;
            p.Act = actdst; p.ldact = li.out;
            launch_gemm(e->st, p);
        }
    }
    return NTF_OK;
}

static int sample_negatives(ntf_engine* e, const StepCtx& c) {
    if (e->cfg.nsd == NTF_NSD_NONE || e->cfg.ns == 0) return NTF_OK;
    if (c.inj && c.inj->neg_idx) return NTF_OK;  // staged already
    Scope t(e, F_SAMPLER);
    uint32_t k0, k1; make_key(e, c.step, 0, T_NEG, k0, k1);
    const int M = e->Mg;   // negatives are drawn over the WHOLE output layer (an expert shard keeps the ones it owns, k_out_special)
    if (e->cfg.nsd == NTF_NSD_UNIFORM) {
        launch_ns_uniform(e->st, c.rows_dev, c.B, M, e->cfg.ns, e->m_indptr, e->m_indices, k0, k1, (uint32_t)c.step, c.row0, e->d_neg_set[c.step & 1]);
    } else if (e->cfg.nsd == NTF_NSD_UNIGRAM) {
        if (!e->al_prob) FAIL(e, NTF_ESTATE, "unigram table not set (ntf_set_unigram)");
        launch_ns_alias(e->st, c.rows_dev, c.B, M, e->cfg.ns, e->m_indptr, e->m_indices, e->al_prob, e->al_alias, e->al_weight, e->al_total,
                        k0, k1, (uint32_t)c.step, c.row0, e->d_neg_set[c.step & 1]);
    } else if (e->cfg.nsd == NTF_NSD_UNIGRAM_B) {
        const char* d = static_cast<const char*>(e->ub_dev[c.ub]);
        const size_t n = (size_t)e->ub_nsup[c.ub];
        launch_ns_alias_sparse(e->st, c.rows_dev, c.B, M, e->cfg.ns, e->m_indptr, e->m_indices, reinterpret_cast<const int32_t*>(d),
                               reinterpret_cast<const float*>(d + n * 8), reinterpret_cast<const int32_t*>(d + n * 4),
                               reinterpret_cast<const float*>(d + n * 12), e->ub_nsup[c.ub], e->ub_total[c.ub], k0, k1, (uint32_t)c.step, c.row0, e->d_neg_set[c.step & 1]);
    } else FAIL(e, NTF_EINVAL, "bad nsd");
    return NTF_OK;
}

// unigram_b (src/mdl/fnn.py:74-76): per-batch expert frequency y.sum(0)/B over the GLOBAL batch rows (host ids).  The table has
// support only on the batch's experts: sort + count them on the host (a few thousand entries), alias over the support, staged through
// two pinned buffers so that the host may prepare step t+1 while step t runs.
// `slot`: which of the two staging sets; `st`: the stream the upload is ordered on (the one whose sampler reads the table)
static int set_batch_unigram(ntf_engine* e, const int64_t* global_rows_host, int n, int slot, hipStream_t st) {
    std::vector<int32_t>& ent = e->ub_entries;
    ent.clear();
    for (int i = 0; i < n; ++i) {
        const int64_t t = global_rows_host[i];
        for (int64_t p = e->h_m_indptr[t]; p < e->h_m_indptr[t + 1]; ++p) ent.push_back(e->h_m_indices[p]);
    }
    std::sort(ent.begin(), ent.end());
    std::vector<int32_t> cols; std::vector<double> w;
    const float invn = 1.0f / (float)n;  // the reference keeps this table in f32
    for (size_t i = 0; i < ent.size();) {
        size_t j = i; while (j < ent.size() && ent[j] == ent[i]) ++j;
        cols.push_back(ent[i]); w.push_back((double)((float)(j - i) * invn));
        i = j;
    }
    const int nsup = (int)cols.size();
    std::vector<float> prob; std::vector<int32_t> alias; double total = 0;
    build_alias(w.data(), nsup, prob, alias, total);
    // staging: [cols | alias | prob | weight] per slot
    const size_t need = (size_t)std::max(nsup, 1) * 16;
    if (e->ub_cap < need) {
        // every stream, not the two named here: called from prefetch_next_head e->st IS e->st4 (the main stream, whose in-flight step may still read its own slot's
        // table, would not be waited for).  Growing is rare (the capacity doubles): a device-wide wait costs nothing that matters
        HIPCHK(e, hipDeviceSynchronize());
        e->hp.ub = -1;
        for (int k = 0; k < 2; ++k) { if (e->ub_host[k]) hipHostFree(e->ub_host[k]); if (e->ub_dev[k]) hipFree(e->ub_dev[k]); e->ub_host[k] = nullptr; e->ub_dev[k] = nullptr; }
        e->ub_cap = need * 2;
        for (int k = 0; k < 2; ++k) { HIPCHK(e, hipHostMalloc(&e->ub_host[k], e->ub_cap, hipHostMallocDefault)); HIPCHK(e, hipMalloc(&e->ub_dev[k], e->ub_cap)); }
        if (!e->ub_ev[0]) { HIPCHK(e, hipEventCreateWithFlags(&e->ub_ev[0], hipEventDisableTiming)); HIPCHK(e, hipEventCreateWithFlags(&e->ub_ev[1], hipEventDisableTiming)); }
        e->ub_used[0] = e->ub_used[1] = false;
    }
    if (e->ub_used[slot]) HIPCHK(e, hipEventSynchronize(e->ub_ev[slot]));  // the copy that last read this pinned buffer has completed
    char* h = static_cast<char*>(e->ub_host[slot]);
    std::memcpy(h, cols.data(), (size_t)nsup * 4);
    std::memcpy(h + (size_t)nsup * 4, alias.data(), (size_t)nsup * 4);
    std::memcpy(h + (size_t)nsup * 8, prob.data(), (size_t)nsup * 4);
    float* hw = reinterpret_cast<float*>(h + (size_t)nsup * 12);
    for (int k = 0; k < nsup; ++k) hw[k] = (float)w[k];
    HIPCHK(e, hipMemcpyAsync(e->ub_dev[slot], h, (size_t)nsup * 16, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipEventRecord(e->ub_ev[slot], st));
    e->ub_used[slot] = true; e->ub_nsup[slot] = nsup; e->ub_total[slot] = total;
    return NTF_OK;
}

struct StreamRestore { ntf_engine* e; hipStream_t main; ~StreamRestore() { e->st = main; } };   // launches and timing scopes follow e->st
static int side_stream(ntf_engine* e) {
    if (!e->st3) { HIPCHK(e, hipStreamCreateWithFlags(&e->st3, hipStreamNonBlocking)); HIPCHK(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming)); HIPCHK(e, hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming)); }
    if (!e->st4) { HIPCHK(e, hipStreamCreateWithFlags(&e->st4, hipStreamNonBlocking)); HIPCHK(e, hipEventCreateWithFlags(&e->ev_aux, hipEventDisableTiming)); }
    return NTF_OK;
}

// gather -> hidden layer -> operand images of one batch as ONE kernel (ntf_head.hip) into workspace `ws`; `want_planes`: also the dW kernel's h planes / s_in words.
// kl == null: no KL terms (they are in the step's sum already); with_bias: the extra workgroups that produce the output layer's bias operand; rflag: where an
// activation beyond the fp16 window is reported (this step's flag, or the next step's slot for a prefetched head).
// the multi-hot input's head (round 6): no one-kernel form (the first layer is a gather-sum over 90 671 rows, not a 128-wide product) - the same chain of launches a step used to
// issue inline, as a unit on stream `st` into workspace `ws`, so that it can be issued for the NEXT batch beside the dW kernel like k_head
static bool mh_head_ok(const ntf_engine* e) {
    return e->mh_head && e->head && e->L == 2 && e->cfg.input_mode == NTF_INPUT_MULTIHOT && e->cfg.bayesian && fused_ok(e) && e->layers[e->L - 1].in == 128 && e->pl_mu != nullptr &&
           (e->cfg.mfma == NTF_MFMA_DEFAULT || e->cfg.mfma == NTF_MFMA_FP16X3);
}
static void head_launch_multihot(ntf_engine* e, hipStream_t st, const StepCtx& c, char* ws, bool want_planes, double* kl, bool with_bias, int* rflag, bool rows_part, bool skip_w0) {
    const LayerInfo& l0 = e->layers[0]; const LayerInfo& lo = e->layers[e->L - 1];
    const int M = e->cfg.dims[e->L];
    float* W0 = e->P + l0.off[NTF_P_WEIGHT]; float* b0 = e->P + l0.off[NTF_P_BIAS];
    if (rows_part) {
        // Flipout operands of the first layer (k_flipout_perturb: + its KL terms unless they are in the step's sum already); the weights' come out of the previous step's sweep
        if (!skip_w0) launch_flipout_perturb(st, e->P + l0.off[NTF_P_RHO_WEIGHT], kl ? W0 : nullptr, l0.nw(), normal_spec(e, c, 0, T_EPS_W), e->Wp[0], 1.0 / (double)l0.nw(), kl);
        launch_flipout_perturb(st, e->P + l0.off[NTF_P_RHO_BIAS], kl ? b0 : nullptr, l0.out, normal_spec(e, c, 0, T_EPS_B), e->bp[0], 1.0 / (double)l0.out, kl);
        launch_multihot_fwd(st, c.rows_dev, c.B, l0.in, l0.out, e->s_indptr, e->s_indices, W0, b0, e->Wp[0], e->bp[0], sign_spec(e, c, 0, T_S_IN, l0.in), sign_spec(e, c, 0, T_S_OUT, l0.out), e->act[1]);
        // zero-padded h, h * s_in, the s_in words and the fp16x3 range check (k_prep_h: phase 1 of the fused forward with the planes marked ready), then the dW kernel's h planes
        FusedOut f;
        f.B = c.B; f.H = lo.in; f.M = M; f.bayes = 1; f.train = 1; f.h = e->act[1]; f.ws = ws;
        f.s_in = sign_spec(e, c, e->L - 1, T_S_IN, lo.in); f.s_out = sign_spec(e, c, e->L - 1, T_S_OUT, lo.out);
        f.bf16x6 = 1; f.np = mfma_np(e); f.h_scale = kH16Scale; f.w_scale = kW16Scale; f.planes_ready = 1; f.h_ready = 0; f.rflag = rflag;
        launch_fused_out_fwd(st, f, 1);
        if (want_planes) launch_fused_prep_planes(st, c.B, lo.in, M, 1, ws, mfma_np(e), kH16Scale, nullptr, 0, 2);
    }
    if (with_bias) launch_flipout_perturb(st, e->P + lo.off[NTF_P_RHO_BIAS], kl ? e->P + lo.off[NTF_P_BIAS] : nullptr, lo.out, normal_spec(e, c, e->L - 1, T_EPS_B), e->bp[e->L - 1], 1.0 / (double)e->Mg, kl);
}

static void head_launch(ntf_engine* e, hipStream_t st, const StepCtx& c, char* ws, bool want_planes, double* kl, bool with_bias, int* rflag, bool rows_part = true, bool skip_w0 = false) {
    if (e->cfg.input_mode == NTF_INPUT_MULTIHOT) { head_launch_multihot(e, st, c, ws, want_planes, kl, with_bias, rflag, rows_part, skip_w0); return; }
    const LayerInfo& l0 = e->layers[0]; const LayerInfo& lo = e->layers[e->L - 1];
    const int M = e->cfg.dims[e->L];
    const FusedWsPtrs wp = fused_ws_ptrs(ws, c.B, lo.in, M);
    HeadArgs a;
    a.B = c.B; a.Bpad = rows_part ? wp.Bpad : 0; a.D = l0.in; a.mode = e->cfg.input_mode == NTF_INPUT_DENSE ? 0 : 1; a.bayes = e->cfg.bayesian;
    a.rows = c.rows_dev; a.s_indptr = e->s_indptr; a.s_indices = e->s_indices; a.table = e->table; a.Xall = e->Xall;
    a.mu0 = e->P + l0.off[NTF_P_WEIGHT]; a.b0 = e->P + l0.off[NTF_P_BIAS];
    a.X = e->act[0]; a.act1 = e->act[1]; a.hz = wp.hz; a.hs = wp.hs; a.sinbits = wp.sinbits;
    a.hb = want_planes ? wp.hb : nullptr; a.sinT = wp.sinT;
    a.h_scale = kH16Scale;
    const bool guard = mfma_np(e) == 2 && rflag != nullptr;
    a.h_limit = guard ? 65504.f / kH16Scale : 0.f; a.rflag = guard ? rflag : nullptr;
    if (e->cfg.bayesian) {
        const double share = e->ep ? 1.0 / (double)e->ep_world : 1.0;   // expert shards: a replicated layer's KL is counted once over the shards
        a.rho0 = e->P + l0.off[NTF_P_RHO_WEIGHT]; a.rhob0 = e->P + l0.off[NTF_P_RHO_BIAS];
        a.eps_w0 = normal_spec(e, c, 0, T_EPS_W); a.eps_b0 = normal_spec(e, c, 0, T_EPS_B);
        a.si0 = sign_spec(e, c, 0, T_S_IN, l0.in); a.so0 = sign_spec(e, c, 0, T_S_OUT, l0.out); a.si1 = sign_spec(e, c, e->L - 1, T_S_IN, lo.in);
        a.klw_w0 = share / (double)l0.nw(); a.klw_b0 = share / (double)l0.out; a.kl = kl;
        if (with_bias) {
            a.M = lo.out; a.rho_b1 = e->P + lo.off[NTF_P_RHO_BIAS]; a.mu_b1 = e->P + lo.off[NTF_P_BIAS]; a.eps_b1 = normal_spec(e, c, e->L - 1, T_EPS_B); a.bp1 = e->bp[e->L - 1];
            a.klw_b1 = 1.0 / (double)e->Mg;
        }
    }
    launch_head(st, a);
}

// The output layer's dW (+ Adam + next-step operands) of a whole step: ONE k_out_dw_q launch.
// Round 5 experiment, kept for -DNTF_DIAG builds (NTF_DW_TAIL=1|2|3) because VERDICT r4 asked for it and the numbers are the answer: the launch cut at its last whole
// round of half-tiles (2 x CUs slots) - [0, rounds x slots) to k_out_dw_q, every slot the same number of tiles, the rest to the split-K form of k_out_dw_p2 +
// k_out_dw_finish (behind it, in front of it, or beside it on another stream).  Measured at config 2 (profiles/r5_dw_tail.md): the 1 536-tile launch alone 0.528 ms (three
// rounds cost 0.176 ms each; the fourth, partial round of the one-launch form costs only 0.07 ms: a CU left with one workgroup runs it faster), the tail 0.088 + 0.078 ms -
// 0.66-0.68 ms against 0.60.  Every launch pays its own ramp (first main loops: HBM idle) and tail (last epilogues: matrix pipe idle), ~0.1 ms; cutting the launch adds one.
static int dw_launch_whole(ntf_engine* e, const FusedDw& f) {
#ifndef NTF_DIAG
    launch_fused_out_dw(e->st, f);
    return NTF_OK;
#else
    const int slots = 2 * e->n_cu, total_q = (f.M + 127) / 128, full_q = total_q / slots * slots, tail_q = total_q - full_q;
    const bool can = e->dw_tail > 0 && f.dz_packed && f.adam && f.ksplit <= 1 && f.wg_count <= 0 && f.H == 128 && full_q > 0 && tail_q > 0 &&
                     tail_q * 10 <= slots * 8 && e->Zout != nullptr;
    if (!can) { launch_fused_out_dw(e->st, f); return NTF_OK; }
    const int tail_p2 = (tail_q + 1) / 2, nib = fused_ldb(f.B) / 32;
    // K ranges per tail tile: as many as fill whole rounds of one 256-expert workgroup per CU (two rounds unless the tail is tiny), at least 4 K blocks each
    int ks = std::max(1, std::min({2 * e->n_cu / tail_p2, 8, std::max(1, nib / 4)}));
    if (const char* v = getenv("NTF_DW_TAIL_KS")) ks = std::max(1, atoi(v));
    if (ks < 2 || fused_dw_part_floats(tail_p2 * 256, f.H, ks) > (int64_t)e->cfg.max_batch * e->cfg.dims[e->L]) { launch_fused_out_dw(e->st, f); return NTF_OK; }
    FusedDw fm = f; fm.wg_begin = 0; fm.wg_count = full_q / 2; fm.no_fallback = 1;
    FusedDw ft = f; ft.wg_begin = full_q / 2; ft.wg_count = tail_p2; ft.ksplit = ks; ft.part = e->Zout; ft.no_fallback = 1;
    if (e->dw_tail == 3) {
        if (!e->st2) { HIPCHK(e, hipStreamCreateWithFlags(&e->st2, hipStreamNonBlocking)); HIPCHK(e, hipEventCreateWithFlags(&e->ev_chunk, hipEventDisableTiming)); HIPCHK(e, hipEventCreateWithFlags(&e->ev_side, hipEventDisableTiming)); }
        HIPCHK(e, hipEventRecord(e->ev_chunk, e->st));
        HIPCHK(e, hipStreamWaitEvent(e->st2, e->ev_chunk, 0));
        launch_fused_out_dw(e->st, fm);
        launch_fused_out_dw(e->st2, ft);
        HIPCHK(e, hipEventRecord(e->ev_side, e->st2));
        HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_side, 0));
    } else if (e->dw_tail == 2) { launch_fused_out_dw(e->st, ft); launch_fused_out_dw(e->st, fm); }
    else { launch_fused_out_dw(e->st, fm); launch_fused_out_dw(e->st, ft); }
    if (f.rflag) { FusedDw fb = f; fb.fallback_only = 1; launch_fused_out_dw(e->st, fb); }
    return NTF_OK;
#endif
}

// Data-parallel pipelining of the step's head (round 5).  The parameters of the output layer arrive by all-gather in the dW chunks' ranges (ntf_dw_chunk_range: 65 536
// experts each); instead of waiting for all of them, producing the operands of the whole layer and then launching the forward kernel, the step runs the operand producer
// and the forward kernel RANGE BY RANGE (up to four ranges of whole dW chunks), each behind its own chunks' all-gathers: while the forward kernel works on range j, RCCL
// moves range j + 1.  fwd_ranges: number of ranges (0: the shape / arithmetic does not allow it); range j = dW chunks [k0, k1), 64-expert tiles [t_lo, t_hi) on `ncg`
// column groups whose dh slabs / loss partials start at cg_off.
struct FwdRange { int k0, k1, t_lo, t_hi, ncg, cg_off; };
static int fwd_ranges(const ntf_engine* e, int B, FwdRange* out, int* ncg_tot) {
    if (!fused_ok(e) || !e->cfg.bayesian || e->layers[e->L - 1].in != 128 || !e->pl_wp || !e->pl_mu || e->L < 2) return 0;
    if (!(e->cfg.mfma == NTF_MFMA_DEFAULT || e->cfg.mfma == NTF_MFMA_FP16X3) || !(e->fwd_kernel < 0 || e->fwd_kernel == 5) || e->ep) return 0;
    const int M = e->cfg.dims[e->L], tile = fused_dw_tile(), total = (M + tile - 1) / tile, nchunk = (total + 255) / 256;
    const int nrb = fused_ldb(B) / 128;
    const int nf = std::min({4, nchunk, nrb});
    if (nf < 2) return 0;
    const int T = (M + 63) / 64;
    int off = 0;
    for (int j = 0; j < nf; ++j) {
        FwdRange r;
        r.k0 = (int)((int64_t)j * nchunk / nf); r.k1 = (int)((int64_t)(j + 1) * nchunk / nf);
        r.t_lo = std::min(T, r.k0 * 1024); r.t_hi = std::min(T, r.k1 * 1024);      // a dW chunk = 256 tiles of 256 experts = 1 024 tiles of 64
        r.ncg = std::max(1, std::min(256 / nrb, r.t_hi - r.t_lo)); r.cg_off = off; off += r.ncg;
        if (out) out[j] = r;
    }
    if (ncg_tot) *ncg_tot = off;
    return nf;
}

// forward + loss (+ backward into G when train).  Loss = sum_rows(...)/global_B + KL * (B/global_B)/global_B
// Head prefetch (round 4; expert shards: round 5).  Everything the NEXT batch of the staged order needs before its forward kernel and that does not depend on the output layer:
// Adam of the hidden layers (their gradients are complete), then gather -> hidden layer -> h images (k_head) on `head_st`; the negative sampler, the special-entry list and
// the transposed s_out words - rows and sign keys only - on the auxiliary stream, which waits for `aux_after` (an event already recorded on a stream that is behind this
// step's forward kernel).  Into the OTHER workspace set, with the KL terms and the range flag in the NEXT step's slots (beside what this step's dW epilogue puts there).
// The caller has checked that the next step's operands are being produced by this step's dW epilogue (pre_valid / pre_step) and that k_head serves the shape.
static int join_side(ntf_engine* e) {
    if (e->join_pending) { HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_join, 0)); e->join_pending = false; }
    return NTF_OK;
}
static int prefetch_next_head(ntf_engine* e, const StepCtx& c, hipStream_t head_st, hipEvent_t aux_after) {
    int r;
    const LayerInfo& lo = e->layers[e->L - 1];
    const int M = e->cfg.dims[e->L];
    StreamRestore restore{e, e->st};
    e->st = head_st;
    {   // Adam of the hidden layers: [0, first float of the output layer) - apply_adam then leaves that range alone
        Scope t(e, F_ADAM);
        const double b1 = 0.9, b2 = 0.999, tt = (double)(e->adam_t + 1);
        int64_t rg[4] = {0, lo.off[NTF_P_WEIGHT], 0, 0}; int nrg = 1;
        if (e->l0_swept) {      // (the first layer's weight / rho_weight segments took their update in launch_flipout_sweep: its bias up to rho_weight, and its rho_bias, remain - L == 2 here)
            const LayerInfo& l0 = e->layers[0];
            rg[0] = l0.off[NTF_P_BIAS]; rg[1] = l0.off[NTF_P_RHO_WEIGHT]; rg[2] = l0.off[NTF_P_RHO_BIAS]; rg[3] = lo.off[NTF_P_WEIGHT]; nrg = 2;
        }
        launch_adam_ranges(e->st, e->P, e->G, e->M1, e->V2, rg, nrg, e->lr, (float)b1, (float)b2, 1e-8f, (float)(1.0 - std::pow(b1, tt)), (float)std::sqrt(1.0 - std::pow(b2, tt)));
        e->hidden_adam_done = true;
    }
    StepCtx n; n.rows_dev = e->hp_next_rows; n.B = e->hp_next_B; n.global_B = e->hp_next_B; n.step = c.step + 1; n.train = true; n.row0 = 0;
    char* ws_next = e->fws_set[n.step & 1];
    // the sampler and the s_out words need the rows and the sign key only: on the auxiliary stream, from the fork on (beside the hidden backward; beside the HBM-bound
    // dW kernel the 30 MB of k_sign_words_T take ~90 us instead of 19 - behind the hidden backward on ONE stream the chain ended 14 us before the dW kernel)
    e->st = e->st4;
    HIPCHK(e, hipStreamWaitEvent(e->st4, aux_after, 0));
    int ub_next = -1;
    if (e->cfg.nsd == NTF_NSD_UNIGRAM_B && e->cfg.ns > 0) {
        // round 5: the next batch's per-batch alias table (host sort + count of its ~3 k experts, 16 B per slot through the other pinned buffer) is staged here too,
        // uploaded on the auxiliary stream in front of its sampler
        ub_next = c.ub ^ 1; n.ub = ub_next;
        if ((r = set_batch_unigram(e, e->hp_next_host, n.B, ub_next, e->st4))) return r;
    }
    if ((r = sample_negatives(e, n))) return r;
    if (e->cfg.bayesian) {
        const SignSpec so = sign_spec(e, n, e->L - 1, T_S_OUT, lo.out);
        Scope t(e, F_OUT_FUSED_AUX); launch_fused_prep_planes(e->st, n.B, lo.in, M, 1, ws_next, 2, kH16Scale, &so, 0, 1);
    }
    HIPCHK(e, hipEventRecord(e->ev_aux, e->st4));
    e->st = head_st;
    { Scope t(e, F_GEMM_HIDDEN); head_launch(e, e->st, n, ws_next, true, e->d_kl + 2, false, range_ptr(e) ? e->d_range + 4 : nullptr, true, e->pre0_step == n.step); }
    e->hp.valid = true; e->hp.step = n.step; e->hp.rows = n.rows_dev; e->hp.B = n.B; e->hp.ub = ub_next;
    return NTF_OK;
}
static bool head_prefetch_possible(const ntf_engine* e, const StepCtx& c, int B) {
    const bool can_head = (mh_head_ok(e) && !e->ep) || (e->head && e->L == 2 && e->cfg.input_mode != NTF_INPUT_MULTIHOT && head_supported(e->cfg.dims[0], e->cfg.dims[1]) &&
                          (e->cfg.mfma == NTF_MFMA_DEFAULT || e->cfg.mfma == NTF_MFMA_FP16X3));
    if (!e->cfg.bayesian && !e->fnn_pipe) return false;
    return e->head_prefetch && can_head && c.fuse_adam && e->cfg.fuse_adam == 1 && e->pre_valid && e->pre_step == c.step + 1 && e->hp_next_B > 0 && !c.inj &&
           (e->cfg.nsd != NTF_NSD_UNIGRAM_B || e->hp_next_host != nullptr) && c.global_B == B;
}

static int run_step(ntf_engine* e, const StepCtx& c, bool accumulate_epoch) {
    int r;
    const int B = c.B, M = e->cfg.dims[e->L];
    const float inv_B = 1.0f / (float)c.global_B;
    const LayerInfo& lo = e->layers[e->L - 1];
    const bool fused = fused_ok(e);
    const double out_nw = (double)e->Mg * lo.in, out_nb = (double)e->Mg;   // element counts of the WHOLE output layer (= lo.nw(), lo.out unless expert-sharded)
    int nslots;
    const int64_t* neg = (e->cfg.nsd != NTF_NSD_NONE && e->cfg.ns > 0) ? e->d_neg_set[c.step & 1] : nullptr;
    bool prod_side = false, aux = false, swt_aux = false, loss_side = false, use_pre = false;
    // one kernel for gather -> hidden layer -> operand images (ntf_head.hip): one hidden layer of 128 units over a dense / mean-pooled input, native generators, fp16x3 planes
    // (round 6: the multi-hot input's head is such a unit too - head_launch_multihot - on one GPU)
    const bool use_head = fused && c.part <= 1 && !c.inj &&
                          ((mh_head_ok(e) && !e->ep) || (e->head && e->L == 2 && e->cfg.input_mode != NTF_INPUT_MULTIHOT && head_supported(e->cfg.dims[0], e->cfg.dims[1]) &&
                                                         (e->cfg.mfma == NTF_MFMA_DEFAULT || e->cfg.mfma == NTF_MFMA_FP16X3)));
    // the hidden layers' backward (and the loss reduction) of a whole train step run on the side stream beside the output layer's dW kernel, see `backward:`
    // (round 5: also of a deferred-dW step - a data-parallel rank's - whose dW chunks the host launches right behind this call: the join then waits behind the last chunk, join_side)
    // (Fnn: the chain alone is too short to pay - it goes to the side stream when the next batch's head follows it there, round 6)
    const bool fnn_side = !e->cfg.bayesian && e->fnn_pipe && e->prefetch && e->head_prefetch && c.fuse_adam && e->cfg.fuse_adam == 1 && !c.inj && !e->ep && use_head && c.global_B == B &&
                          e->hp_next_B > 0 && (e->cfg.nsd != NTF_NSD_UNIGRAM_B || e->hp_next_host != nullptr) && range_ptr(e) && e->pl_mu && lo.in == 128;
    const bool side = fused && (e->cfg.bayesian || fnn_side) && e->L > 1 && c.train && c.part == 0 && (!c.defer_dw || e->dp_side_bwd) && e->side_bwd && !(c.fuse_adam && e->cfg.fuse_adam == 2);
    bool hp_hit = false, hp_stale = false;
    bool chain_ok = false, chain_lean = false, chain_fill = false, chain_nof32 = false; PerturbChain pch;
    FwdRange fr[4]; int fr_tot = 0;
    const int nfr = (c.chunk_cb && c.defer_dw && c.train && c.part == 0 && !c.inj && e->dp_ranges) ? fwd_ranges(e, B, fr, &fr_tot) : 0;      // > 0: producer + forward kernel range by range
    if (fused) e->fws = e->fws_set[c.step & 1];     // (every kernel of a step works in ONE of the two workspace sets: a prefetched head of step t + 1 fills the other beside step t's dW kernel)
    if (c.part >= 2) goto backward;   // expert-sharded step, later phases
    // the Flipout operand producers add each layer's KL to d_kl[0]; the 4 bytes behind it are this step's fp16x3 range flag
    {
        const bool pre0 = e->pre0_step == c.step;      // the previous step's launch_flipout_sweep left this step's first-layer operand and KL term (multi-hot Flipout layer 0)
        const bool pre_ok = fused && e->pre_valid && e->pre_step == c.step && (e->cfg.bayesian ? e->pl_wp != nullptr : e->pl_mu != nullptr) &&      // (Fnn, round 6: the planes of mu alone)
                            !(c.inj && e->cfg.bayesian && c.inj->eps_w[e->L - 1]) &&   // (an injected eps: the operands are made from it in this step)
                            !(pre0 && c.inj);                       // (likewise the first layer's: its KL term sits in the prefetched sum - the step starts from scratch)
        e->pre_valid = false;   // consumed, or stale
        use_pre = pre_ok;
        // the previous step's side stream may have run this batch's head already (see `head prefetch` below)
        const bool had = e->hp.valid; e->hp.valid = false;
        if (had && use_pre && !use_head) use_pre = false;     // (a step that cannot take ANY k_head - injected tensors - starts from scratch: the prefetched head's KL terms are dropped with the scalars)
        if (use_pre) e->pre_used += 1;
        if (had && use_pre && use_head) {
            hp_hit = c.train && e->hp.step == c.step && e->hp.rows == c.rows_dev && e->hp.B == B && !c.inj;
            hp_stale = !hp_hit;       // another batch than the one it was issued for: its KL terms (functions of the parameters alone) are in this step's sum already
            if (hp_hit) e->hp_used += 1;
        }
    }
    // the step's KL sum and fp16x3 range flag: zero, or the values the previous step's dW epilogue produced for this one (moved into place by that step's Adam launch
    // if pre_rotated, else here)
    // (an evaluation step of a chain that has the output layer's KL term kept: the sum starts from it, the lean producer below adds none)
    chain_ok = c.chain && !c.train && fused && e->cfg.bayesian && !use_pre && !nfr && !c.inj && e->pl_wp && e->pl_mu && mfma_np(e) == 2 && range_ptr(e) && e->d_chain;
    chain_lean = chain_ok && e->chain_valid; chain_fill = chain_ok && !e->chain_valid;
    if (chain_fill) HIPCHK(e, hipMemsetAsync(e->d_chain, 0, 16, e->st));
    if (chain_fill) { pch.kl_out2 = e->d_chain; pch.mu_flag_out = reinterpret_cast<int*>(e->d_chain + 1); e->chain_valid = true; }
    if (chain_lean) pch.raise_if = reinterpret_cast<const int*>(e->d_chain + 1);
    // a lean evaluation step of a chain reads the fp16 planes only: its producer writes no f32 copy of sigma * eps (120 of its 360 MB at config 2); a step whose range flag
    // is raised makes the copy itself, below (the conditional launch a train step on prefetched operands uses)
    chain_nof32 = chain_lean && e->lean && e->pl_wp != nullptr;
    if (e->cfg.bayesian || (range_ptr(e) && e->fnn_pipe)) { if (!(use_pre && e->pre_rotated)) launch_step_scalars(e->st, e->d_kl, use_pre ? 1 : 0, chain_lean ? e->d_chain : nullptr); }
    else if (range_ptr(e)) HIPCHK(e, hipMemsetAsync(e->d_range, 0, 4, e->st));
    e->pre_rotated = false;
    if (fused && e->side_bwd && !hp_hit) {
        // Three streams through the step's head (round 3; profiles/r3_step_timeline.md).  What precedes the forward kernel is a chain of small latency-bound
        // launches and one HBM-bound pass, and most links of it do not depend on each other:
        //   side (st3): the output layer's operand producer (eps, sigma, Wp, split planes, KL: one pass over 2 x M x H floats, 0.12 ms at config 2) - parameters only;
        //   aux  (st4): what needs the row ids / the sign key only - the bias producer, the negative sampler, the transposed s_out words of the dW kernel;
        //   main (st):  gather -> hidden layers -> h planes, which is then shorter than the producer beside it.
        if ((r = side_stream(e))) return r;
        HIPCHK(e, hipEventRecord(e->ev_fork, e->st));
        StreamRestore guard{e, e->st};
        if (e->cfg.bayesian && !use_pre && !nfr) {
            HIPCHK(e, hipStreamWaitEvent(e->st3, e->ev_fork, 0));
            e->st = e->st3;
            { Scope t(e, F_FLIPOUT_OPERAND);
              launch_flipout_perturb(e->st, e->P + lo.off[NTF_P_RHO_WEIGHT], chain_lean ? nullptr : e->P + lo.off[NTF_P_WEIGHT], lo.nw(), normal_spec(e, c, e->L - 1, T_EPS_W), chain_nof32 ? nullptr : e->Wp[e->L - 1],
                                     1.0 / out_nw, e->d_kl, e->pl_wp, (e->pl_wp && !chain_lean) ? e->pl_mu : nullptr, e->P + lo.off[NTF_P_WEIGHT], lo.in, mfma_np(e), kW16Scale, range_ptr(e), nullptr, pch); }
            HIPCHK(e, hipEventRecord(e->ev_join, e->st3));
            prod_side = true;
        }
        HIPCHK(e, hipStreamWaitEvent(e->st4, e->ev_fork, 0));
        e->st = e->st4;
        if (e->cfg.bayesian && !use_head) { Scope t(e, F_FLIPOUT_OPERAND);
            launch_flipout_perturb(e->st, e->P + lo.off[NTF_P_RHO_BIAS], e->P + lo.off[NTF_P_BIAS], lo.out, normal_spec(e, c, e->L - 1, T_EPS_B), e->bp[e->L - 1],
                                   1.0 / out_nb, e->d_kl); }
        if ((r = sample_negatives(e, c))) return r;
        if (c.train && e->cfg.bayesian && e->cfg.mfma != NTF_MFMA_F32 && mfma_np(e) == 2 && lo.in == 128 && e->pl_mu != nullptr) {
            const SignSpec so = sign_spec(e, c, e->L - 1, T_S_OUT, lo.out), si = sign_spec(e, c, e->L - 1, T_S_IN, lo.in);
            if (so.inj == nullptr && si.inj == nullptr) {   // (injected signs: the words are transposed from the packed image k_sign_bits writes on the main stream)
                Scope t(e, F_OUT_FUSED_AUX);
                launch_fused_prep_planes(e->st, B, lo.in, M, 1, e->fws, 2, kH16Scale, &so, 0, 1);
                swt_aux = true;
            }
        }
        HIPCHK(e, hipEventRecord(e->ev_aux, e->st4));
        aux = true;
    }
    if (!use_head && (r = make_input(e, c))) return r;
    if (!aux && !hp_hit) { if ((r = sample_negatives(e, c))) return r; }
    if (fused) {
        e->use_pre0 = use_pre && e->pre0_step == c.step;      // (decided with use_pre: a step that drops the prefetched scalars drops the first layer's KL term with them)
        e->pre0_step = ~0ull;
        if (e->use_pre0) e->pre0_used += 1;
        const bool skip_w0 = e->use_pre0;      // (the multi-hot head below takes the same decision forward_layers does)
        if (!use_head && (r = forward_layers(e, c, false, true))) { e->use_pre0 = false; return r; }
        e->use_pre0 = false;
        FusedOut f;
        f.B = B; f.H = lo.in; f.M = M; f.bayes = e->cfg.bayesian; f.train = c.train;
        f.h = e->act[e->L - 1];
        f.mu = e->P + lo.off[NTF_P_WEIGHT]; f.mu_b = e->P + lo.off[NTF_P_BIAS];
        f.tnw = e->cfg.tnw; f.tpw = e->cfg.tpw; f.inv_B = inv_B;
        f.dzT = e->dZout; f.dh_slab = e->dh_slab; f.ws = e->fws;
        f.dh = e->L > 1 ? e->dAct[(e->L - 1) & 1] : nullptr;
        f.h_mask = e->L > 1 ? e->act[e->L - 1] : nullptr;
        if (e->cfg.bayesian) {
            if (!prod_side && !use_pre && !nfr) { Scope t(e, F_FLIPOUT_OPERAND);
              launch_flipout_perturb(e->st, e->P + lo.off[NTF_P_RHO_WEIGHT], chain_lean ? nullptr : f.mu, lo.nw(), normal_spec(e, c, e->L - 1, T_EPS_W), chain_nof32 ? nullptr : e->Wp[e->L - 1],
                                     1.0 / out_nw, e->d_kl, e->pl_wp, (e->pl_wp && !chain_lean) ? e->pl_mu : nullptr, f.mu, lo.in, mfma_np(e), kW16Scale, range_ptr(e), nullptr, pch); }   // + the split planes of Wp and mu
            if (!aux && !use_head) { Scope t(e, F_FLIPOUT_OPERAND);
              launch_flipout_perturb(e->st, e->P + lo.off[NTF_P_RHO_BIAS], f.mu_b, lo.out, normal_spec(e, c, e->L - 1, T_EPS_B), e->bp[e->L - 1],
                                     1.0 / out_nb, e->d_kl); }
            f.planes_ready = e->pl_wp != nullptr;
            f.wp = e->Wp[e->L - 1]; f.bp = e->bp[e->L - 1];
            f.s_in = sign_spec(e, c, e->L - 1, T_S_IN, lo.in); f.s_out = sign_spec(e, c, e->L - 1, T_S_OUT, lo.out);
        }
        if (!e->cfg.bayesian && use_pre) f.planes_ready = 1;      // (Fnn, round 6: the previous step's dW epilogue wrote the planes of the updated mu)
        f.bf16x6 = e->pl_mu != nullptr; f.mu_pl = e->pl_mu; f.wp_pl = e->pl_wp;
        f.np = mfma_np(e); f.w_scale = kW16Scale; f.h_scale = kH16Scale; f.dz_scale = dz_scale16(e, c.global_B);
        f.rflag = range_ptr(e);
        if (e->fwd_kernel >= 0) f.wide = e->fwd_kernel;   // A/B runs: NTF_FWD_KERNEL = 0, 1, 2 (ntf_fused.h), read when the engine is created
        f.eval_kernel = e->eval_kernel;
        f.rows = c.rows_dev; f.m_indptr = e->m_indptr; f.m_indices = e->m_indices; f.neg = neg; f.ns = e->cfg.ns; f.row_fix = e->row_fix;
        f.c_lo = e->ep_lo;
        if (use_head) {
            Scope t(e, F_GEMM_HIDDEN);
            // (hp_hit: this batch's head ran beside the previous step's dW kernel; hp_stale: a head ran for another batch - redo it here, without the KL terms and the bias operand)
            if (!hp_hit) head_launch(e, e->st, c, e->fws, c.train && e->cfg.mfma != NTF_MFMA_F32, hp_stale ? nullptr : e->d_kl, !hp_stale, f.rflag, true, skip_w0);
            f.h_ready = 1;
            if (!f.planes_ready) launch_fused_out_fwd(e->st, f, 1);   // (Fnn: the split planes of mu are made per step)
            if (c.train && e->cfg.mfma != NTF_MFMA_F32 && !swt_aux && !hp_hit) {   // (one stream: the s_out words were not made beside the head)
                const bool dz_packed = f.np == 2 && lo.in == 128 && e->pl_mu != nullptr;
                launch_fused_prep_planes(e->st, B, lo.in, M, e->cfg.bayesian, e->fws, f.np, f.h_scale, dz_packed ? &f.s_out : nullptr, 0, 1);
            }
        } else {
        { Scope t(e, F_OUT_FUSED_AUX); launch_fused_out_fwd(e->st, f, 1); }
            if (c.train && e->cfg.mfma != NTF_MFMA_F32) {
                // operands of the dW kernel that depend on h and on the sign keys only (split planes of h / h*s_in, transposed s_out words): prepared here, in the
                // step's head (beside the side-stream producer), not between the forward and the dW kernel
                Scope t(e, F_OUT_FUSED_AUX);
                const bool dz_packed = f.np == 2 && lo.in == 128 && e->pl_mu != nullptr;
                const bool so_inj = e->cfg.bayesian && (f.s_out.inj != nullptr || f.s_in.inj != nullptr);
                launch_fused_prep_planes(e->st, B, lo.in, M, e->cfg.bayesian, e->fws, f.np, f.h_scale, dz_packed ? &f.s_out : nullptr, so_inj, swt_aux ? 2 : 3);
            }
        }
        if (prod_side) HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_join, 0));
        if (aux) HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_aux, 0));   // (the sparse fix-up reads the negatives, the dW kernel the s_out words: both long done by now)
        if (chain_nof32 || (use_pre && e->cfg.bayesian && e->lean && !(hp_hit && e->f32_copy_step == c.step + 1))) {      // (f32_copy_step: the previous step's Adam launch carried this job - good only if the head that
                                                                                      // ran beside that step IS this batch's: a head redone here may raise the flag anew)
            // this step's operands came from the previous step's dW epilogue, which (lean) left no f32 copy of sigma * eps: only a step that falls back to the exact-f32 kernels
            // reads one, and makes it here - a capped grid that exits at once unless the range flag is raised (behind the head: k_head may still raise it)
            Scope t(e, F_FLIPOUT_OPERAND);
            launch_flipout_perturb(e->st, e->P + lo.off[NTF_P_RHO_WEIGHT], nullptr, lo.nw(), normal_spec(e, c, e->L - 1, T_EPS_W), e->Wp[e->L - 1], 0.0, nullptr,
                                   nullptr, nullptr, nullptr, 0, 3, 1.f, nullptr, range_ptr(e));
        }
#ifdef NTF_DIAG
        if (e->cosched > 0 && c.train) {
            f.ncg_limit = e->cosched;
            static const bool fwd_only = getenv("NTF_COSCHED_FWD_ONLY") != nullptr;      // the forward kernel alone on its reduced grid (no dW at all in the step)
            if (e->cosched_have && !fwd_only) {
                if ((r = side_stream(e))) return r;
                if (!e->ev_co0) { HIPCHK(e, hipEventCreateWithFlags(&e->ev_co0, hipEventDisableTiming)); HIPCHK(e, hipEventCreateWithFlags(&e->ev_co1, hipEventDisableTiming)); }
                HIPCHK(e, hipEventRecord(e->ev_co0, e->st));
                { Scope t(e, F_OUT_FUSED_FWD); launch_fused_out_fwd(e->st, f, 2); }       // enqueued first: its workgroups take their CUs, the dW workgroups the rest
                HIPCHK(e, hipStreamWaitEvent(e->st3, e->ev_co0, 0));
                { StreamRestore guard{e, e->st}; e->st = e->st3; Scope t(e, F_OUT_FUSED_DW); launch_fused_out_dw(e->st, e->cosched_dw); }
                HIPCHK(e, hipEventRecord(e->ev_co1, e->st3));
                HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_co1, 0));
            } else { Scope t(e, F_OUT_FUSED_FWD); launch_fused_out_fwd(e->st, f, 2); }
        } else
#endif
        if (nfr) {
            // range by range: the caller's callback orders this stream behind the all-gathers of the range's parameter chunks; then its operands (eps, sigma eps, the
            // fp16 planes of sigma eps and mu, its share of the KL) and its forward launch.  The whole-layer exact-f32 launch of a range fallback follows the last range.
            f.planes_ready = 1; f.split_fallback = 1; f.chunk_ncg_tot = fr_tot;
            for (int j = 0; j < nfr; ++j) {
                if (c.chunk_cb(j, c.chunk_user) != 0) FAIL(e, NTF_ESTATE, "step_staged_deferred_cb: the chunk callback reported a failure");
                const int64_t r0 = (int64_t)fr[j].t_lo * 64, r1 = std::min<int64_t>((int64_t)fr[j].t_hi * 64, M), o0 = r0 * lo.in, n0 = (r1 - r0) * lo.in;
                { Scope t(e, F_FLIPOUT_OPERAND);
                  NormalSpec es = normal_spec(e, c, e->L - 1, T_EPS_W); es.qbase += o0 / 4;
                  const int64_t pl0 = r0 / 32 * (32 * mfma_np(e)) * lo.in;      // the planes of 32-expert tile r0 / 32 (fused_planes_elems layout)
                  launch_flipout_perturb(e->st, e->P + lo.off[NTF_P_RHO_WEIGHT] + o0, f.mu + o0, n0, es, e->Wp[e->L - 1] + o0, 1.0 / out_nw, e->d_kl,
                                         e->pl_wp + pl0, e->pl_mu + pl0, f.mu + o0, lo.in, mfma_np(e), kW16Scale, range_ptr(e)); }
                f.chunk_t_lo = fr[j].t_lo; f.chunk_t_hi = fr[j].t_hi; f.chunk_cg_off = fr[j].cg_off; f.chunk_ncg = fr[j].ncg;
                { Scope t(e, F_OUT_FUSED_FWD); launch_fused_out_fwd(e->st, f, 2); }
            }
            f.chunk_ncg = 0;
            { Scope t(e, F_OUT_FUSED_FWD); launch_fused_out_fwd(e->st, f, 8); }
        } else
        { Scope t(e, F_OUT_FUSED_FWD); launch_fused_out_fwd(e->st, f, 2); }
        { Scope t(e, F_OUT_FUSED_AUX); launch_fused_out_fwd(e->st, f, 4); }
        nslots = fused_loss_slots(M);
    } else {
        if ((r = forward_layers(e, c, false, false))) return r;
        Scope t(e, F_LOSS);
        nslots = loss_dense_nchunk(M);
        launch_loss_dense(e->st, e->Zout, M, B, M, e->cfg.tnw, inv_B, c.train ? e->dZout : nullptr, e->partial, nslots);
        launch_loss_special(e->st, e->Zout, M, B, M, c.rows_dev, e->m_indptr, e->m_indices, neg, e->cfg.ns, e->cfg.tpw, e->cfg.tnw, inv_B,
                            c.train ? e->dZout : nullptr, e->row_fix);
    }
    loss_side = side;   // a whole train step: the loss reduction is not on the way to the dW kernel - it goes first on the side stream, beside it
    if (!loss_side) {
        Scope t(e, F_LOSS);
        const double kl_scale = ((double)B / (double)c.global_B) / (double)c.global_B;
        launch_loss_finalize(e->st, e->partial, nslots, e->row_fix, B, inv_B, e->cfg.bayesian ? e->d_kl : nullptr, kl_scale, e->d_loss,
                             accumulate_epoch ? e->d_acc : nullptr, e->d_acc_steps);
    }
    if (!c.train || c.part == 1) return NTF_OK;

backward:
    const float kl_share = (float)B / (float)c.global_B;
    // Whole step on one GPU: the hidden layers' backward (a chain of small kernels, ~0.1 ms) needs d(hidden) only, not the output layer's dW kernel (0.4-0.6 ms,
    // whose last round leaves CUs idle): it runs on a side stream beside it; both are joined before Adam.
    StreamRestore restore{e, e->st};
    if (side) {
        if ((r = side_stream(e))) return r;
        HIPCHK(e, hipEventRecord(e->ev_fork, e->st));
        HIPCHK(e, hipStreamWaitEvent(e->st3, e->ev_fork, 0));
        if (loss_side) {
            e->st = e->st3;
            { Scope t(e, F_LOSS);
              const double kl_scale = ((double)B / (double)c.global_B) / (double)c.global_B;
              launch_loss_finalize(e->st, e->partial, nslots, e->row_fix, B, inv_B, e->cfg.bayesian ? e->d_kl : nullptr, kl_scale, e->d_loss,
                                   accumulate_epoch ? e->d_acc : nullptr, e->d_acc_steps); }
            e->st = restore.main;
        }
    }
    bool l0_sweep_now = false;
    for (int l = e->L - 1; l >= 0; --l) {
        const LayerInfo& li = e->layers[l];
        const bool last = (l == e->L - 1);
        if (c.part == 2 && !last) break;       // the hidden layers wait for the sum of d(hidden) over the expert shards
        if (c.part == 3 && last) continue;
        if (side && l == e->L - 2) e->st = e->st3;   // from here on: launches and timing scopes on the side stream (which waits for ev_fork, above)
        const float* in = e->act[l];
        float* gW = e->G + li.off[NTF_P_WEIGHT]; float* gb = e->G + li.off[NTF_P_BIAS];
        float* gRW = e->cfg.bayesian ? e->G + li.off[NTF_P_RHO_WEIGHT] : nullptr;
        float* gRb = e->cfg.bayesian ? e->G + li.off[NTF_P_RHO_BIAS] : nullptr;
        SignSpec sin_, sout_;
        if (e->cfg.bayesian) { sin_ = sign_spec(e, c, l, T_S_IN, li.in); sout_ = sign_spec(e, c, l, T_S_OUT, li.out); }
        if (last && fused) {
            FusedDw f;
            f.B = B; f.H = li.in; f.M = M; f.bayes = e->cfg.bayesian;
            f.dzT = e->dZout; f.h = in; f.g_mu = gW; f.g_rho = gRW; f.g_b = gb; f.g_bp = gRb; f.ws = e->fws;
            f.mu = e->P + li.off[NTF_P_WEIGHT];
            f.s_out = sout_; f.s_out_inj = e->cfg.bayesian && (sout_.inj != nullptr || sin_.inj != nullptr);
            f.bf16x6 = e->cfg.mfma != NTF_MFMA_F32;   // default: bf16x6
            f.np = mfma_np(e); f.a_scale = dz_scale16(e, c.global_B); f.h_scale = kH16Scale; f.rflag = range_ptr(e);
            f.dz_packed = f.bf16x6 && f.np == 2 && li.in == 128 && e->pl_mu != nullptr;   // the fp16x3 forward kernels (H = 128) store packed plane pairs
            e->last_dz_packed_scale = f.dz_packed ? f.a_scale : 0.f;
            if (f.dz_packed && !c.defer_dw && !(c.fuse_adam && e->cfg.fuse_adam == 2)) {
                // few expert tiles (a narrow expert shard under a wide minibatch, or a small model) leave most CUs idle at one workgroup per 256 experts:
                // split every tile's K (batch) range over several workgroups.  Scratch: the dense-logits buffer of the generic path, idle in a fused step.
                const int tiles = (M + fused_dw_tile() - 1) / fused_dw_tile(), nib = fused_ldb(B) / 32;
                int ks = e->dw_ksplit > 0 ? e->dw_ksplit : (tiles <= 128 ? std::min({256 / tiles, 8, std::max(1, nib / 4)}) : 1);
                while (ks > 1 && fused_dw_part_floats(M, li.in, ks) > (int64_t)e->cfg.max_batch * M) --ks;
                if (ks > 1) { f.ksplit = ks; f.part = e->Zout; }
            }
            // (launch_fused_prep_planes - the h planes and the transposed s_out words this kernel reads - ran in the step's head)
            if (e->cfg.bayesian) { f.rho = e->P + li.off[NTF_P_RHO_WEIGHT]; f.wp = e->Wp[l]; f.klw = kl_share / ((float)out_nw * (float)c.global_B); f.cur_eps = normal_spec(e, c, l, T_EPS_W); }
            if (c.defer_dw) {
                const int tile = fused_dw_tile(), total = (M + tile - 1) / tile;
                e->pend = f; e->pend_valid = true; e->pend_chunks = (total + 255) / 256;
                if (e->cfg.bayesian) { e->pend_eps_b = normal_spec(e, c, l, T_EPS_B); e->pend_klw_b = kl_share / ((float)out_nb * (float)c.global_B); }
                continue;  // its bias-gradient finalisation follows the last chunk
            }
            if (c.fuse_adam && e->cfg.fuse_adam == 2) {
                // chunked: dW of expert chunk k on the main stream, Adam of chunk k on the side stream while dW of chunk k+1 runs
                if (!e->st2) { HIPCHK(e, hipStreamCreateWithFlags(&e->st2, hipStreamNonBlocking)); HIPCHK(e, hipEventCreateWithFlags(&e->ev_chunk, hipEventDisableTiming)); HIPCHK(e, hipEventCreateWithFlags(&e->ev_side, hipEventDisableTiming)); }
                const double b1 = 0.9, b2 = 0.999, tt = (double)(e->adam_t + 1);
                const float bc1 = (float)(1.0 - std::pow(b1, tt)), bc2s = (float)std::sqrt(1.0 - std::pow(b2, tt));
                const int tile = fused_dw_tile(), total = (M + tile - 1) / tile, chunk = 256;   // one workgroup per CU per round
                Scope t(e, F_OUT_FUSED_DW);
                for (int w0 = 0; w0 < total; w0 += chunk) {
                    f.wg_begin = w0; f.wg_count = chunk;
                    launch_fused_out_dw(e->st, f);
                    HIPCHK(e, hipEventRecord(e->ev_chunk, e->st));
                    HIPCHK(e, hipStreamWaitEvent(e->st2, e->ev_chunk, 0));
                    const int64_t lo = (int64_t)w0 * tile * li.in, hi = std::min<int64_t>((int64_t)(w0 + chunk) * tile, M) * li.in;
                    for (int kind : {NTF_P_WEIGHT, NTF_P_RHO_WEIGHT}) {
                        if (kind == NTF_P_RHO_WEIGHT && !e->cfg.bayesian) continue;
                        const int64_t o = li.off[kind] + lo;
                        launch_adam(e->st2, e->P + o, e->G + o, e->M1 + o, e->V2 + o, hi - lo, e->lr, (float)b1, (float)b2, 1e-8f, bc1, bc2s);
                    }
                }
                HIPCHK(e, hipEventRecord(e->ev_side, e->st2));
                HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_side, 0));  // later kernels on the main stream read the updated parameters
                e->adam_in_dw = true;
                goto dw_done;
            }
            if (c.fuse_adam) {
                const double b1 = 0.9, b2 = 0.999;
                const double tt = (double)(e->adam_t + 1);
                const double bc1 = 1.0 - std::pow(b1, tt), bc2 = 1.0 - std::pow(b2, tt);
                f.adam = 1; f.lr_over_bc1 = e->lr / (float)bc1; f.b1 = (float)b1; f.b2 = (float)b2; f.eps = 1e-8f; f.bc2_sqrt = (float)std::sqrt(bc2);
                f.w_mu = e->P + li.off[NTF_P_WEIGHT]; f.m_mu = e->M1 + li.off[NTF_P_WEIGHT]; f.v_mu = e->V2 + li.off[NTF_P_WEIGHT];
                if (e->cfg.bayesian) { f.w_rho = e->P + li.off[NTF_P_RHO_WEIGHT]; f.m_rho = e->M1 + li.off[NTF_P_RHO_WEIGHT]; f.v_rho = e->V2 + li.off[NTF_P_RHO_WEIGHT]; }
                e->adam_in_dw = true;
                if (e->prefetch && e->cfg.bayesian && f.dz_packed && e->pl_wp && e->pl_mu && range_ptr(e)) {
                    // the Adam epilogue holds the updated mu / rho: it is also the operand producer of step + 1 (FusedDw.produce)
                    StepCtx nx; nx.step = c.step + 1;
                    f.produce = 1; f.lean = e->lean; f.nx_eps = normal_spec(e, nx, l, T_EPS_W); f.nx_wp = e->Wp[l]; f.nx_pl_wp = e->pl_wp; f.nx_pl_mu = e->pl_mu; f.nx_pscale = kW16Scale;
                    f.nx_klw = 1.0 / out_nw; f.nx_kl = e->d_kl + 2; f.nx_rflag = e->d_range + 4;
                    e->pre_valid = true; e->pre_step = c.step + 1;
                }
                if (e->prefetch && e->fnn_pipe && !e->cfg.bayesian && f.dz_packed && e->pl_mu && range_ptr(e)) {
                    // Fnn: the epilogue's updated mu goes out as the next step's fp16 planes too (no eps, no KL: the range flag of the planes is all that travels with them)
                    f.produce = 1; f.nx_pl_mu = e->pl_mu; f.nx_pl_wp = nullptr; f.nx_pscale = kW16Scale; f.nx_kl = e->d_kl + 2; f.nx_rflag = e->d_range + 4;
                    e->pre_valid = true; e->pre_step = c.step + 1;
                }
                if (e->cfg.bayesian) { e->fin_pend = true; e->fin_eps = normal_spec(e, c, l, T_EPS_B); e->fin_klw = kl_share / ((float)out_nb * (float)c.global_B); }
            }
#ifdef NTF_DIAG
            if (e->cosched > 0) { e->cosched_dw = f; e->cosched_have = true; goto dw_done; }      // (this step's dW rides beside the NEXT step's forward kernel: timing only)
#endif
            { Scope t(e, F_OUT_FUSED_DW); if ((r = dw_launch_whole(e, f))) return r; }
        dw_done:;
        } else {
            const float* dZ = last ? e->dZout : e->dAct[(l + 1) & 1];
            { Scope t(e, F_BIAS_GRAD); launch_bias_grad(e->st, dZ, li.out, B, li.out, sout_, gb, gRb, e->gemm_slab, kGemmSlabFloats); }
            if (l == 0 && e->cfg.input_mode == NTF_INPUT_MULTIHOT) {
                Scope t(e, F_MULTIHOT);
                // round 6: in a step whose Adam is applied in this call (the output layer's already ran in its dW epilogue), this layer's gradient goes straight into
                // launch_flipout_sweep below, which also clears the rows it reads: no memset of the two 46-MB buffers, a byte per skill row says which rows the scatter touched
                l0_sweep_now = e->l0_sweep && e->cfg.bayesian && c.fuse_adam && e->adam_in_dw && !c.defer_dw && c.part == 0 && !e->ep && e->l0_touched &&
                               (li.nw() & 3) == 0 && (li.out & 3) == 0;
                if (!(l0_sweep_now && e->g0_clean)) {
                    HIPCHK(e, hipMemsetAsync(gW, 0, (size_t)li.nw() * 4, e->st));
                    if (gRW) HIPCHK(e, hipMemsetAsync(gRW, 0, (size_t)li.nw() * 4, e->st));
                }
                e->g0_clean = false; e->pre0_step = ~0ull;
                if (l0_sweep_now) HIPCHK(e, hipMemsetAsync(e->l0_touched, 0, (size_t)li.in, e->st));
                launch_multihot_bwd(e->st, c.rows_dev, B, li.in, li.out, e->s_indptr, e->s_indices, dZ, sin_, sout_, gW, gRW, l0_sweep_now ? e->l0_touched : nullptr);
            } else {
                Scope t(e, last ? F_OUT_BWD_DW : F_GEMM_HIDDEN);
                GemmArgs g;
                g.M = li.out; g.N = li.in; g.K = B;
                g.A = dZ; g.sam = 1; g.sak = li.out;
                g.B = in; g.sbk = li.in; g.sbn = 1;
                g.C = gW; g.ldc = li.in;
                const int wtiles = ((li.out + 63) / 64) * ((li.in + 63) / 64);
                if (wtiles < 64 && B >= 256) { g.ksplit = std::max(1, std::min({B >= 2048 ? 64 : 16, B / 64, (int)(kGemmSlabFloats / ((int64_t)li.out * li.in))})); g.slab = e->gemm_slab; }
                if (g.ksplit <= 1) { g.ksplit = 1; g.slab = nullptr; }
                launch_gemm(e->st, g);
                if (e->cfg.bayesian) { g.sa = sout_; g.sa_t = 1; g.sb = sin_; g.sb_t = 0; g.C = gRW; launch_gemm(e->st, g); }
            }
            if (l > 0) {
                Scope t(e, last ? F_OUT_BWD_DA : F_GEMM_HIDDEN);
                float* dA = e->dAct[l & 1];
                GemmArgs g;
                g.M = B; g.N = li.in; g.K = li.out;
                g.A = dZ; g.sam = li.out; g.sak = 1;
                g.B = e->P + li.off[NTF_P_WEIGHT]; g.sbk = li.in; g.sbn = 1;
                g.C = dA; g.ldc = li.in;
                g.mask = e->act[l]; g.ldmask = li.in;
                const int tiles = ((B + 63) / 64) * ((li.in + 63) / 64);
                int ks = 1;
                if (li.out >= 4096) ks = std::max(1, std::min({64, 1024 / tiles, (int)(kGemmSlabFloats / ((int64_t)B * li.in))}));
                g.ksplit = ks; g.slab = e->gemm_slab;
                launch_gemm(e->st, g);
                if (e->cfg.bayesian) {
                    g.B = e->Wp[l]; g.sa = sout_; g.sa_t = 0; g.sc = sin_;
                    g.accumulate = 1;
                    launch_gemm(e->st, g);
                }
            }
        }
        if (e->cfg.bayesian) {
            Scope t(e, F_FLIPOUT_FINAL);
            if (l == 0 && l0_sweep_now) {
                // finalize + Adam + the next step's sigma * eps and KL term of the 90 671 x 128 pairs in ONE pass (was: k_flipout_grad_finalize here, the flat Adam behind
                // the join, k_flipout_perturb at the head of the next step - 0.28 + 0.11 + 0.09 ms at config 3)
                FlipoutSweep a;
                a.mu = e->P + li.off[NTF_P_WEIGHT]; a.rho = e->P + li.off[NTF_P_RHO_WEIGHT]; a.g_mu = gW; a.g_rho = gRW;
                a.m_mu = e->M1 + li.off[NTF_P_WEIGHT]; a.v_mu = e->V2 + li.off[NTF_P_WEIGHT]; a.m_rho = e->M1 + li.off[NTF_P_RHO_WEIGHT]; a.v_rho = e->V2 + li.off[NTF_P_RHO_WEIGHT];
                a.n = li.nw(); a.eps = normal_spec(e, c, l, T_EPS_W); a.klw = kl_share / ((float)li.nw() * (float)c.global_B);
                const double b1 = 0.9, b2 = 0.999, tt = (double)(e->adam_t + 1);
                a.lr = e->lr; a.b1 = (float)b1; a.b2 = (float)b2; a.adam_eps = 1e-8f; a.bc1 = (float)(1.0 - std::pow(b1, tt)); a.bc2_sqrt = (float)std::sqrt(1.0 - std::pow(b2, tt));
                StepCtx nx; nx.step = c.step + 1;
                const bool produce = e->pre_valid && e->pre_step == c.step + 1;      // (the dW epilogue produced the output layer's operands for that step: same protocol, same slot)
                a.nx_eps = normal_spec(e, nx, l, T_EPS_W); a.nx_wp = e->Wp[l]; a.nx_klw = 1.0 / (double)li.nw(); a.nx_kl = produce ? e->d_kl + 2 : nullptr;
                a.touched = e->l0_touched; a.H = li.out;
                launch_flipout_sweep(e->st, a);
                e->l0_swept = true; e->g0_clean = true;
                if (produce) e->pre0_step = c.step + 1;
            } else
            if (!(last && fused))  // the fused dW kernel finalises the output layer's weights in its epilogue
                launch_flipout_grad_finalize(e->st, e->P + li.off[NTF_P_WEIGHT], e->P + li.off[NTF_P_RHO_WEIGHT], gW, gRW, li.nw(),
                                             normal_spec(e, c, l, T_EPS_W), kl_share / ((float)li.nw() * (float)c.global_B));
            if (!(last && e->fin_pend))   // (fused-Adam step: the output layer's biases are finalised inside the Adam launch, apply_adam)
            launch_flipout_grad_finalize(e->st, e->P + li.off[NTF_P_BIAS], e->P + li.off[NTF_P_RHO_BIAS], gb, gRb, li.out,
                                         normal_spec(e, c, l, T_EPS_B), kl_share / ((last ? (float)out_nb : (float)li.out) * (float)c.global_B));
        }
    }
    if (side) {
        // ---- head prefetch: behind this step's hidden-layer backward the side stream is idle while the dW kernel runs (~0.3 ms at config 2).  Everything the NEXT batch of
        // the staged order needs before its forward kernel and that does not depend on the output layer goes there: Adam of the hidden layers (their gradients are
        // complete), the negative sampler, the transposed s_out words and k_head - into the other workspace set, with the KL terms and the range flag in the NEXT step's
        // slots (beside what this step's dW epilogue puts there).  What stays between the dW kernel and the next forward kernel: Adam of the output biases (+ the
        // rotation of the scalars), the bias operand (apply_adam) and the two range-fallback launches.  Taken by run_step when the next call IS that batch (hp_hit).
        if (head_prefetch_possible(e, c, B) && !e->ep) {
            if ((r = prefetch_next_head(e, c, e->st3, e->ev_fork))) return r;
            e->st = e->st3;
        }
        // the auxiliary stream (the next batch's sampler and sign words) joins the SIDE stream, so that the main stream waits once, not twice: a cross-stream wait is a
        // barrier packet in the main queue - four interleaved rounds on one box, 1.3645 against 1.3692 ms (round 5)
        if (e->hp.valid && e->hp.step == c.step + 1) HIPCHK(e, hipStreamWaitEvent(e->st3, e->ev_aux, 0));
        HIPCHK(e, hipEventRecord(e->ev_join, e->st3));
        e->st = restore.main;
        if (c.defer_dw) e->join_pending = true;      // the dW chunks come next on the main stream, beside this chain
        else HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_join, 0));
    }
    const int ep_hp = !e->ep ? 0 : e->ep_head_prefetch >= 0 ? e->ep_head_prefetch : ((int64_t)B * 8 <= (int64_t)M ? 2 : 0);
    if (ep_hp && c.part == 3 && head_prefetch_possible(e, c, B)) {
        // ---- expert shards (round 5): phase 3 - the hidden layers' backward, on the stream that waited for the d(hidden) exchange - runs beside this shard's dW kernel (phase 2,
        // side stream).  Where that kernel outlasts the backward (ranks of 2-4: fewer rows, more experts per rank) the stream would idle until the join: the next batch's head
        // (hidden Adam, k_head; sampler and sign words on the auxiliary stream) goes there.  At a rank of 8 backward and dW end together (profiles/r5_ep/rank_of_8_step_timeline.txt)
        // and a narrow shard's split-K dW launch is ONE round of workgroups on every CU - no ramp, no tail: what is issued beside it only slows it (the s_out words take 0.32 ms
        // instead of 0.03 when they start with the kernel) - hence the rule in ep_head_prefetch's comment.  Same kernels on the same inputs either way (tests/test_gpu_ep.py).
        if ((r = side_stream(e))) return r;
        if (!e->ep_side || ep_hp == 2) HIPCHK(e, hipEventRecord(e->ev_fork, e->st));     // (ep_side: recorded behind phase 1, ntf_step_staged_ep; 2: the auxiliary stream starts here, behind the backward)
        if ((r = prefetch_next_head(e, c, e->st, e->ev_fork))) return r;
        HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_aux, 0));
    }
    return NTF_OK;
}

static int apply_adam(ntf_engine* e) {
    Scope t(e, F_ADAM);
    e->adam_t += 1;
    const double b1 = 0.9, b2 = 0.999;
    const double bc1 = 1.0 - std::pow(b1, (double)e->adam_t), bc2 = 1.0 - std::pow(b2, (double)e->adam_t);
    if (!e->adam_in_dw) { e->pre_valid = false; launch_adam(e->st, e->P, e->G, e->M1, e->V2, e->n_params, e->lr, (float)b1, (float)b2, 1e-8f, (float)bc1, (float)std::sqrt(bc2)); return NTF_OK; }
    // the output layer's weight (and rho_weight) segments were updated inside the dW kernel: one launch over the rest
    e->adam_in_dw = false;
    const LayerInfo& lo = e->layers[e->L - 1];
    const int64_t w0 = lo.off[NTF_P_WEIGHT], w1 = lo.off[NTF_P_BIAS];  // segments are laid out weight, bias, rho_weight, rho_bias
    int64_t rg[12]; int fin[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
    if (e->l0_swept && !e->hidden_adam_done) {
        // the first layer's weight / rho_weight segments were updated by launch_flipout_sweep: its bias (up to rho_weight), its rho_bias and the later hidden layers remain
        // (hidden_adam_done: the prefetched head's own Adam launch took them - prefetch_next_head)
        const LayerInfo& l0 = e->layers[0];
        rg[2 * n] = l0.off[NTF_P_BIAS]; rg[2 * n + 1] = l0.off[NTF_P_RHO_WEIGHT]; ++n;
        rg[2 * n] = l0.off[NTF_P_RHO_BIAS]; rg[2 * n + 1] = e->L > 2 ? e->layers[1].off[NTF_P_WEIGHT] : w0; ++n;
        if (e->L > 2) { rg[2 * n] = e->layers[1].off[NTF_P_WEIGHT]; rg[2 * n + 1] = w0; ++n; }
    } else
    if (!e->hidden_adam_done) { rg[2 * n] = 0; rg[2 * n + 1] = w0; ++n; }      // (head prefetch: the hidden layers' Adam ran on the side stream, in front of the next batch's head)
    e->hidden_adam_done = false; e->l0_swept = false;
    if (e->cfg.bayesian) { const int64_t r0 = lo.off[NTF_P_RHO_WEIGHT], r1 = lo.off[NTF_P_RHO_BIAS];
        // the live lo.out floats of the two bias segments, not their 256-byte padding: a finalised range would push the KL gradient into the padding of rho_bias
        // (p = 0, g = 0 there) and read an injected eps_b past its lo.out floats
        (void)r0;
        rg[2 * n] = w1; rg[2 * n + 1] = w1 + lo.out; fin[n] = e->fin_pend ? 1 : 0; ++n; rg[2 * n] = r1; rg[2 * n + 1] = r1 + lo.out; fin[n] = e->fin_pend ? 2 : 0; ++n; }
    else { rg[2 * n] = w1; rg[2 * n + 1] = w1 + lo.out; ++n; }
    const bool rotate = (e->cfg.bayesian || e->fnn_pipe) && e->pre_valid && e->pre_step == e->step;   // this step's dW epilogue left the next step's KL / range flag behind the current ones
    // the prefetched head of the next step left out what depends on the output layer's biases, which this launch updates: their Flipout operand sigma_b eps_b and KL of
    // the next step.  Round 5: produced by this very launch from the updated values it holds (round 4: k_head's bias workgroups alone in a launch behind it)
    const bool bias_nx = e->cfg.bayesian && e->hp.valid && rotate && e->hp.step == e->step && e->merge_bias;
    NormalSpec nx_eps;
    if (bias_nx) { StepCtx nx; nx.step = e->hp.step; nx.B = e->hp.B; nx.global_B = e->hp.B; nx_eps = normal_spec(e, nx, e->L - 1, T_EPS_B); }
#ifdef NTF_DIAG
    static const int diag_skip = getenv("NTF_SKIP") ? atoi(getenv("NTF_SKIP")) : 0;     // timing only (results garbage): 1 - no launch here; 2 - the rotation alone
    if (diag_skip == 1 && bias_nx) {} else if (diag_skip == 2 && bias_nx) launch_step_scalars(e->st, e->d_kl, 1); else
#endif
    // ... and, as extra workgroups of the same launch, the f32 copy of the next step's sigma * eps that only a step falling back to the exact-f32 kernels reads (lean: the dW
    // epilogue wrote none) - round 4 issued it as a launch of its own in front of every step, a no-op in all but the rarest.  *only_if: the NEXT step's range flag, still in
    // its slot behind the current one (this launch's last workgroup rotates it) and complete: the dW epilogue and the prefetched head are behind this launch
    F32CopyJob f32c; const bool with_copy = bias_nx && e->lean && e->f32_copy_merged && range_ptr(e) != nullptr;
    if (with_copy) {
        StepCtx nx; nx.step = e->hp.step; nx.B = e->hp.B; nx.global_B = e->hp.B;
        f32c.rho = e->P + lo.off[NTF_P_RHO_WEIGHT]; f32c.out = e->Wp[e->L - 1]; f32c.n = lo.nw(); f32c.eps = normal_spec(e, nx, e->L - 1, T_EPS_W); f32c.only_if = e->d_range + 4;
        e->f32_copy_step = e->hp.step + 1;      // (+ 1: 0 = never)
    }
    launch_adam_ranges(e->st, e->P, e->G, e->M1, e->V2, rg, n, e->lr, (float)b1, (float)b2, 1e-8f, (float)bc1, (float)std::sqrt(bc2), fin, &e->fin_eps, e->fin_klw,
                       rotate ? e->d_kl : nullptr, bias_nx ? e->bp[e->L - 1] : nullptr, bias_nx ? &nx_eps : nullptr, 1.0 / (double)e->Mg, with_copy ? &f32c : nullptr);
    e->pre_rotated = rotate; e->fin_pend = false;
    if (e->cfg.bayesian && e->hp.valid && rotate && e->hp.step == e->step && !bias_nx) {
        StepCtx nx; nx.step = e->hp.step; nx.B = e->hp.B; nx.global_B = e->hp.B; nx.rows_dev = e->hp.rows; nx.train = true;
        Scope t2(e, F_FLIPOUT_OPERAND);
        head_launch(e, e->st, nx, e->fws_set[nx.step & 1], false, e->d_kl, true, nullptr, false);
    }
    return NTF_OK;
}

// Adam on the given [lo, hi) float ranges of the flat buffers only (one optimiser step: the step count advances once).  Data parallel with
// sharded optimiser state: after a reduce-scatter every rank owns 1/G of each gradient range, updates that part, and the parameters are
// all-gathered - the 28 B/parameter of Adam's traffic are paid once per node instead of once per GPU.
static int apply_adam_ranges(ntf_engine* e, const int64_t* lo_hi, int n) {
    Scope t(e, F_ADAM);
    e->adam_t += 1;
    e->adam_in_dw = false; e->pre_valid = false;
    const double b1 = 0.9, b2 = 0.999;
    const double bc1 = 1.0 - std::pow(b1, (double)e->adam_t), bc2 = 1.0 - std::pow(b2, (double)e->adam_t);
    for (int k = 0; k < n; ++k) {
        const int64_t lo = lo_hi[2 * k], hi = lo_hi[2 * k + 1];
        if (lo < 0 || hi > e->n_params || lo > hi || (lo & 3)) FAIL(e, NTF_EINVAL, "apply_ranges: a range outside the flat buffers, reversed, or not 16-byte aligned");
    }
    launch_adam_ranges(e->st, e->P, e->G, e->M1, e->V2, lo_hi, n, e->lr, (float)b1, (float)b2, 1e-8f, (float)bc1, (float)std::sqrt(bc2));
    return NTF_OK;
}

static int read_loss(ntf_engine* e, float* loss_out) {
    if (!loss_out) return NTF_OK;
    HIPCHK(e, hipMemcpyAsync(loss_out, e->d_loss, 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return NTF_OK;
}

static int step_common(ntf_engine* e, const int64_t* rows, int32_t B, int32_t global_B, const ntf_inject* inj, float* loss_out, bool train,
                       bool apply, bool rows_on_device, const int64_t* global_rows_host, int n_global, bool defer_dw = false, uint32_t row0 = 0) {
    if (!e) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    int r;
    if ((r = check_ready(e, true))) return r;
    if (global_B < B) FAIL(e, NTF_EINVAL, "global_B < B");
    if (train && e->ep && e->ep_world > 1)
        FAIL(e, NTF_ESTATE, "expert-sharded engine: a train step needs the sum of d(hidden) over the shards - use ntf_step_staged_ep");
    if (e->ep && global_B != B) FAIL(e, NTF_EINVAL, "expert-sharded engine: every shard steps the whole minibatch (global_B == B)");
    if (e->ep_open == 2 && e->ep_side) HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_join, 0));
    e->ep_open = 0;
    if ((r = join_side(e))) return r;       // (an abandoned deferred step's side-stream chain still orders before this one)
    StepCtx c; c.B = B; c.global_B = global_B; c.inj = inj; c.train = train; c.step = e->step++;
    c.fuse_adam = train && apply && e->cfg.fuse_adam && global_B == B && fused_ok(e);
    c.chain = !train && e->eval_chain != 0 && !inj;
    c.defer_dw = defer_dw && train && !apply && fused_ok(e);
    c.row0 = row0;
    if (c.defer_dw) { c.chunk_cb = e->cb_fn; c.chunk_user = e->cb_user; }
    e->pend_valid = false;
    if ((r = stage_rows(e, rows, B, rows_on_device, &c.rows_dev))) return r;
    if ((r = stage_all_inj(e, c))) return r;
    if (e->cfg.nsd == NTF_NSD_UNIGRAM_B && e->cfg.ns > 0 && !(inj && inj->neg_idx)) {
        if (!global_rows_host) FAIL(e, NTF_EINVAL, "unigram_b needs the batch rows on the host");
        // the previous step's head prefetch may have staged THIS batch's table already (beside its dW kernel): the table is a function of the batch's rows alone
        if (e->hp.valid && e->hp.ub >= 0 && e->hp.step == c.step && e->hp.rows == c.rows_dev && e->hp.B == B && global_B == B) c.ub = e->hp.ub;
        else { c.ub = (e->ub_slot ^= 1); if ((r = set_batch_unigram(e, global_rows_host, n_global, c.ub, e->st))) return r; }
        e->ub_slot = c.ub;
    }
    if ((r = run_step(e, c, true))) return r;
    e->last_B = B; e->last_global_B = global_B; e->neg_step = c.step;
    if (train && apply && (r = apply_adam(e))) return r;
    if (loss_out && (r = join_side(e))) return r;      // (the loss reduction of a deferred step runs on the side stream)
    if ((r = read_loss(e, loss_out))) return r;
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) FAIL(e, NTF_EHIP, std::string("kernel launch: ") + hipGetErrorString(s));
    return NTF_OK;
}

extern "C" int ntf_train_step(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, float* loss_out) {
    return step_common(e, rows, B, B, inj, loss_out, true, true, false, rows, B);
}
extern "C" int ntf_eval_step(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, float* loss_out) {
    return step_common(e, rows, B, B, inj, loss_out, false, false, false, rows, B);
}
extern "C" int ntf_backward(ntf_engine* e, const int64_t* rows, int32_t B, int32_t global_B, const ntf_inject* inj, float* loss_out) {
    return step_common(e, rows, B, global_B, inj, loss_out, true, false, false, rows, B);
}
extern "C" int ntf_apply(ntf_engine* e) {
    if (!e) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (int r = join_side(e)) return r;
    return apply_adam(e);
}

extern "C" int ntf_apply_ranges(ntf_engine* e, const int64_t* lo_hi, int32_t n) {
    if (!e || (n > 0 && !lo_hi) || n < 0) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (int r = join_side(e)) return r;
    return apply_adam_ranges(e, lo_hi, n);
}

extern "C" int ntf_stage_order(ntf_engine* e, const int64_t* order, int64_t n) {
    if (!e || !order || n < 1) { if (e) e->err = "stage_order: bad arguments"; return NTF_EINVAL; }
    HIPCHK(e, hipSetDevice(e->cfg.device));
    const int64_t limit = row_limit(e);
    for (int64_t i = 0; i < n; ++i) if (order[i] < 0 || order[i] >= limit) FAIL(e, NTF_EINVAL, "stage_order: row id out of range");
    if (e->order_cap < n) { dfree(e->d_order); DM(e, &e->d_order, n); e->order_cap = n; }
    e->hp.rows = nullptr;      // a head prefetched from the previous order can never be this order's batch (its KL terms stay counted: run_step's `hp_stale`)
    e->h_order.assign(order, order + n);
    HIPCHK(e, hipMemcpyAsync(e->d_order, e->h_order.data(), (size_t)n * 8, hipMemcpyHostToDevice, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return NTF_OK;
}
extern "C" int ntf_step_staged(ntf_engine* e, int64_t offset, int32_t B, int64_t global_offset, int32_t global_B, int32_t train, int32_t apply,
                               float* loss_out) {
    if (!e) return NTF_EINVAL;
    const int64_t n = (int64_t)e->h_order.size();
    if (offset < 0 || B < 1 || offset + B > n || global_offset < 0 || global_offset + global_B > n || offset < global_offset ||
        offset + B > global_offset + global_B)
        FAIL(e, NTF_EINVAL, "step_staged: shard / batch outside the staged order");
    // the batch that follows in the staged order (an epoch walks it front to back, src/mdl/fnn.py:118): what the head prefetch works for
    const int64_t no = offset + B; const int nB = (int)std::min<int64_t>(B, n - no);
    if (train && apply && global_offset == offset && global_B == B && nB >= 1) { e->hp_next_rows = e->d_order + no; e->hp_next_B = nB; e->hp_next_host = e->h_order.data() + no; }
    else { e->hp_next_B = 0; e->hp_next_host = nullptr; }
    const int rc = step_common(e, e->d_order + offset, B, global_B, nullptr, loss_out, train != 0, apply != 0, true, e->h_order.data() + global_offset, global_B,
                               false, (uint32_t)(offset - global_offset));
    e->hp_next_B = 0; e->hp_next_host = nullptr;
    return rc;
}

extern "C" int ntf_step_staged_deferred(ntf_engine* e, int64_t offset, int32_t B, int64_t global_offset, int32_t global_B, float* loss_out) {
    if (!e) return NTF_EINVAL;
    const int64_t n = (int64_t)e->h_order.size();
    if (offset < 0 || B < 1 || offset + B > n || global_offset < 0 || global_offset + global_B > n || offset < global_offset ||
        offset + B > global_offset + global_B)
        FAIL(e, NTF_EINVAL, "step_staged_deferred: shard / batch outside the staged order");
    return step_common(e, e->d_order + offset, B, global_B, nullptr, loss_out, true, false, true, e->h_order.data() + global_offset, global_B, true,
                       (uint32_t)(offset - global_offset));
}
// ---- data-parallel pipelining of the head (see fwd_ranges): the ranges, and the deferred step with a callback in front of every range
extern "C" int ntf_fwd_ranges(ntf_engine* e, int32_t B, int32_t* n_ranges, int32_t* k0_k1 /* nullable: 2 x 4 ints, dW chunks [k0, k1) of each range */) {
    if (!e || !n_ranges) return NTF_EINVAL;
    FwdRange fr[4]; int tot = 0;
    const int n = (B >= 1 && B <= e->cfg.max_batch && e->dp_ranges) ? fwd_ranges(e, B, fr, &tot) : 0;
    *n_ranges = n;
    if (k0_k1) for (int j = 0; j < n; ++j) { k0_k1[2 * j] = fr[j].k0; k0_k1[2 * j + 1] = fr[j].k1; }
    return NTF_OK;
}
extern "C" int ntf_step_staged_deferred_cb(ntf_engine* e, int64_t offset, int32_t B, int64_t global_offset, int32_t global_B, float* loss_out,
                                           int (*before_range)(int32_t range, void* user), void* user) {
    if (!e) return NTF_EINVAL;
    const int64_t n = (int64_t)e->h_order.size();
    if (offset < 0 || B < 1 || offset + B > n || global_offset < 0 || global_offset + global_B > n || offset < global_offset ||
        offset + B > global_offset + global_B)
        FAIL(e, NTF_EINVAL, "step_staged_deferred_cb: shard / batch outside the staged order");
    e->cb_fn = before_range; e->cb_user = user;
    const int rc = step_common(e, e->d_order + offset, B, global_B, nullptr, loss_out, true, false, true, e->h_order.data() + global_offset, global_B, true,
                               (uint32_t)(offset - global_offset));
    e->cb_fn = nullptr; e->cb_user = nullptr;
    return rc;
}
static int dw_chunk_span(ntf_engine* e, int k, int64_t& off_w, int64_t& off_r, int64_t& cnt, int& wg_begin) {
    const LayerInfo& li = e->layers[e->L - 1];
    const int tile = fused_dw_tile(), M = li.out, total = (M + tile - 1) / tile, n = (total + 255) / 256;
    if (!fused_ok(e) || k < 0 || k >= n) FAIL(e, NTF_EINVAL, "dw_chunk: bad chunk index (or the fused output-layer path does not apply to this shape)");
    wg_begin = k * 256;
    const int64_t lo = (int64_t)wg_begin * tile * li.in, hi = std::min<int64_t>((int64_t)(wg_begin + 256) * tile, M) * li.in;
    off_w = li.off[NTF_P_WEIGHT] + lo; cnt = hi - lo;
    off_r = e->cfg.bayesian ? li.off[NTF_P_RHO_WEIGHT] + lo : -1;
    return NTF_OK;
}
extern "C" int ntf_dw_chunks(ntf_engine* e, int32_t* n_chunks) {
    if (!e || !n_chunks) return NTF_EINVAL;
    const int tile = fused_dw_tile(), total = (e->layers[e->L - 1].out + tile - 1) / tile;
    *n_chunks = fused_ok(e) ? (total + 255) / 256 : 0;
    return NTF_OK;
}
extern "C" int ntf_dw_chunk_range(ntf_engine* e, int32_t k, int64_t* off_weight, int64_t* off_rho, int64_t* count) {
    if (!e || !off_weight || !off_rho || !count) return NTF_EINVAL;
    int wg;
    return dw_chunk_span(e, k, *off_weight, *off_rho, *count, wg);
}
extern "C" int ntf_dw_chunk(ntf_engine* e, int32_t k) {
    if (!e) return NTF_EINVAL;
    if (!e->pend_valid) FAIL(e, NTF_ESTATE, "dw_chunk: no deferred dW kernel pending (call ntf_step_staged_deferred first)");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    int64_t ow, orr, cnt; int wg;
    int r = dw_chunk_span(e, k, ow, orr, cnt, wg);
    if (r) return r;
    const LayerInfo& li = e->layers[e->L - 1];
    FusedDw f = e->pend;
    f.wg_begin = wg; f.wg_count = 256;
    { Scope t(e, F_OUT_FUSED_DW); launch_fused_out_dw(e->st, f); }
    if (k == e->pend_chunks - 1) {
        if (e->cfg.bayesian) {
            Scope t(e, F_FLIPOUT_FINAL);
            launch_flipout_grad_finalize(e->st, e->P + li.off[NTF_P_BIAS], e->P + li.off[NTF_P_RHO_BIAS], e->G + li.off[NTF_P_BIAS],
                                         e->G + li.off[NTF_P_RHO_BIAS], li.out, e->pend_eps_b, e->pend_klw_b);
        }
        e->pend_valid = false;
        if ((r = join_side(e))) return r;       // the hidden layers' gradients (side stream, beside the chunks) are complete for whatever the caller queues next on this stream
    }
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) FAIL(e, NTF_EHIP, std::string("kernel launch: ") + hipGetErrorString(s));
    return NTF_OK;
}
// One train step of an expert-sharded engine in three phases (include/opentf_amd.h): the host sums d(hidden) over the shards between phases 1 and 3,
// while phase 2 (this shard's output-layer backward, the longest kernel after the forward) runs.
extern "C" int ntf_step_staged_ep(ntf_engine* e, int64_t offset, int32_t B, int32_t phase) {
    if (!e) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (!e->ep) FAIL(e, NTF_ESTATE, "step_staged_ep: the engine was not created as an expert shard (ntf_config.expert_lo / experts_global / ep_world)");
    int r;
    if (phase == 1) {
        const int64_t n = (int64_t)e->h_order.size();
        if (offset < 0 || B < 1 || offset + B > n) FAIL(e, NTF_EINVAL, "step_staged_ep: batch outside the staged order");
        if ((r = check_ready(e, true))) return r;
        StepCtx c; c.B = B; c.global_B = B; c.train = true; c.step = e->step++; c.part = 1;
        c.fuse_adam = e->cfg.fuse_adam != 0;   // no gradient exchange for the output layer: its Adam may always ride in / beside the dW kernel
        if (e->ep_open == 2 && e->ep_side) HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_join, 0));   // an abandoned step's side-stream kernel still orders before this one
        e->pend_valid = false; e->ep_open = 0;
        if ((r = stage_rows(e, e->d_order + offset, B, true, &c.rows_dev))) return r;
        if (e->cfg.nsd == NTF_NSD_UNIGRAM_B && e->cfg.ns > 0) {
            // (phase 3 of the previous step may have staged THIS batch's table beside its dW kernel, as step_common's head prefetch does)
            if (e->hp.valid && e->hp.ub >= 0 && e->hp.step == c.step && e->hp.rows == c.rows_dev && e->hp.B == B) c.ub = e->hp.ub;
            else { c.ub = (e->ub_slot ^= 1); if ((r = set_batch_unigram(e, e->h_order.data() + offset, B, c.ub, e->st))) return r; }
            e->ub_slot = c.ub;
        }
        if ((r = run_step(e, c, true))) return r;
        e->last_B = B; e->last_global_B = B; e->neg_step = c.step;
        // the batch that follows in the staged order: what phase 3's head prefetch works for
        { const int64_t no = offset + B; const int nB = (int)std::min<int64_t>(B, n - no);
          if (nB >= 1) { e->hp_next_rows = e->d_order + no; e->hp_next_B = nB; e->hp_next_host = e->h_order.data() + no; } else { e->hp_next_B = 0; e->hp_next_host = nullptr; } }
        e->ep_ctx = c; e->ep_open = 1;
        // phase 2 (this shard's dW kernel) on the side stream, beside the exchange and phase 3: one rank of 2 runs 1.684 -> 1.626 ms, of 4 1.605 -> 1.584, of 8
        // unchanged (its split-K dW launch is a single round: no idle tail to fill)
        e->ep_side = e->side_bwd >= 1 && e->L > 1 && e->cfg.fuse_adam != 2;
        if (e->ep_side) { if ((r = side_stream(e))) return r; HIPCHK(e, hipEventRecord(e->ev_fork, e->st)); }
    } else if (phase == 2 || phase == 3) {
        if (e->ep_open != phase - 1) FAIL(e, NTF_ESTATE, phase == 2 ? "step_staged_ep: phase 2 without a pending phase 1" : "step_staged_ep: phase 3 without a pending phase 2");
        StepCtx c = e->ep_ctx; c.part = phase;
        e->ep_open = phase == 2 ? 2 : 0;
        if (phase == 2 && e->ep_side) {
            // the output layer's backward goes to the side stream: the main stream is about to wait for the d(hidden) exchange and then runs phase 3,
            // neither of which this kernel needs or feeds - it runs beside both and is joined before Adam
            StreamRestore restore{e, e->st};
            HIPCHK(e, hipStreamWaitEvent(e->st3, e->ev_fork, 0));
            e->st = e->st3;
            if ((r = run_step(e, c, false))) return r;
            HIPCHK(e, hipEventRecord(e->ev_join, e->st3));
        } else if ((r = run_step(e, c, false))) return r;
        if (phase == 3) {
            e->hp_next_B = 0; e->hp_next_host = nullptr;
            if (e->ep_side) HIPCHK(e, hipStreamWaitEvent(e->st, e->ev_join, 0));
            if ((r = apply_adam(e))) return r;
        }
    } else FAIL(e, NTF_EINVAL, "step_staged_ep: phase must be 1, 2 or 3");
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) FAIL(e, NTF_EHIP, std::string("kernel launch: ") + hipGetErrorString(s));
    return NTF_OK;
}
// where phase 1 leaves this shard's partial d(hidden) and phase 3 expects the sum: [B, h[-1]] floats (null when there is no hidden layer)
extern "C" int ntf_dh_buffer(ntf_engine* e, void** dev_ptr, int64_t* n_floats) {
    if (!e || !dev_ptr || !n_floats) return NTF_EINVAL;
    *dev_ptr = e->L > 1 ? e->dAct[(e->L - 1) & 1] : nullptr;
    *n_floats = e->L > 1 ? (int64_t)e->cfg.max_batch * e->cfg.dims[e->L - 1] : 0;
    return NTF_OK;
}

extern "C" int ntf_param_segment(ntf_engine* e, int layer, int kind, int64_t* off, int64_t* count) {
    if (!e || !off || !count) return NTF_EINVAL;
    return param_span(e, layer, kind, *off, *count);
}

static int run_epoch(ntf_engine* e, const int64_t* order, int64_t n, int32_t B, float* mean_loss, bool train) {
    if (!e || !order || n < 1 || B < 1 || B > e->cfg.max_batch) { if (e) e->err = "epoch: bad arguments"; return NTF_EINVAL; }
    int r = ntf_stage_order(e, order, n);
    if (r) return r;
    HIPCHK(e, hipMemsetAsync(e->d_acc, 0, 8, e->st));
    HIPCHK(e, hipMemsetAsync(e->d_acc_steps, 0, 8, e->st));
    e->eval_chain = train ? 0 : 1; e->chain_valid = false;      // (evaluation: nothing but these steps touches the engine until the loop ends)
    for (int64_t o = 0; o < n; o += B) {
        const int b = (int)std::min<int64_t>(B, n - o);
        if ((r = ntf_step_staged(e, o, b, o, b, train, train, nullptr))) { e->eval_chain = 0; return r; }
    }
    e->eval_chain = 0; e->chain_valid = false;
    double sum = 0; int64_t steps = 0;
    if ((r = ntf_epoch_loss(e, &sum, &steps))) return r;
    if (mean_loss) *mean_loss = steps ? (float)(sum / (double)steps) : 0.f;
    return NTF_OK;
}
extern "C" int ntf_train_epoch(ntf_engine* e, const int64_t* order, int64_t n, int32_t B, float* mean_loss) { return run_epoch(e, order, n, B, mean_loss, true); }
extern "C" int ntf_eval_epoch(ntf_engine* e, const int64_t* order, int64_t n, int32_t B, float* mean_loss) { return run_epoch(e, order, n, B, mean_loss, false); }
extern "C" int ntf_epoch_loss(ntf_engine* e, double* sum, int64_t* steps) {
    if (!e) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    double s = 0; int64_t k = 0;
    if (int r = join_side(e)) return r;
    HIPCHK(e, hipMemcpyAsync(&s, e->d_acc, 8, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipMemcpyAsync(&k, e->d_acc_steps, 8, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipMemsetAsync(e->d_acc, 0, 8, e->st));
    HIPCHK(e, hipMemsetAsync(e->d_acc_steps, 0, 8, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    if (sum) *sum = s;
    if (steps) *steps = k;
    return NTF_OK;
}

// ------------------------------------------------------------------------------------------ inference
static int infer_pass(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, StepCtx& c) {
    int r;
    if ((r = check_ready(e, false))) return r;
    c.B = B; c.global_B = B; c.inj = inj; c.train = false; c.step = e->step++;
    if ((r = stage_rows(e, rows, B, false, &c.rows_dev))) return r;
    StepCtx ci = c;
    ntf_inject noneg;
    if (inj) { noneg = *inj; noneg.neg_idx = nullptr; ci.inj = &noneg; }
    if ((r = stage_all_inj(e, ci))) return r;
    if ((r = make_input(e, c))) return r;
    return forward_layers(e, c, true, false);  // logits (post leaky_relu) in dZout
}

static int infer_pass_fused(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, int pass, int passes, bool want_unc, bool logits);
static int range_raised(ntf_engine* e, bool& raised);

extern "C" int ntf_logits(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, float* logits_host) {
    if (!e || !logits_host) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    const int M = e->cfg.dims[e->L];
    if (e->pl_mu && fused_ok(e)) {
        // the kernel the inference ships (k_out_fwd_b6 in its probs mode), storing leaky_relu(z) instead of its sigmoid
        if (!e->Pbuf) DM(e, &e->Pbuf, (int64_t)e->cfg.max_batch * M);
        const uint64_t step0 = e->step;
        if (range_ptr(e)) HIPCHK(e, hipMemsetAsync(e->d_range, 0, 4, e->st));
        int r = infer_pass_fused(e, rows, B, inj, 0, 1, false, true); if (r) return r;
        bool raised; if ((r = range_raised(e, raised))) return r;
        if (!raised) {
            HIPCHK(e, hipMemcpyAsync(logits_host, e->Pbuf, (size_t)B * M * 4, hipMemcpyDeviceToHost, e->st));
            HIPCHK(e, hipStreamSynchronize(e->st));
            return NTF_OK;
        }
        e->step = step0;
    }
    StepCtx c; int r = infer_pass(e, rows, B, inj, c); if (r) return r;
    HIPCHK(e, hipMemcpyAsync(logits_host, e->dZout, (size_t)B * M * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return NTF_OK;
}

extern "C" int ntf_get_negatives(ntf_engine* e, int64_t* host, int64_t count) {
    if (!e || !host) return NTF_EINVAL;
    const int B = e->last_B, ns = e->cfg.ns;
    if (B < 1 || ns < 1 || e->cfg.nsd == NTF_NSD_NONE) FAIL(e, NTF_ESTATE, "negatives: no step with sampled negatives has run");
    if (count != (int64_t)B * ns) FAIL(e, NTF_EINVAL, "negatives: count != B * ns of the last step");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (e->st4) HIPCHK(e, hipStreamSynchronize(e->st4));
    HIPCHK(e, hipMemcpyAsync(host, e->d_neg_set[e->neg_step & 1], (size_t)count * 8, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return NTF_OK;
}

// d loss / d z of the output layer as the last backward left it (fused path: transposed dzT; generic path: [B, M]) -> host [B, M]
extern "C" int ntf_get_dlogits(ntf_engine* e, float* host, int64_t count) {
    if (!e || !host) return NTF_EINVAL;
    const int M = e->cfg.dims[e->L], B = e->last_B;
    if (B < 1) FAIL(e, NTF_ESTATE, "dlogits: no backward has run");
    if (count != (int64_t)B * M) FAIL(e, NTF_EINVAL, "dlogits: count != B * M of the last backward");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    const float* src = e->dZout;
    if (fused_ok(e)) {
        if (!e->Pbuf) DM(e, &e->Pbuf, (int64_t)e->cfg.max_batch * M);
        // fp16x3 step: dzT holds packed plane pairs of dz * scale - unless that step fell back to the f32 kernels
        float inv = 0.f;
        if (e->last_dz_packed_scale > 0.f) {
            int32_t raised = 0;
            HIPCHK(e, hipMemcpyAsync(&raised, e->d_range, 4, hipMemcpyDeviceToHost, e->st));
            HIPCHK(e, hipStreamSynchronize(e->st));
            if (!raised) inv = 1.f / e->last_dz_packed_scale;
        }
        launch_fused_probs_finish(e->st, B, e->layers[e->L - 1].in, M, e->fws, e->dZout, e->Pbuf, nullptr, 1.f, true, inv);
        src = e->Pbuf;
    }
    HIPCHK(e, hipMemcpyAsync(host, src, (size_t)B * M * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return NTF_OK;
}

// one MC pass of the inference through the fused bf16x6 forward kernel: probabilities accumulate in the transposed buffer dZout
static int infer_pass_fused(ntf_engine* e, const int64_t* rows, int32_t B, const ntf_inject* inj, int pass, int passes, bool want_unc, bool logits) {
    int r;
    if ((r = check_ready(e, false))) return r;
    StepCtx c; c.B = B; c.global_B = B; c.inj = inj; c.train = false; c.step = e->step++;
    if ((r = stage_rows(e, rows, B, false, &c.rows_dev))) return r;
    StepCtx ci = c;
    ntf_inject noneg;
    if (inj) { noneg = *inj; noneg.neg_idx = nullptr; ci.inj = &noneg; }
    if ((r = stage_all_inj(e, ci))) return r;
    if ((r = make_input(e, c))) return r;
    if ((r = forward_layers(e, c, true, true))) return r;   // hidden layers only
    const int M = e->cfg.dims[e->L];
    const LayerInfo& lo = e->layers[e->L - 1];
    FusedOut f;
    f.B = B; f.H = lo.in; f.M = M; f.bayes = e->cfg.bayesian; f.train = 0;
    f.h = e->act[e->L - 1];
    f.mu = e->P + lo.off[NTF_P_WEIGHT]; f.mu_b = e->P + lo.off[NTF_P_BIAS];
    f.tnw = e->cfg.tnw; f.tpw = e->cfg.tpw; f.inv_B = 1.f / (float)B;
    f.dzT = e->dZout; f.dh_slab = e->dh_slab; f.ws = e->fws;
    f.bf16x6 = 1; f.mu_pl = e->pl_mu; f.wp_pl = e->pl_wp;
    f.np = mfma_np(e); f.w_scale = kW16Scale; f.h_scale = kH16Scale; f.dz_scale = 1.f;
    if (e->cfg.bayesian) {
        e->pre_valid = false;   // this pass's operands overwrite any prefetched ones
        // (lean: no f32 copy of sigma * eps - the passes read the planes; a raised range flag sends the whole call to the exact-f32 path, which makes its own operands.  Round 6
        //  measured the producer of pass p + 1 beside the forward kernel of pass p, into a second plane buffer: -2 % on a call, +10 % on an evaluation step - removed,
        //  profiles/r6_eval_prefetch_ab.md)
        const bool nof32 = e->lean && !inj && mfma_np(e) == 2 && e->pl_wp && range_ptr(e);
        { Scope t(e, F_FLIPOUT_OPERAND);
          // (the planes of mu are the first pass's: the passes of one call run back to back on unchanged parameters, and the range flag they share is read behind the last)
          launch_flipout_perturb(e->st, e->P + lo.off[NTF_P_RHO_WEIGHT], nullptr, lo.nw(), normal_spec(e, c, e->L - 1, T_EPS_W), nof32 ? nullptr : e->Wp[e->L - 1], 0.0, e->d_kl,
                                 e->pl_wp, pass == 0 ? e->pl_mu : nullptr, f.mu, lo.in, mfma_np(e), kW16Scale, range_ptr(e));
          launch_flipout_perturb(e->st, e->P + lo.off[NTF_P_RHO_BIAS], nullptr, lo.out, normal_spec(e, c, e->L - 1, T_EPS_B), e->bp[e->L - 1], 0.0, e->d_kl); }
        f.wp = e->Wp[e->L - 1]; f.bp = e->bp[e->L - 1];
        f.s_in = sign_spec(e, c, e->L - 1, T_S_IN, lo.in); f.s_out = sign_spec(e, c, e->L - 1, T_S_OUT, lo.out);
        f.planes_ready = 1;
    }
    f.probs = 1; f.pacc = pass > 0; f.pscale = 1.0f / (float)passes; f.plogit = logits; f.rflag = range_ptr(e);
    { Scope t(e, F_OUT_FUSED_AUX); launch_fused_out_fwd(e->st, f, 1); }
    { Scope t(e, F_OUT_FUSED_FWD); launch_fused_out_fwd(e->st, f, 2); }
    { Scope t(e, F_INFER); launch_fused_probs_finish(e->st, B, lo.in, M, e->fws, e->dZout, e->Pbuf, want_unc ? e->ent_mc : nullptr, 1.0f / (float)passes, pass == passes - 1); }
    return NTF_OK;
}

// fp16x3 inference: did an operand of the passes just queued leave the fp16 window?  (one 4-byte read; the caller synchronises for its outputs anyway)
static int range_raised(ntf_engine* e, bool& raised) {
    raised = false;
    if (!range_ptr(e)) return NTF_OK;
    int32_t v = 0;
    HIPCHK(e, hipMemcpyAsync(&v, e->d_range, 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    raised = v != 0;
    if (raised) e->range_fallbacks_host += 1;
    return NTF_OK;
}

static int infer_probs(ntf_engine* e, const int64_t* rows, int32_t B, int32_t nmc, const ntf_inject* inj_per_mc, bool want_unc) {
    const int M = e->cfg.dims[e->L];
    if (!e->Pbuf) DM(e, &e->Pbuf, (int64_t)e->cfg.max_batch * M);
    const int passes = e->cfg.bayesian ? std::max(1, nmc) : 1;
    if (want_unc) HIPCHK(e, hipMemsetAsync(e->ent_mc, 0, (size_t)B * 4, e->st));
    if (e->pl_mu && fused_ok(e)) {     // split-product forward kernel (H = 128): no dense logits, MC mean accumulated on the fly
        const uint64_t step0 = e->step;
        if (range_ptr(e)) HIPCHK(e, hipMemsetAsync(e->d_range, 0, 4, e->st));
        for (int p = 0; p < passes; ++p) { int r = infer_pass_fused(e, rows, B, inj_per_mc ? &inj_per_mc[p] : nullptr, p, passes, want_unc, false); if (r) return r; }
        bool raised; int r = range_raised(e, raised); if (r) return r;
        if (!raised) {
            if (want_unc) { Scope t(e, F_INFER); launch_row_entropy(e->st, e->Pbuf, B, M, e->ent_mean); }
            return NTF_OK;
        }
        e->step = step0;   // an operand left the fp16 window: the same passes (same generator keys) again on the exact-f32 path below
        if (want_unc) HIPCHK(e, hipMemsetAsync(e->ent_mc, 0, (size_t)B * 4, e->st));
    }
    for (int p = 0; p < passes; ++p) {
        StepCtx c; int r = infer_pass(e, rows, B, inj_per_mc ? &inj_per_mc[p] : nullptr, c); if (r) return r;
        Scope t(e, F_INFER);
        launch_sigmoid_acc(e->st, e->dZout, B, M, 1.0f / (float)passes, p > 0, e->Pbuf, want_unc ? e->ent_mc : nullptr);
    }
    if (want_unc) { Scope t(e, F_INFER); launch_row_entropy(e->st, e->Pbuf, B, M, e->ent_mean); }
    return NTF_OK;
}

static int copy_unc(ntf_engine* e, int B, float* pred_unc, float* model_unc) {
    if (!pred_unc && !model_unc) return NTF_OK;
    std::vector<float> em(B), ec(B);
    HIPCHK(e, hipMemcpyAsync(em.data(), e->ent_mean, (size_t)B * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipMemcpyAsync(ec.data(), e->ent_mc, (size_t)B * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    for (int i = 0; i < B; ++i) { if (pred_unc) pred_unc[i] = em[i]; if (model_unc) model_unc[i] = em[i] - ec[i]; }
    return NTF_OK;
}

extern "C" int ntf_forward(ntf_engine* e, const int64_t* rows, int32_t B, int32_t nmc, const ntf_inject* inj_per_mc, float* probs_host,
                           float* pred_unc, float* model_unc) {
    if (!e || !probs_host) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    int r = infer_probs(e, rows, B, nmc, inj_per_mc, pred_unc || model_unc); if (r) return r;
    HIPCHK(e, hipMemcpyAsync(probs_host, e->Pbuf, (size_t)B * e->cfg.dims[e->L] * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return copy_unc(e, B, pred_unc, model_unc);
}

extern "C" int ntf_forward_topk(ntf_engine* e, const int64_t* rows, int32_t B, int32_t nmc, int32_t K, float* values_host,
                                int32_t* indices_host, float* pred_unc, float* model_unc) {
    if (!e || !values_host || !indices_host) return NTF_EINVAL;
    const int M = e->cfg.dims[e->L];
    if (K < 1 || K > M || K > 2048) FAIL(e, NTF_EINVAL, "topk: K must be in [1, min(M, 2048)]");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    int r = infer_probs(e, rows, B, nmc, nullptr, pred_unc || model_unc); if (r) return r;
    if (e->tk_cap < (int64_t)e->cfg.max_batch * K) { dfree(e->tk_vals); dfree(e->tk_idx); e->tk_cap = (int64_t)e->cfg.max_batch * K; DM(e, &e->tk_vals, e->tk_cap); DM(e, &e->tk_idx, e->tk_cap); }
    { Scope t(e, F_INFER); launch_topk_rows(e->st, e->Pbuf, B, M, K, e->tk_vals, e->tk_idx, nullptr); }
    HIPCHK(e, hipMemcpyAsync(values_host, e->tk_vals, (size_t)B * K * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipMemcpyAsync(indices_host, e->tk_idx, (size_t)B * K * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    return copy_unc(e, B, pred_unc, model_unc);
}

extern "C" int ntf_gather_meanpool(ntf_engine* e, const int64_t* rows, int64_t n, float* out_host) {
    if (!e || n < 1) return NTF_EINVAL;
    if (!e->table || !e->s_indptr) FAIL(e, NTF_ESTATE, "skill CSR / table not set");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (!rows && n > e->s_rows) FAIL(e, NTF_EINVAL, "gather: n exceeds the CSR rows");
    int64_t* drows = nullptr; float* dout = nullptr;
    if (rows) {
        for (int64_t i = 0; i < n; ++i) if (rows[i] < 0 || rows[i] >= e->s_rows) FAIL(e, NTF_EINVAL, "gather: row id out of range");
        DM(e, &drows, n); HIPCHK(e, hipMemcpy(drows, rows, n * 8, hipMemcpyHostToDevice));
    }
    int rc = dmalloc(e, &dout, n * e->table_d);
    if (rc) { dfree(drows); return rc; }
    { Scope t(e, F_GATHER); launch_gather_meanpool(e->st, e->s_indptr, e->s_indices, e->table, drows, n, e->table_d, 1, dout); }
    hipError_t s = hipStreamSynchronize(e->st);
    if (s == hipSuccess && out_host) s = hipMemcpy(out_host, dout, n * e->table_d * 4, hipMemcpyDeviceToHost);
    dfree(drows); dfree(dout);
    if (s != hipSuccess) FAIL(e, NTF_EHIP, std::string("gather: ") + hipGetErrorString(s));
    return NTF_OK;
}

// ------------------------------------------------------------------------------------------ views
extern "C" int ntf_grad_buffer(ntf_engine* e, void** dev_ptr, int64_t* n_floats) {
    if (!e || !dev_ptr || !n_floats) return NTF_EINVAL; *dev_ptr = e->G; *n_floats = e->n_params; return NTF_OK;
}
extern "C" int ntf_param_buffer(ntf_engine* e, void** dev_ptr, int64_t* n_floats) {
    if (!e || !dev_ptr || !n_floats) return NTF_EINVAL; *dev_ptr = e->P; *n_floats = e->n_params;
    e->pre_valid = false;   // whoever takes the raw view may write through it: the next step runs the stand-alone operand producer
    return NTF_OK;
}
extern "C" int ntf_params_touched(ntf_engine* e) {
    if (!e) return NTF_EINVAL;
    e->pre_valid = false;
    return NTF_OK;
}
extern "C" int ntf_moment_buffers(ntf_engine* e, void** dev_m1, void** dev_v2, int64_t* n_floats) {
    if (!e || !dev_m1 || !dev_v2 || !n_floats) return NTF_EINVAL;
    *dev_m1 = e->M1; *dev_v2 = e->V2; *n_floats = e->n_params;
    return NTF_OK;
}
extern "C" int ntf_synchronize(ntf_engine* e) {
    if (!e) return NTF_EINVAL;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (int r = join_side(e)) return r;
    HIPCHK(e, hipStreamSynchronize(e->st));
    return NTF_OK;
}
extern "C" int ntf_kernel_times(ntf_engine* e, int enable, const char** names, double* ms, int64_t* calls, int cap) {
    if (!e) return NTF_EINVAL;
    hipSetDevice(e->cfg.device);
    join_side(e);
    drain_times(e);
    int n = std::min(cap, (int)F_COUNT);
    for (int i = 0; i < n; ++i) { if (names) names[i] = kFamNames[i]; if (ms) ms[i] = e->fam_ms[i]; if (calls) calls[i] = e->fam_calls[i]; }
    for (int i = 0; i < F_COUNT; ++i) { e->fam_ms[i] = 0; e->fam_calls[i] = 0; }
    e->timing = enable < 0 ? 0 : (enable > 4 ? 1 : enable);
    return F_COUNT;
}

extern "C" int ntf_k_gemm_f32(void* stream, int m, int n, int k, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
                              int64_t sbn, float* C, int64_t ldc) {
    if (m < 1 || n < 1 || k < 1 || !A || !B || !C) return NTF_EINVAL;
    GemmArgs g; g.M = m; g.N = n; g.K = k; g.A = A; g.sam = sam; g.sak = sak; g.B = B; g.sbk = sbk; g.sbn = sbn; g.C = C; g.ldc = ldc;
    launch_gemm((hipStream_t)stream, g);
    return hipGetLastError() == hipSuccess ? NTF_OK : NTF_EHIP;
}

extern "C" int ntf_k_fill_normal(void* stream, uint64_t seed, uint64_t step, int layer, int64_t n, float* dev_out) {
    if (n < 1 || !dev_out) return NTF_EINVAL;
    ntf_engine tmp; tmp.seed = seed;
    NormalSpec s; make_key(&tmp, step, layer, T_EPS_W, s.k0, s.k1); s.tag = (uint32_t)(layer * 8 + T_EPS_W); s.step = (uint32_t)step;
    launch_fill_normal((hipStream_t)stream, s, n, dev_out);
    return hipGetLastError() == hipSuccess ? NTF_OK : NTF_EHIP;
}
extern "C" int ntf_k_fill_sign(void* stream, uint64_t seed, uint64_t step, int layer, int rows, int cols, float* dev_out) {
    if (rows < 1 || cols < 1 || !dev_out) return NTF_EINVAL;
    ntf_engine tmp; tmp.seed = seed;
    SignSpec s; s.enabled = 1; s.ld = cols; make_key(&tmp, step, layer, T_S_OUT, s.k0, s.k1);
    launch_fill_sign((hipStream_t)stream, s, rows, cols, dev_out);
    return hipGetLastError() == hipSuccess ? NTF_OK : NTF_EHIP;
}

// The device generators' own draws of one tensor of one step of THIS engine (its seed, its expert shard): what a native - non-injected - step with that step index
// consumes, in the reference's layouts (bayesian-torch LinearFlipout.forward: eps_weight [out, in], eps_bias [out], sign_input [rows, in], sign_output [rows, out];
// signs as +1 / -1, row r = position r of the step's minibatch).  Test hook: feeding these to the oracle ties the kernels that regenerate them in place (the forward
// kernels' sign words, the dW epilogue's Philox re-draw, the operand producers of the previous step's epilogue) to ONE exported tensor per (step, layer, kind).
extern "C" int ntf_get_noise(ntf_engine* e, uint64_t step, int32_t layer, int32_t kind, int32_t rows, float* host, int64_t count) {
    if (!e || !host) return NTF_EINVAL;
    if (!e->cfg.bayesian) FAIL(e, NTF_ESTATE, "noise: not a Bayesian model");
    if (layer < 0 || layer >= e->L || kind < 0 || kind > 3) FAIL(e, NTF_EINVAL, "noise: bad layer / kind (0 eps_weight, 1 eps_bias, 2 sign_input, 3 sign_output)");
    const LayerInfo& li = e->layers[layer];
    const bool sign = kind >= 2;
    if (sign && (rows < 1 || rows > e->cfg.max_batch)) FAIL(e, NTF_EINVAL, "noise: rows must be in [1, max_batch] for a sign tensor");
    const int64_t n = kind == 0 ? li.nw() : kind == 1 ? (int64_t)li.out : (int64_t)rows * (kind == 2 ? li.in : li.out);
    if (count != n) FAIL(e, NTF_EINVAL, "noise: element count mismatch");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    float* tmp = nullptr;
    DM(e, &tmp, n);
    StepCtx c; c.step = step; c.B = sign ? rows : 0; c.global_B = c.B;
    if (sign) { SignSpec s = sign_spec(e, c, layer, kind == 2 ? T_S_IN : T_S_OUT, kind == 2 ? li.in : li.out); s.inj = nullptr; launch_fill_sign(e->st, s, rows, kind == 2 ? li.in : li.out, tmp); }
    else { NormalSpec s = normal_spec(e, c, layer, kind == 0 ? T_EPS_W : T_EPS_B); s.inj = nullptr; launch_fill_normal(e->st, s, n, tmp); }
    hipError_t st = hipStreamSynchronize(e->st);
    int rc = NTF_OK;
    if (st == hipSuccess) {
        if (kind == 0 && stored_transposed(e, layer, NTF_P_WEIGHT)) {   // the producer walks the stored [S, H] order of a multi-hot first layer
            std::vector<float> t((size_t)n);
            st = hipMemcpy(t.data(), tmp, n * 4, hipMemcpyDeviceToHost);
            if (st == hipSuccess) transpose_host(t.data(), host, li.in, li.out);
        } else st = hipMemcpy(host, tmp, n * 4, hipMemcpyDeviceToHost);
    }
    hipFree(tmp);
    if (st != hipSuccess) { e->err = std::string("noise: ") + hipGetErrorString(st); rc = NTF_EHIP; }
    return rc;
}
