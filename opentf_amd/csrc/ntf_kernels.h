// Internal launcher interface between the engine (ntf_engine.hip) and the gfx950 kernels
// (ntf_kernels.hip, ntf_fused.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ntf {

constexpr float kLeakySlope = 0.01f;  // torch leaky_relu default, src/mdl/fnn.py:25

// +1/-1 sign tensor [rows, ld]: injected f32 array or a counter hash of (row, col) under a key.
struct SignSpec {
    const float* inj = nullptr;
    int64_t ld = 0;
    uint32_t k0 = 0, k1 = 0;
    int enabled = 0;
};

// normal(0,1) tensor: injected f32 array or Philox4x32-10 + Box-Muller under (key, element index).
struct NormalSpec {
    const float* inj = nullptr;
    uint32_t k0 = 0, k1 = 0, tag = 0, step = 0;
    int64_t qbase = 0;   // generator index of the tensor's first group of four: an expert shard of the output layer draws what the whole layer would draw
};

struct GemmArgs {
    int M = 0, N = 0, K = 0;
    const float* A = nullptr; int64_t sam = 0, sak = 0;  // A(m,k) = A[m*sam + k*sak]
    const float* B = nullptr; int64_t sbk = 0, sbn = 0;  // B(k,n) = B[k*sbk + n*sbn]
    SignSpec sa; int sa_t = 0;   // sign on A elements at (m,k), or (k,m) when sa_t
    SignSpec sb; int sb_t = 0;   // sign on B elements at (k,n), or (n,k) when sb_t
    float* C = nullptr; int64_t ldc = 0;        // value output / accumulate source
    float* Act = nullptr; int64_t ldact = 0;    // leaky_relu(value) output
    const float* bias = nullptr;                // [N]
    SignSpec sc;                                // sign on the output at (m,n)
    const float* mask = nullptr; int64_t ldmask = 0; // value *= (mask[m,n] > 0 ? 1 : slope)
    int accumulate = 0;                         // value += C[m,n]
    int ksplit = 1;                             // >1: K split over blockIdx.z into `slab` [ksplit, M, N], summed in slice order
    float* slab = nullptr;
    float alpha = 1.0f;
};

void launch_gemm(hipStream_t st, const GemmArgs& a);

void launch_gather_meanpool(hipStream_t st, const int64_t* indptr, const int32_t* indices, const float* table,
                            const int64_t* rows, int64_t n, int d, int mean, float* out);
void launch_gather_dense_rows(hipStream_t st, const float* X, int d, const int64_t* rows, int64_t n, float* out);
// multi-hot input: layer 0 as a gather-sum of W columns / scatter-add of its gradient over the skill CSR (no dense X)
void launch_multihot_fwd(hipStream_t st, const int64_t* rows, int B, int S, int H, const int64_t* indptr, const int32_t* indices,
                         const float* W, const float* b, const float* Wp /*nullable*/, const float* bp, SignSpec sin, SignSpec sout, float* act);
// touched (nullable): [S] bytes, set to 1 for every skill row the scatter adds to (the caller clears it first): launch_flipout_sweep then reads gradients of those rows only
void launch_multihot_bwd(hipStream_t st, const int64_t* rows, int B, int S, int H, const int64_t* indptr, const int32_t* indices,
                         const float* dZ, SignSpec sin, SignSpec sout, float* gW /*zeroed*/, float* gWp /*nullable, zeroed*/, uint8_t* touched = nullptr);

// Wp = softplus(rho) * eps  (eps generated or injected)
// also accumulates w * KL(N(mu, softplus(rho)^2) || N(0,1)) summed over the tensor into kl_out when mu != nullptr
// planes_w / planes_mu != null (H % 4 == 0, n = rows * H): also writes the bf16 split planes of out / of pmu (fused_planes_elems layout, rows past n/H untouched)
// evaluation steps of ONE ntf_eval_epoch call run back to back on unchanged parameters: the first producer launch of the chain also leaves the output layer's KL term
// (kl_out2) and whether the planes of mu left the fp16 window (mu_flag_out); the later ones skip both (mu = planes_mu = null), start the step's KL sum from the kept value
// (launch_step_scalars, start_from) and raise the step's range flag if raise_if says so
struct PerturbChain { double* kl_out2 = nullptr; int* mu_flag_out = nullptr; const int* raise_if = nullptr; };
void launch_flipout_perturb(hipStream_t st, const float* rho, const float* mu, int64_t n, NormalSpec eps, float* out, double w, double* kl_out,
                            uint16_t* planes_w = nullptr, uint16_t* planes_mu = nullptr, const float* pmu = nullptr, int H = 0,
                            int np = 3 /*3: bf16 three-way planes, 2: fp16 two-way planes of value * pscale*/, float pscale = 1.f,
                            int* rflag = nullptr /*np = 2: raised when a plane operand leaves the fp16 window*/,
                            const int* only_if = nullptr /*device flag: the launch does nothing unless it is non-zero (the f32 sigma * eps of a step on prefetched operands that falls back to the exact-f32 kernels)*/,
                            PerturbChain ch = PerturbChain());
// g_rho = gWp * eps * sigmoid(rho) + kl' ; g_mu += kl'   (KL of N(mu, sigma^2) against N(0,1), mean over n, times klw)
void launch_flipout_grad_finalize(hipStream_t st, const float* mu, const float* rho, float* g_mu, float* g_rho /*in: gWp*/,
                                  int64_t n, NormalSpec eps, float klw);
// sum over the layer of the elementwise KL -> adds mean (times w) into out[0] (double)
void launch_kl_value(hipStream_t st, const float* mu, const float* rho, int64_t n, double w, double* out);

// dense pass of the output-layer loss: every label treated as an un-sampled negative (y=0, weight tnw).
// dZ[i,c] = tnw * sigmoid(l) * lrelu'(z) * inv_B ; partial[i, chunk] = sum of tnw * softplus(l)
void launch_loss_dense(hipStream_t st, const float* Z, int64_t ld, int B, int M, float tnw, float inv_B,
                       float* dZ, float* partial, int nchunk);
int loss_dense_nchunk(int M);
// sparse fix-up for positives (member CSR row) and selected negatives: rewrites dZ there and returns the
// loss correction per row.  neg may be null (nsd None).  write_dz = 0 for eval.
void launch_loss_special(hipStream_t st, const float* Z, int64_t ld, int B, int M, const int64_t* rows,
                         const int64_t* m_indptr, const int32_t* m_indices, const int64_t* neg, int ns,
                         float tpw, float tnw, float inv_B, float* dZ, float* row_fix);
// loss = (sum_i (sum_chunk partial[i,:] + row_fix[i])) * inv_B + kl[0]*kl_scale ; out[0] = loss, acc[0] += loss
void launch_loss_finalize(hipStream_t st, const float* partial, int nchunk, const float* row_fix, int B, float inv_B,
                          const double* kl, double kl_scale, float* out, double* acc, int64_t* acc_steps);

// bias gradients: g_b[n] = sum_b dZ[b,n];  g_rho_b[n] = (sum_b dZ[b,n]*s_out(b,n)) * eps_b[n] * sigmoid(rho_b[n])
void launch_bias_grad(hipStream_t st, const float* dZ, int64_t ld, int B, int N, SignSpec sout, float* g_b, float* g_pert /*nullable*/, float* scratch = nullptr, int64_t scratch_floats = 0);

// uniform negatives: ns distinct columns per row among the row's non-members (src/mdl/fnn.py:48-56)
void launch_ns_uniform(hipStream_t st, const int64_t* rows, int B, int M, int ns, const int64_t* m_indptr,
                       const int32_t* m_indices, uint32_t k0, uint32_t k1, uint32_t step, uint32_t row0, int64_t* out);
// weighted negatives by alias table (src/mdl/fnn.py:58-72), fallback to uniform over all columns when the
// row's negatives have zero total weight
void launch_ns_alias(hipStream_t st, const int64_t* rows, int B, int M, int ns, const int64_t* m_indptr,
                     const int32_t* m_indices, const float* prob, const int32_t* alias, const double* weight,
                     double total_weight, uint32_t k0, uint32_t k1, uint32_t step, uint32_t row0, int64_t* out);

// the same over a sparse support (unigram_b: the experts of the current batch): cols sorted ascending, tables indexed by support slot
void launch_ns_alias_sparse(hipStream_t st, const int64_t* rows, int B, int M, int ns, const int64_t* m_indptr, const int32_t* m_indices,
                            const int32_t* cols, const float* prob, const int32_t* alias, const float* weight, int nsup, double total_weight,
                            uint32_t k0, uint32_t k1, uint32_t step, uint32_t row0, int64_t* out);

void launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                 float eps, float bc1, float bc2_sqrt);
void launch_step_scalars(hipStream_t st, double* kl, int take_next, const double* start_from = nullptr);
// the conditional f32 copy of the next step's sigma * eps as extra workgroups of the ticketed Adam launch (k_adam_ranges, f32c): out[0, n) = softplus(rho) * eps iff *only_if != 0
struct F32CopyJob { const float* rho; float* out; int64_t n; NormalSpec eps; const int* only_if; };
void launch_adam_ranges(hipStream_t st, float* P, float* G, float* M1, float* V2, const int64_t* lo_hi, int n, float lr, float b1, float b2,
                        float eps, float bc1, float bc2_sqrt, const int* fin = nullptr, const NormalSpec* fin_eps = nullptr, float fin_klw = 0.f,
                        double* rotate = nullptr,    // up to four ranges per launch; fin / rotate: see k_adam_ranges
                        float* nx_bp = nullptr, const NormalSpec* nx_eps = nullptr, double nx_klw = 0.0,   // with rotate: also the next step's output-bias operand + its KL (k_adam_ranges, nx)
                        const F32CopyJob* f32c = nullptr);
// One sweep over a Flipout weight tensor (mu, rho) of a hidden layer in a step that applies Adam (the multi-hot first layer of BASELINE config 3: 90 671 x 128 pairs):
// Flipout chain rule of the rho gradient + KL gradients (k_flipout_grad_finalize) -> Adam on mu and rho in place (k_adam) -> the NEXT step's operand sigma' eps'
// and KL(mu', rho') * nx_klw added to *nx_kl (k_flipout_perturb) - the output layer's dW epilogue arithmetic for a layer whose gradient is a scatter.  The gradients
// are CONSUMED: read and overwritten with zeros (the buffer is then ready for the next step's scatter, no memset); with `touched` (row flags of launch_multihot_bwd,
// row = element / H) only the flagged rows are read and cleared - the others hold zeros by that invariant.  n and H multiples of 4.  52-60 B of HBM traffic per pair.
struct FlipoutSweep {
    float *mu, *rho, *g_mu, *g_rho, *m_mu, *v_mu, *m_rho, *v_rho; int64_t n;
    NormalSpec eps; float klw;
    float lr, b1, b2, adam_eps, bc1, bc2_sqrt;
    NormalSpec nx_eps; float* nx_wp; double nx_klw; double* nx_kl;
    const uint8_t* touched; int H;
};
void launch_flipout_sweep(hipStream_t st, const FlipoutSweep& a);
void launch_fill(hipStream_t st, float* p, int64_t n, float v);
// probabilities: out = (accumulate ? out : 0) + sigmoid(leaky(Z)) * scale ; ent_acc[i] += sum_c -p log(p+1e-15)
void launch_sigmoid_acc(hipStream_t st, const float* Act, int64_t n_rows, int M, float scale, int accumulate, float* out,
                        float* ent_rows /*nullable: per-row entropy of THIS pass*/);
void launch_row_entropy(hipStream_t st, const float* P, int n_rows, int M, float* ent);
void launch_topk_rows(hipStream_t st, const float* P, int n_rows, int M, int K, float* vals, int32_t* idx, void* workspace);
size_t topk_workspace_bytes(int n_rows, int M, int K);

// test hooks for the device generators
void launch_fill_normal(hipStream_t st, NormalSpec s, int64_t n, float* out);
void launch_fill_sign(hipStream_t st, SignSpec s, int rows, int cols, float* out);

}  // namespace ntf
