// Fused output-layer kernels (the hot spot: [B,H] x [H,M] with M up to millions of experts).
#pragma once
#include "ntf_kernels.h"

namespace ntf {

// forward + sparse-label loss + d(logits) + d(hidden) of the output layer in one pass over the weights
struct FusedOut {
    int B = 0, H = 0, M = 0, bayes = 0, train = 0;
    const float* h = nullptr;                       // [B, H] leaky_relu output of the last hidden layer
    const float *mu = nullptr, *mu_b = nullptr;     // [M, H], [M]
    const float *wp = nullptr, *bp = nullptr;       // flipout perturbation operands sigma*eps [M, H], [M] (wp: read by the exact-f32 / bf16x6 kernels only; the fp16x3 step reads wp_pl)
    SignSpec s_in, s_out;
    float tnw = 1.f, tpw = 1.f, inv_B = 1.f;
    float* dzT = nullptr;                           // out: d loss / d z, TRANSPOSED [M, ldb], ldb = fused_ldb(B)
    float* dh_slab = nullptr;                       // scratch: per column-group partial d(hidden)
    float* dh = nullptr;                            // out: [B, H] d loss / d(pre-activation of the hidden layer), may be null
    const float* h_mask = nullptr;                  // leaky_relu' mask source for dh (= h), null: no mask
    float* loss_partial = nullptr;                  // [B, fused_loss_slots(M)]
    void* ws = nullptr;                             // fused_workspace_bytes(B, H, M)
    // sparse fix-up
    const int64_t* rows = nullptr; const int64_t* m_indptr = nullptr; const int32_t* m_indices = nullptr;
    const int64_t* neg = nullptr; int ns = 0; float* row_fix = nullptr;
    int c_lo = 0;                                   // expert shard of the output layer: m_indices / neg hold GLOBAL expert ids, mu .. are rows [c_lo, c_lo + M)
    // bf16x6 arithmetic (H = 128): scratch for the bf16 split planes of mu / Wp, fused_planes_elems(M, H) uint16 each
    int bf16x6 = 0; uint16_t* mu_pl = nullptr; uint16_t* wp_pl = nullptr;
    int np = 2;                                     // 2: fp16x3 (operands scaled by exact powers of two before their two-way fp16 split); 3 (bf16x6) is no longer instantiated
    float w_scale = 1.f, h_scale = 1.f, dz_scale = 1.f;
    int wide = 5;                                   // np = 2 training step: 5 (default) k_out_fwd_h3p - a logit wave and a gradient wave per 32 rows, two waves per SIMD, 32-expert steps;
                                                    // 0 the 32-expert-tile kernel k_out_fwd_b6 (NTF_FWD_KERNEL=0: A/B runs).  Round 3's one-wave form (3) and sixteen-row-wave form (4) are retired
    int eval_kernel = 1;                            // np = 2 evaluation loss: 1 (default, round 6) k_out_fwd_h3e - eight logit waves on 256 rows, the two waves of a SIMD half a step apart; 0 k_out_fwd_b6
    int ncg_limit = 0;                              // > 0 (diagnostics, NTF_COSCHED): at most this many column groups, i.e. a forward grid of NRB * ncg_limit workgroups that leaves CUs free
    int split_fallback = 0;                         // the exact-f32 forward launch behind the split-product kernel is NOT part of phase 2 but a phase of its own (8)
    // the split-product forward over a RANGE of the experts (k_out_fwd_h3p only; data-parallel ranks launch one range per all-gathered parameter chunk): 64-expert tiles
    // [chunk_t_lo, chunk_t_hi) on chunk_ncg column groups, whose dh slabs / loss partials are chunk_cg_off .. of chunk_ncg_tot in all.  chunk_ncg_tot > 0 with chunk_ncg = 0:
    // not a launch of a range but the phases behind them (4: the fix-up sums chunk_ncg_tot partials; 8 with split_fallback: the whole-layer exact-f32 launch)
    int chunk_t_lo = 0, chunk_t_hi = 0, chunk_cg_off = 0, chunk_ncg = 0, chunk_ncg_tot = 0;
    int planes_ready = 0;
    int h_ready = 0;                                // the zero-padded h, h * s_in and the s_in words are in the workspace already (ntf_head.hip): phase 1 skips k_prep_h
    // inference (train = 0, probs = 1): dzT[c][i] (+)= sigmoid(leaky_relu(z)) * pscale instead of the loss; the row entropy partials go to the workspace
    int probs = 0, pacc = 0; float pscale = 1.f;
    int plogit = 0;                                 // probs pass that stores leaky_relu(z) instead of its sigmoid (ntf_logits)
    // fp16x3 (np = 2): operands are range-checked where they are split; a raised rflag[0] turns the split-product kernels of this step into
    // no-ops and lets the exact-f32 kernels launched right behind them run instead (rflag[1] counts such steps).  Null: no check.
    int* rflag = nullptr;
};

// weight / bias gradients of the output layer from dzT (K = batch); for Flipout the rho gradient is finalised here
// (times eps * sigmoid(rho), plus the KL terms), eps being recovered as wp / softplus(rho).
struct FusedDw {
    int B = 0, H = 0, M = 0, bayes = 0;
    const float* dzT = nullptr; const float* h = nullptr;
    const float *mu = nullptr, *rho = nullptr, *wp = nullptr;
    float *g_mu = nullptr, *g_rho = nullptr, *g_b = nullptr, *g_bp = nullptr;
    float klw = 0.f;
    void* ws = nullptr;                             // the workspace launch_fused_out_fwd filled for this step
    SignSpec s_out; int s_out_inj = 0;              // s_out keys; s_out_inj: signs were injected this step (packed image in ws)
    // adam != 0 (single GPU): Adam on mu / rho runs in the epilogue (in place), g_mu / g_rho are not written
    int adam = 0;
    int bf16x6 = 0;                                 // 1: split-product MFMAs (f32-accurate, see ntf_fused.hip) instead of the f32 MFMA
    int np = 2;                                     // 2: fp16 two-way split of scaled operands, three products (3, the bf16 three-way split, is no longer instantiated)
    float a_scale = 1.f, h_scale = 1.f;             // np = 2: exact power-of-two scales of dz (applied in the kernel) and of the h planes (applied by the producer)
    int wg_begin = 0, wg_count = 0;                 // wg_count > 0: launch only the expert tiles [wg_begin, wg_begin + wg_count) of fused_dw_tile() experts each
    int* rflag = nullptr;                           // fp16x3 range guard, see FusedOut
    int dz_packed = 0;                              // np = 2, H = 128: dzT holds the forward kernel's packed fp16 plane pairs (see pack_planes)
    int ksplit = 1; float* part = nullptr;          // dz_packed path: split every launched expert tile's K (batch) range over ksplit workgroups; part = scratch
                                                    // of fused_dw_part_floats(M, H, ksplit) floats (k_out_dw_q<.., SPLIT> + k_out_dw_finish).  For few expert tiles (a narrow expert shard under a wide minibatch).
    int no_fallback = 0, fallback_only = 0;         // a step whose dW is issued as several launches (the tail split): the split-product launches carry no exact-f32 launch behind
                                                    // them (no_fallback), ONE launch over the whole layer follows (fallback_only: nothing but that kernel)
    float *w_mu = nullptr, *w_rho = nullptr, *m_mu = nullptr, *v_mu = nullptr, *m_rho = nullptr, *v_rho = nullptr;
    float lr_over_bc1 = 0.f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, bc2_sqrt = 1.f;
    // produce != 0 (adam, bayes, H = 128, fp16x3 planes): the Adam epilogue also writes the NEXT step's operands from the updated parameters - eps' (nx_eps: the
    // generator of step + 1), Wp' (f32, nx_wp), the split planes of Wp' and mu' (nx_pl_*, scale nx_pscale), KL' * nx_klw added to *nx_kl, *nx_rflag raised when an
    // operand leaves the fp16 window - what k_flipout_perturb would do in its own pass at the head of that step
    NormalSpec cur_eps;                             // Flipout: THIS step's eps generator - the rho gradient's eps is drawn again (or read from the injected tensor), never recovered as wp / sigma
    int lean = 0;                                   // with produce: do not write the f32 copy of the next step's sigma * eps (nx_wp) - every reader of that step takes the planes, and a step that
                                                    // falls back to the exact-f32 kernels makes the copy itself (launch_flipout_perturb(only_if)): 56 instead of 64 B of HBM traffic per mu / rho pair
    int produce = 0; NormalSpec nx_eps; float* nx_wp = nullptr; uint16_t *nx_pl_wp = nullptr, *nx_pl_mu = nullptr; float nx_pscale = 1.f; double nx_klw = 0.0;
    double* nx_kl = nullptr; int* nx_rflag = nullptr;
};

bool fused_supported(int H);
int fused_loss_slots(int M);
int64_t fused_dh_slab_floats(int B, int H, int M);
size_t fused_workspace_bytes(int B, int H, int M);
// where phase 1 of launch_fused_out_fwd and launch_fused_prep_planes leave their images in the workspace (the one-kernel head, ntf_head.hip, writes the same ones)
struct FusedWsPtrs { float *hz, *hs; uint32_t* sinbits; uint16_t* hb; uint32_t* sinT; int Bpad; };
FusedWsPtrs fused_ws_ptrs(void* ws, int B, int H, int M);
int fused_ldb(int B);
int64_t fused_planes_elems(int M, int H);   // uint16 elements of one matrix's split planes
int64_t fused_dw_part_floats(int M, int H, int ksplit);
int fused_dw_tile();   // experts per workgroup of the dW kernel (dzT rows are padded to a multiple of it)
// phases: 1 = operand preparation (zero-padded h, h*s_in, sign images), 2 = the fused MFMA kernel, 4 = sparse fix-up + dh reduction, 8 = (split_fallback) the exact-f32 forward launch of a range fallback
void launch_fused_out_fwd(hipStream_t st, const FusedOut& f, int phases = 7);
// s_out != null: also the transposed s_out sign words the packed fp16x3 dW kernel reads (k_sign_words_T)
void launch_fused_prep_planes(hipStream_t st, int B, int H, int M, int bayes, void* ws, int np = 2, float h_scale = 1.f, const SignSpec* s_out = nullptr, int s_out_inj = 0, int which = 3);
void launch_fused_out_dw(hipStream_t st, const FusedDw& f);
// after a probs pass: ent_rows[i] += scale * the pass's entropy terms (nullable); transpose: P [B, M] = PT^T
// unpack_inv_scale > 0: PT holds packed fp16 plane pairs (the fp16x3 step's dzT): P = (hi + lo) * unpack_inv_scale
void launch_fused_probs_finish(hipStream_t st, int B, int H, int M, void* ws, const float* PT, float* P, float* ent_rows, float scale, bool transpose,
                               float unpack_inv_scale = 0.f);

}  // namespace ntf
