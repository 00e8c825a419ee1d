// One-kernel head of a fused step (ntf_head.hip): gather -> hidden Flipout layer -> the output layer's operand images.  Internal, not part of the C ABI.
#pragma once
#include "ntf_kernels.h"

namespace ntf {

struct HeadArgs {
    int B = 0, Bpad = 0, D = 0, mode = 1 /*0: dense rows of Xall, 1: CSR mean pool of the skill table*/, bayes = 0, nrb = 0;
    const int64_t* rows = nullptr;
    const int64_t* s_indptr = nullptr; const int32_t* s_indices = nullptr; const float* table = nullptr; const float* Xall = nullptr;
    // hidden layer (layer 0): [128, D] weights, Flipout generators and sign keys of this step, KL weights (share / element count)
    const float *mu0 = nullptr, *b0 = nullptr, *rho0 = nullptr, *rhob0 = nullptr;
    NormalSpec eps_w0, eps_b0; SignSpec si0, so0;
    double klw_w0 = 0.0, klw_b0 = 0.0; double* kl = nullptr;
    // outputs: act[0] (X), act[1] (h), and what k_prep_h / k_prep_planes_T leave in the fused workspace
    float *X = nullptr, *act1 = nullptr, *hz = nullptr, *hs = nullptr; uint32_t* sinbits = nullptr; uint16_t* hb = nullptr;   // hb null: no planes (no dW kernel follows)
    uint32_t* sinT = nullptr;           // with hb: the s_in words of this K block in k_out_dw_q's bit order (k_sin_words_T)
    SignSpec si1;                       // the OUTPUT layer's s_in keys (h * s_in, s_in words)
    float h_scale = 1.f, h_limit = 0.f; int* rflag = nullptr;
    // operand producer of the output layer's bias (extra workgroups): bp1 = softplus(rho_b1) eps_b1, KL * klw_b1
    int64_t M = 0; const float *rho_b1 = nullptr, *mu_b1 = nullptr; NormalSpec eps_b1; float* bp1 = nullptr; double klw_b1 = 0.0;
};

bool head_supported(int D, int H);
void launch_head(hipStream_t st, const HeadArgs& a);

}  // namespace ntf
