// The head of a fused step as ONE kernel (round 3).  Between two steps' big kernels the engine used to queue a chain of small dependent launches -
// gather (src/mdl/emb/gnn.py:485 / src/mdl/ntf.py:22-24) -> two Flipout operand producers of the hidden layer -> two GEMMs (src/mdl/fnn.py:25) ->
// zero-padded h, h * s_in, s_in words -> the dW kernel's h planes - each a few microseconds of work behind a launch boundary, ~0.1 ms in all on the path
// to the forward kernel (profiles/r3_step_timeline.md).  For the shape every BASELINE configuration with a skill table has (ONE hidden layer of 128 units over a
// dense / mean-pooled input of d in {64, 128, 256}) this kernel does all of it: a workgroup owns 32 batch rows,
//   1. gathers them (CSR mean pool in CSR order, bit for bit k_gather_pool; or dense rows) into LDS and into act[0] (the hidden layer's backward reads it),
//   2. multiplies by mu0 (and, Flipout, (x * s_in) by sigma0 * eps0 generated in registers: Philox + Box-Muller, the arithmetic of k_flipout_perturb) on the
//      exact-f32 MFMA, wave w = hidden units 32 w .. 32 w + 31, lane half h = the k range [h d/2, (h+1) d/2) so that a lane streams its weight row by float4,
//   3. bias, s_out, leaky_relu -> act[1], the zero-padded copy, h * s_in, the s_in words and the fp16x3 range check (k_prep_h), and
//   4. the K-block-tiled fp16 split planes of h / h * s_in for the dW kernel (k_prep_planes_T).
// Workgroup 0 adds the hidden layer's KL; extra workgroups behind the row blocks are the operand producer of the output layer's BIAS (rho_b -> sigma eps, KL).
#include "ntf_head.h"
#include "ntf_device.h"
#include <algorithm>

namespace ntf {

constexpr int HH = 128;          // hidden width
constexpr int HS_LD = HH + 1;    // LDS row stride of the h tile (column reads of step 4 conflict-free)

template <bool BAYES, int D>
__global__ __launch_bounds__(256) void k_head(HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, half = lane >> 5;
    __shared__ double red[8];
    if ((int)blockIdx.x >= p.nrb) {
        // ---- operand producer of the output layer's bias: bp = softplus(rho_b) * eps_b, KL(mu_b, rho_b) * w (k_flipout_perturb without planes)
        if (!BAYES) return;
        const int64_t n = p.M, quads = (n + 3) / 4;
        const int nb = (int)gridDim.x - p.nrb;
        float kl = 0.f;
        for (int64_t q = (int64_t)((int)blockIdx.x - p.nrb) * 256 + tid; q < quads; q += (int64_t)nb * 256) {
            const int64_t e0 = q * 4;
            float z[4];
            normal4(p.eps_b1, q, e0, n, z);
            if (e0 + 3 < n) {
                const float4 r4 = *reinterpret_cast<const float4*>(p.rho_b1 + e0), m4 = *reinterpret_cast<const float4*>(p.mu_b1 + e0);
                const float rv[4] = {r4.x, r4.y, r4.z, r4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w};
                float ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float ls;
                    const float sigma = softplus_rho_fast(rv[j], ls);
                    ov[j] = sigma * z[j];
                    kl += -ls + 0.5f * (sigma * sigma + mv[j] * mv[j]) - 0.5f;
                }
                *reinterpret_cast<float4*>(p.bp1 + e0) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            } else {
                for (int j = 0; j < 4 && e0 + j < n; ++j) {
                    const float sigma = softplus_rho(p.rho_b1[e0 + j]);
                    p.bp1[e0 + j] = sigma * z[j];
                    const float m = p.mu_b1[e0 + j]; kl += -logf(sigma) + 0.5f * (sigma * sigma + m * m) - 0.5f;
                }
            }
        }
        const double s = wave_reduce_sum_d((double)kl);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0 && p.kl) atomicAdd(p.kl, (red[0] + red[1] + red[2] + red[3]) * p.klw_b1);
        return;
    }

    constexpr int XLD = D + 1;                            // LDS row stride of the x tile: a half-wave reads one column of 32 rows conflict-free
    float* Xs = reinterpret_cast<float*>(smem);           // [32][D + 1]
    float* Hs = Xs + 32 * XLD;                            // [32][HS_LD]
    uint32_t* SW = reinterpret_cast<uint32_t*>(Hs + 32 * HS_LD);   // [32][4] s_in words of the output layer
    const int i0 = (int)blockIdx.x * 32;

    // ---- 2a (d <= 128: before the gather, whose dependent load chain rows -> indptr -> indices -> table it then overlaps): this lane's weights - hidden unit j, the k half -
    //         with sigma0 * eps0 made in registers (Philox + Box-Muller, k_flipout_perturb's arithmetic) and the layer's KL terms
    const int j = wave * 32 + il;
    const int kb = half * (D / 2);
    constexpr bool PRE = D <= 128;
    constexpr int NQ = PRE ? D / 8 : 1;
    float mvA[NQ][4], wvA[NQ][4];
    float klw = 0.f;
    const float* mrow = p.mu0 + (int64_t)j * D + kb;
    const float* rrow = BAYES ? p.rho0 + (int64_t)j * D + kb : nullptr;
    auto weights_of = [&](int t4, float (&mv)[4], float (&wv)[4]) {
        const float4 m4 = *reinterpret_cast<const float4*>(mrow + 4 * t4);
        mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
        wv[0] = wv[1] = wv[2] = wv[3] = 0.f;
        if (BAYES) {
            const float4 r4 = *reinterpret_cast<const float4*>(rrow + 4 * t4);
            const float rv[4] = {r4.x, r4.y, r4.z, r4.w};
            float z[4];
            const int64_t e0 = (int64_t)j * D + kb + 4 * t4;
            normal4(p.eps_w0, e0 >> 2, e0, INT64_MAX, z);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float ls;
                const float sigma = softplus_rho_fast(rv[e], ls);
                wv[e] = sigma * z[e];
                klw += -ls + 0.5f * (sigma * sigma + mv[e] * mv[e]) - 0.5f;
            }
        }
    };
    if (PRE) {
#pragma unroll
        for (int t4 = 0; t4 < NQ; ++t4) weights_of(t4, mvA[t4], wvA[t4]);
    }

    // ---- 1. the 32 input rows -> Xs, act[0]
    {
        constexpr int G = D / 4, TPW = 64 / G;                              // G lanes per row, float4 each (D in {64, 128, 256})
        const int sub = lane / G, sl = lane % G;
#pragma unroll
        for (int it = 0; it < 8 / TPW; ++it) {       // (unrolled: the passes' dependent load chains rows -> indptr -> indices -> table run side by side)
            const int rl = wave * 8 + it * TPW + sub, t = i0 + rl;          // local / batch row
            const bool live = t < p.B;
            const int64_t team = live ? p.rows[t] : 0;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c0 = sl * 4;
            if (p.mode == 0) { if (live) acc = *reinterpret_cast<const float4*>(p.Xall + team * D + c0); }
            else {
                const int64_t beg = live ? p.s_indptr[team] : 0, end = live ? p.s_indptr[team + 1] : 0;
                const int nnz = (int)(end - beg);
                for (int base = 0; base < nnz; base += G) {            // (as k_gather_pool: indices read coalesced by the sub-group, <= 8 table rows in flight, CSR-ordered sum)
                    const int mine = base + sl;
                    const int my_idx = (mine < nnz) ? p.s_indices[beg + mine] : 0;
                    const int cnt = min(G, nnz - base);
                    for (int j0 = 0; j0 < cnt; j0 += 8) {
                        float4 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int j = j0 + u;
                            const int s = __shfl(my_idx, sub * G + (j < cnt ? j : 0), 64);
                            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (j < cnt) v[u] = *reinterpret_cast<const float4*>(p.table + (int64_t)s * D + c0);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (j0 + u < cnt) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                    }
                }
                if (live) { const float c = (float)nnz; acc.x /= c; acc.y /= c; acc.z /= c; acc.w /= c; }
            }
            if (live) *reinterpret_cast<float4*>(p.X + (int64_t)t * D + c0) = acc;
            float* xr = Xs + rl * XLD + c0;
            xr[0] = acc.x; xr[1] = acc.y; xr[2] = acc.z; xr[3] = acc.w;    // rows past B: zeros
        }
    }
    __syncthreads();

    // ---- 2. z = x mu0^T (+ (x * s_in) (sigma0 * eps0)^T): wave = 32 hidden units, lane = (hidden unit j, k half)
    f32x16 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }
    uint32_t sw0 = 0u;                           // s_in word of row i0 + il for the 32 columns being multiplied (kb is a multiple of 32)
    const float* xrow = Xs + il * XLD + kb;
    auto mma4 = [&](int t4, const float (&mv)[4], const float (&wv)[4]) {
        if (BAYES && (t4 & 7) == 0) sw0 = sign_word(p.si0.k0, p.si0.k1, (uint32_t)(i0 + il), (uint32_t)((kb >> 5) + (t4 >> 3)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * t4 + e;                          // column kb + k of row i0 + il
            const float a = xrow[k];
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, mv[e], acc1, 0, 0, 0);
            if (BAYES) {
                const uint32_t bit = (sw0 >> (k & 31)) & 1u;
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(__float_as_uint(a) ^ (bit << 31)), wv[e], acc2, 0, 0, 0);
            }
        }
    };
    if (PRE) {
#pragma unroll
        for (int t4 = 0; t4 < NQ; ++t4) mma4(t4, mvA[t4], wvA[t4]);
    } else {
#pragma unroll 4
        for (int t4 = 0; t4 < D / 8; ++t4) {          // d = 256: the weights do not fit in registers beside the gather (unrolled by 4: four steps' loads go together)
            float mv[4], wv[4];
            weights_of(t4, mv, wv);
            mma4(t4, mv, wv);
        }
    }

    // ---- 3. bias, s_out, leaky_relu; outputs of k_prep_h
    const float b_mu = p.b0[j];
    float b_p = 0.f, klb = 0.f;
    if (BAYES) {
        float z[4];
        normal4(p.eps_b0, j >> 2, (int64_t)(j & ~3), HH, z);
        float ls;
        const float sigma = softplus_rho_fast(p.rhob0[j], ls);
        const int jq = j & 3;
        b_p = sigma * (jq == 0 ? z[0] : jq == 1 ? z[1] : jq == 2 ? z[2] : z[3]);
        if (half == 0) klb = -ls + 0.5f * (sigma * sigma + b_mu * b_mu) - 0.5f;
    }
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * half, i = i0 + rl;
        float z = acc1[r] + b_mu;
        if (BAYES) {
            const uint32_t w = sign_word(p.so0.k0, p.so0.k1, (uint32_t)i, (uint32_t)wave);
            const float v = acc2[r] + b_p;
            z += ((w >> il) & 1u) ? -v : v;
        }
        float h = z > 0.f ? z : z * kLeakySlope;
        if (i >= p.B) h = 0.f;
        else p.act1[(int64_t)i * HH + j] = h;
        if (p.rflag && !(fabsf(h) <= p.h_limit)) bad = true;      // also NaN / inf
        p.hz[(int64_t)i * HH + j] = h;
        Hs[rl * HS_LD + j] = h;
        if (BAYES) {
            const uint32_t w1 = i < p.B ? sign_word(p.si1.k0, p.si1.k1, (uint32_t)i, (uint32_t)wave) : 0u;
            p.hs[(int64_t)i * HH + j] = ((w1 >> il) & 1u) ? -h : h;
            if (il == 0) { p.sinbits[(int64_t)i * 4 + wave] = w1; SW[rl * 4 + wave] = w1; }
        }
    }
    if (bad) *p.rflag = 1;
    if (BAYES && blockIdx.x == 0) {   // the hidden layer's KL, once
        const double s1 = wave_reduce_sum_d((double)klw), s2 = wave_reduce_sum_d((double)klb);
        if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
    }
    __syncthreads();
    if (BAYES && blockIdx.x == 0 && tid == 0 && p.kl)      // (kl == null: a head redone for another batch than the prefetched one - its KL terms, functions of the parameters alone, are counted)
        atomicAdd(p.kl, (red[0] + red[1] + red[2] + red[3]) * p.klw_w0 + (red[4] + red[5] + red[6] + red[7]) * p.klw_b0);

    // ---- 4. the dW kernel's planes of this K block: [plane = h hi, h lo, (hs hi, hs lo)][slot][32 rows] fp16, slot = 32 (j % 4) + j / 4 (k_prep_planes_T)
    if (p.hb) {
        uint32_t* tile = reinterpret_cast<uint32_t*>(p.hb + (size_t)blockIdx.x * (BAYES ? 4 : 2) * HH * 32);
        const int rp = tid & 15;                                      // rows 2 rp, 2 rp + 1
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int slot = (tid >> 4) + 16 * it, jj = 4 * (slot & 31) + (slot >> 5);
            const float x0 = Hs[(2 * rp) * HS_LD + jj], x1 = Hs[(2 * rp + 1) * HS_LD + jj];
            uint32_t pq[3];
            split_pair_np<2>(x0, x1, p.h_scale, pq);
            tile[(0 * HH + slot) * 16 + rp] = pq[0]; tile[(1 * HH + slot) * 16 + rp] = pq[1];
            if (BAYES) {
                const uint32_t b0 = (SW[(2 * rp) * 4 + (jj >> 5)] >> (jj & 31)) & 1u, b1 = (SW[(2 * rp + 1) * 4 + (jj >> 5)] >> (jj & 31)) & 1u;
                split_pair_np<2>(b0 ? -x0 : x0, b1 ? -x1 : x1, p.h_scale, pq);
                tile[(2 * HH + slot) * 16 + rp] = pq[0]; tile[(3 * HH + slot) * 16 + rp] = pq[1];
            }
        }
        if (BAYES && p.sinT && tid < HH) {   // k_sin_words_T for this K block: bit order of k_out_dw_q's fragment masks
            uint32_t w = 0u;
#pragma unroll
            for (int r = 0; r < 32; ++r) w |= ((SW[r * 4 + (tid >> 5)] >> (tid & 31)) & 1u) << (16 * (r & 1) + 4 * (r >> 3) + 3 - ((r & 7) >> 1));
            p.sinT[(size_t)blockIdx.x * HH + tid] = w;
        }
    }
}

bool head_supported(int D, int H) { return H == HH && (D == 64 || D == 128 || D == 256); }

void launch_head(hipStream_t st, const HeadArgs& a) {
    HeadArgs p = a;
    p.nrb = p.Bpad / 32;
    const int nbias = (p.bayes && p.M > 0) ? (int)std::min<int64_t>(((p.M + 3) / 4 + 255) / 256, 512) : 0;
    const size_t lds = (size_t)(32 * (p.D + 1) + 32 * HS_LD) * 4 + 32 * 4 * 4;
#define NTF_HEAD_D(DD) do { if (p.bayes) hipLaunchKernelGGL((k_head<true, DD>), dim3(p.nrb + nbias), dim3(256), lds, st, p);   \
                            else hipLaunchKernelGGL((k_head<false, DD>), dim3(p.nrb), dim3(256), lds, st, p); } while (0)
    if (p.D == 64) NTF_HEAD_D(64); else if (p.D == 128) NTF_HEAD_D(128); else NTF_HEAD_D(256);
#undef NTF_HEAD_D
}

}  // namespace ntf
