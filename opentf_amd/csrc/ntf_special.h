// The sparse fix-up of the fused output layer (src/mdl/fnn.py:32-46 - the positives and the sampled negatives of a row take the weight tpw, the positives
// label 1): argument block and device functions of the fix-up kernel (ntf_special.hip).
// Internal, not part of the C ABI.
#pragma once
#include "ntf_fused_common.h"

namespace ntf {

struct SpecialArgs {
    int B, M, Bpad, NCG, nCB, ns;
    int nslab;   // dh slabs to sum (NCG)
    const float *h, *hs, *mu, *mu_b, *wp, *bp, *slab, *lossp, *h_mask;
    const uint32_t *sbits, *sinbits;
    const int64_t *rows, *m_indptr, *neg; const int32_t* m_indices;
    float tpw, tnw, inv_B;
    float *dzT, *dh, *row_fix;
    uint32_t so_k0, so_k1; int so_inj;
    float dz_pack_scale;   // > 0: dzT holds packed fp16 plane pairs of dz * scale (fp16x3 step) ... unless *rflag is raised (the f32 kernels ran)
    const int* rflag;
    int c_lo;              // expert shard: labels and negatives name GLOBAL expert ids, this launch owns [c_lo, c_lo + M)
    const uint16_t* wp_pl; float wp_inv_scale;   // fp16x3 step: sigma * eps as the two fp16 planes the forward kernel multiplied with (k_out_fwd_h3x's tile layout); null: the f32 copy `wp`
    int fb_ncg;            // > 0: the split-product forward ran as several range launches (NCG / nslab count THEIR column groups); a step that fell back to the exact-f32 kernels
                           // has the whole-layer launch's fb_ncg column groups instead
};

// The special entries of a team - its positives (member CSR row) and its sampled negatives - as the sparse fix-up visits them: entry `sidx` of the row's
// npos + ns candidates -> (global expert id or -1, label).  A negative that names a member of the team, or repeats an earlier negative, is dropped (src/mdl/fnn.py:48-56
// draws distinct non-members; injected indices may not be).
__device__ __forceinline__ int special_candidate(const int32_t* __restrict__ m_indices, const int64_t* __restrict__ neg, int64_t pb, int npos, int ns, int i, int sidx, float& y) {
    y = 0.f;
    if (sidx < npos) { y = 1.f; return m_indices[pb + sidx]; }
    const int qn = sidx - npos;
    if (!neg || qn >= ns) return -1;
    int c = (int)neg[(int64_t)i * ns + qn];
    for (int k = 0; k < npos; ++k) if (m_indices[pb + k] == c) c = -1;
    for (int k = 0; k < qn; ++k) if (c >= 0 && (int)neg[(int64_t)i * ns + k] == c) c = -1;
    return c;
}
// The logit z (pre-activation) of one (team i, expert cc) entry by a QUARTER-WAVE of 16 lanes, lane l holding hidden units 8 l .. 8 l + 7 of h (hr) and of h * s_in (hsr):
// z = h . mu[cc] + mu_b[cc] + s_out(i, cc) ((h s_in) . Wp[cc] + bp[cc]).  The products are explicit fmaf chains (the value does not depend on how the
// compiler would contract them).  mu_r / wp_r: the weights it multiplied with (the caller's dh terms); so: the entry's s_out sign.
template <bool BAYES>
__device__ __forceinline__ float special_z16(const float* __restrict__ mu, const float* __restrict__ mu_b, const float* __restrict__ wp, const float* __restrict__ bp,
                                             const uint16_t* __restrict__ wp_pl, float wp_inv_scale, bool wp_planes, const uint32_t* __restrict__ sbits, int nCB,
                                             uint32_t so_k0, uint32_t so_k1, int so_inj, int i, int cc, int l, const float (&hr)[8], const float (&hsr)[8],
                                             float (&mu_r)[8], float (&wp_r)[8], float& so) {
    constexpr int H = 128;
    float d1 = 0.f, d2 = 0.f;
    {
        const float4 a = *reinterpret_cast<const float4*>(mu + (int64_t)cc * H + 8 * l), b = *reinterpret_cast<const float4*>(mu + (int64_t)cc * H + 8 * l + 4);
        mu_r[0] = a.x; mu_r[1] = a.y; mu_r[2] = a.z; mu_r[3] = a.w; mu_r[4] = b.x; mu_r[5] = b.y; mu_r[6] = b.z; mu_r[7] = b.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { d1 = fmaf(hr[k], mu_r[k], d1); wp_r[k] = 0.f; }
    if (BAYES) {
        if (wp_planes) {   // the lane's 8 consecutive hidden units of row cc: 16 bytes from each plane; value = (hi + lo) / scale - exactly what the dense pass multiplied with
            typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
            const uint16_t* r0 = wp_pl + ((int64_t)(cc >> 5) * 64 + (cc & 31)) * H + 8 * l;     // [tile of 32 rows][plane][row][H]
            const h8_t hi = *reinterpret_cast<const h8_t*>(r0), lo = *reinterpret_cast<const h8_t*>(r0 + 32 * H);
#pragma unroll
            for (int k = 0; k < 8; ++k) wp_r[k] = ((float)hi[k] + (float)lo[k]) * wp_inv_scale;
        } else {
            const float4 a = *reinterpret_cast<const float4*>(wp + (int64_t)cc * H + 8 * l), b = *reinterpret_cast<const float4*>(wp + (int64_t)cc * H + 8 * l + 4);
            wp_r[0] = a.x; wp_r[1] = a.y; wp_r[2] = a.z; wp_r[3] = a.w; wp_r[4] = b.x; wp_r[5] = b.y; wp_r[6] = b.z; wp_r[7] = b.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) d2 = fmaf(hsr[k], wp_r[k], d2);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) d1 += __shfl_xor(d1, o, 64);
    float z = d1 + mu_b[cc];
    so = 1.f;
    if (BAYES) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) d2 += __shfl_xor(d2, o, 64);
        const uint32_t sw_ = so_inj ? sbits[(int64_t)i * nCB + (cc >> 5)] : sign_word(so_k0, so_k1, (uint32_t)i, (uint32_t)(cc >> 5));
        so = ((sw_ >> (cc & 31)) & 1u) ? -1.f : 1.f;
        z += (d2 + bp[cc]) * so;
    }
    return z;
}
// d loss / d z of a special entry (label y, positive-weight tpw) as the fp16x3 step stores it: the two fp16 planes of dz * scale packed in a dword
__device__ __forceinline__ float special_dz(float z, float y, float tpw, float inv_B, float& sp, float& sg, float& dact) {
    bce_terms(z, sp, sg, dact);
    return tpw * (sg - y) * dact * inv_B;
}
__device__ __forceinline__ uint32_t special_dz_packed(float dzt, float scale) { uint32_t pq[3]; split_pair_np<2>(dzt, 0.f, scale, pq); return (pq[0] & 0xFFFFu) | (pq[1] << 16); }

// one wave per team: loss terms, dz and d(hidden) terms of the special entries + the sums of the forward kernel's partials (ntf_special.hip)
void launch_out_special(hipStream_t st, int H, bool bayes, bool train, bool dh, const SpecialArgs& s);

}  // namespace ntf
