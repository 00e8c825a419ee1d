// Sparse fix-up side of the fused output layer for gfx950 (split from ntf_fused.hip in round 5): k_out_special - one wave per team: the special entries' loss terms, dz and
// d(hidden) terms, and the sums of the forward kernel's partials.
#include "ntf_special.h"

namespace ntf {

// One wave per team.  H = 128: the wave works as FOUR quarter-waves of 16 lanes x 8 consecutive hidden units, each quarter taking every fourth
// special entry (and every fourth dh slab): the ~8 dependent dot-product / reduction / BCE chains of a team run four abreast, rows are read as
// 32-byte pieces (round 2: 55 -> ~20 us per step at B = 1000).  Other widths keep one entry at a time over the whole wave (NV values per lane).
template <int H, bool BAYES, bool TRAIN, bool DH>
__global__ __launch_bounds__(64) void k_out_special(SpecialArgs p) {
    constexpr bool QUAD = (H == 128);
    constexpr int NQ = QUAD ? 4 : 1;                 // entries in flight
    constexpr int NV = QUAD ? 8 : (H + 63) / 64;     // hidden units per lane
    const int i = blockIdx.x, lane = threadIdx.x;
    const int q = QUAD ? (lane >> 4) : 0, l = QUAD ? (lane & 15) : lane;
    auto hidx = [&](int k) { return QUAD ? 8 * l + k : l + 64 * k; };      // this lane's k-th hidden unit
    const bool packed = p.dz_pack_scale > 0.f && !(p.rflag && *p.rflag);
    const bool wp_planes = BAYES && QUAD && p.wp_pl != nullptr && !(p.rflag && *p.rflag);   // (a step that fell back to the f32 kernels: its planes are saturated, the f32 copy was made for it)
    const int ncg = (p.fb_ncg > 0 && p.rflag && *p.rflag) ? p.fb_ncg : p.NCG, nslab = (p.fb_ncg > 0 && p.rflag && *p.rflag) ? p.fb_ncg : p.nslab;
    float rl = 0.f;
    for (int cg = lane; cg < ncg; cg += 64) rl += p.lossp[(int64_t)i * ncg + cg];
    rl = wave_reduce_sum(rl);
    float acc[NV], hr[NV], hsr[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = hidx(k);
        acc[k] = 0.f; hr[k] = 0.f; hsr[k] = 0.f;
        if (j < H) { hr[k] = p.h[(int64_t)i * H + j]; if (BAYES) hsr[k] = p.hs[(int64_t)i * H + j]; }
    }
    if (TRAIN && DH) {
        for (int cg = q; cg < nslab; cg += NQ) {
            const float* sl = p.slab + ((int64_t)cg * p.Bpad + i) * H;
#pragma unroll
            for (int k = 0; k < NV; ++k) { const int j = hidx(k); if (j < H) acc[k] += sl[j]; }
        }
    }
    auto group_sum = [&](float v) {    // sum over the lanes that share one entry: a quarter (16 lanes) or the whole wave
        if (!QUAD) return wave_reduce_sum(v);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    const int64_t team = p.rows[i];
    const int64_t pb = p.m_indptr[team];
    const int npos = (int)(p.m_indptr[team + 1] - pb);
    const int total = npos + (p.neg ? p.ns : 0);
    float fix = 0.f;
    for (int s0 = 0; s0 < total; s0 += NQ) {       // wave-uniform trip count: the shuffles below need every lane
        const int sidx = s0 + q;
        float y = 0.f;
        int c = sidx < total ? special_candidate(p.m_indices, p.neg, pb, npos, p.ns, i, sidx, y) : -1;
        if (c >= 0) c -= p.c_lo;                   // global -> this shard's expert index (entries of other shards fall outside [0, M))
        const bool live = c >= 0 && c < p.M;
        const int cc = live ? c : 0;
        float mu_r[NV], wp_r[NV], z, so = 1.f;
        if constexpr (QUAD) z = special_z16<BAYES>(p.mu, p.mu_b, p.wp, p.bp, p.wp_pl, p.wp_inv_scale, wp_planes, p.sbits, p.nCB, p.so_k0, p.so_k1, p.so_inj, i, cc, l, hr, hsr, mu_r, wp_r, so);
        else {
            float d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int j = hidx(k);
                mu_r[k] = 0.f; wp_r[k] = 0.f;
                if (j < H) { mu_r[k] = p.mu[(int64_t)cc * H + j]; d1 += hr[k] * mu_r[k]; if (BAYES) { wp_r[k] = p.wp[(int64_t)cc * H + j]; d2 += hsr[k] * wp_r[k]; } }
            }
            d1 = group_sum(d1);
            z = d1 + p.mu_b[cc];
            if (BAYES) {
                d2 = group_sum(d2);
                const uint32_t sw_ = p.so_inj ? p.sbits[(int64_t)i * p.nCB + (cc >> 5)] : sign_word(p.so_k0, p.so_k1, (uint32_t)i, (uint32_t)(cc >> 5));
                so = ((sw_ >> (cc & 31)) & 1u) ? -1.f : 1.f;
                z += (d2 + p.bp[cc]) * so;
            }
        }
        float sp, sg, dact;
        const float dzt = special_dz(z, y, p.tpw, p.inv_B, sp, sg, dact);
        const float lz = z > 0.f ? z : z * kLeakySlope;
        if (live && l == 0) fix += p.tpw * (sp - lz * y) - p.tnw * sp;
        if (TRAIN && live) {
            const float delta = dzt - p.tnw * sg * dact * p.inv_B;
            if (l == 0) {
                if (packed) reinterpret_cast<uint32_t*>(p.dzT)[dzt_index(c, i, p.Bpad)] = special_dz_packed(dzt, p.dz_pack_scale);
                else p.dzT[dzt_index(c, i, p.Bpad)] = dzt;
            }
            if (DH) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int j = hidx(k);
                    if (j < H) {
                        acc[k] += delta * mu_r[k];
                        if (BAYES) {
                            const float si = ((p.sinbits[(int64_t)i * (H / 32) + (j >> 5)] >> (j & 31)) & 1u) ? -1.f : 1.f;
                            acc[k] += delta * so * wp_r[k] * si;
                        }
                    }
                }
            }
        }
    }
    if (QUAD) {   // quarters -> one
        fix += __shfl_xor(fix, 16, 64); fix += __shfl_xor(fix, 32, 64);
#pragma unroll
        for (int k = 0; k < NV; ++k) { acc[k] += __shfl_xor(acc[k], 16, 64); acc[k] += __shfl_xor(acc[k], 32, 64); }
    }
    if (lane == 0) p.row_fix[i] = rl + fix;
    if (TRAIN && DH && q == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = hidx(k);
            if (j < H) {
                float v = acc[k];
                if (p.h_mask) v *= (p.h_mask[(int64_t)i * H + j] > 0.f) ? 1.f : kLeakySlope;
                p.dh[(int64_t)i * H + j] = v;
            }
        }
    }
}

void launch_out_special(hipStream_t st, int H, bool bayes, bool train, bool dh, const SpecialArgs& s) {
#ifdef NTF_DIAG
    static const int diag_skip = getenv("NTF_SKIP") ? atoi(getenv("NTF_SKIP")) : 0;     // timing only (results garbage): 4 - no sparse fix-up launch
    if (diag_skip == 4 && train) return;
#endif
#define NTF_SPK(HH, BY) do { if (!train) hipLaunchKernelGGL((k_out_special<HH, BY, false, false>), dim3(s.B), dim3(64), 0, st, s);         \
        else if (dh) hipLaunchKernelGGL((k_out_special<HH, BY, true, true>), dim3(s.B), dim3(64), 0, st, s);                                  \
        else hipLaunchKernelGGL((k_out_special<HH, BY, true, false>), dim3(s.B), dim3(64), 0, st, s); } while (0)
#define NTF_SPH(HH) do { if (bayes) NTF_SPK(HH, true); else NTF_SPK(HH, false); } while (0)
    if (H == 128) NTF_SPH(128); else if (H == 64) NTF_SPH(64); else NTF_SPH(32);
#undef NTF_SPH
#undef NTF_SPK
}

}  // namespace ntf
