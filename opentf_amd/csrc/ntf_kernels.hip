// gfx950 kernels of the OpeNTF fnn/bnn hot path: generic f32-MFMA GEMM, CSR gathers, Flipout operand
// producers, sparse-label weighted-BCE, negative samplers, KL, Adam.  The fused output-layer kernels
// live in ntf_fused.hip.  Written for CDNA4 only (wave64, v_mfma_f32_32x32x2_f32).
#include "ntf_kernels.h"
#include <cstdlib>
#include "ntf_device.h"
#include <algorithm>
#include <cstdint>

namespace ntf {

// =====================================================================================
// Generic GEMM  C(m,n) = sum_k A(m,k) B(k,n)  on v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain).
// 64x64 block tile, 4 waves each owning a 32x32 accumulator, K staged 32 deep through LDS as
// [row][k] images with an odd row stride (33) so that both the staging writes and the fragment reads
// (32 lanes = 32 rows, same k) are bank-conflict free.  Arbitrary strides: each operand is read along
// whichever of its two axes is contiguous.  Used for the hidden layers, for odd shapes, and as the
// unfused reference path of the output layer.
// =====================================================================================
constexpr int GBM = 64, GBN = 64, GBK = 32, GLD = GBK + 1;

__global__ __launch_bounds__(256) void k_gemm(GemmArgs a) {
    __shared__ float As[GBM * GLD];
    __shared__ float Bs[GBN * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    int kper = (a.K + a.ksplit - 1) / a.ksplit;
    kper = (kper + GBK - 1) / GBK * GBK;
    const int kbeg = blockIdx.z * kper;
    const int kend = min(a.K, kbeg + kper);
    const bool a_kc = (a.sak == 1), b_nc = (a.sbn == 1);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;

    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        float ra[8], rb[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            int mm, kk;
            if (a_kc) { kk = tid & 31; mm = (tid >> 5) + p * 8; } else { mm = tid & 63; kk = (tid >> 6) + p * 4; }
            const int gm = m0 + mm, gk = k0 + kk;
            float v = 0.f;
            if (gm < a.M && gk < kend) {
                v = a.A[(int64_t)gm * a.sam + (int64_t)gk * a.sak];
                if (a.sa.enabled) v *= a.sa_t ? sign_at(a.sa, gk, gm) : sign_at(a.sa, gm, gk);
            }
            ra[p] = v;
            int nn, kb;
            if (b_nc) { nn = tid & 63; kb = (tid >> 6) + p * 4; } else { kb = tid & 31; nn = (tid >> 5) + p * 8; }
            const int gn = n0 + nn, gkb = k0 + kb;
            float w = 0.f;
            if (gn < a.N && gkb < kend) {
                w = a.B[(int64_t)gkb * a.sbk + (int64_t)gn * a.sbn];
                if (a.sb.enabled) w *= a.sb_t ? sign_at(a.sb, gn, gkb) : sign_at(a.sb, gkb, gn);
            }
            rb[p] = w;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            int mm, kk;
            if (a_kc) { kk = tid & 31; mm = (tid >> 5) + p * 8; } else { mm = tid & 63; kk = (tid >> 6) + p * 4; }
            As[mm * GLD + kk] = ra[p];
            int nn, kb;
            if (b_nc) { nn = tid & 63; kb = (tid >> 6) + p * 4; } else { kb = tid & 31; nn = (tid >> 5) + p * 8; }
            Bs[nn * GLD + kb] = rb[p];
        }
        __syncthreads();
        const float* ap = &As[(wm * 32 + (lane & 31)) * GLD + (lane >> 5)];
        const float* bp = &Bs[(wn * 32 + (lane & 31)) * GLD + (lane >> 5)];
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk], bp[2 * kk], acc, 0, 0, 0);
    }

    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= a.N) return;
    const float bias = (a.bias && blockIdx.z == 0) ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= a.M) continue;
        float v = acc[r] * a.alpha + bias;
        if (a.sc.enabled) v *= sign_at(a.sc, row, col);
        if (a.mask) v *= (a.mask[(int64_t)row * a.ldmask + col] > 0.f) ? 1.f : kLeakySlope;
        if (a.ksplit > 1) {
            a.slab[((int64_t)blockIdx.z * a.M + row) * a.N + col] = v;  // partial sums; k_gemm_reduce adds them in slice order
        } else {
            if (a.accumulate) v += a.C[(int64_t)row * a.ldc + col];
            if (a.C) a.C[(int64_t)row * a.ldc + col] = v;
            if (a.Act) a.Act[(int64_t)row * a.ldact + col] = v > 0.f ? v : v * kLeakySlope;
        }
    }
}

__global__ void k_gemm_reduce(const float* __restrict__ slab, int ks, int64_t mn, int N, float* __restrict__ C, int64_t ldc, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= mn) return;
    float s = 0.f;
    for (int z = 0; z < ks; ++z) s += slab[(int64_t)z * mn + e];
    const int64_t o = (e / N) * ldc + (e % N);
    C[o] = accumulate ? C[o] + s : s;
}

void launch_gemm(hipStream_t st, const GemmArgs& a) {
    if (a.M <= 0 || a.N <= 0) return;
    dim3 grid((a.N + GBN - 1) / GBN, (a.M + GBM - 1) / GBM, a.ksplit > 1 ? a.ksplit : 1);
    hipLaunchKernelGGL(k_gemm, grid, dim3(256), 0, st, a);
    if (a.ksplit > 1) {
        const int64_t mn = (int64_t)a.M * a.N;
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, a.slab, a.ksplit, mn, a.N, a.C, a.ldc, a.accumulate);
    }
}

// =====================================================================================
// CSR gathers.  One team per sub-group of G lanes (G*4 >= d floats, 16-byte loads), the team's column ids
// are read coalesced by the sub-group and broadcast with shuffles; up to 8 table rows are in flight per
// sub-group before the (CSR-ordered, f32) accumulation, which keeps the result bit-identical to the
// sequential host sum.  mean=1: Gnn.get_dense_vecs (src/mdl/emb/gnn.py:485); mean=0: multi-hot @ W^T.
// =====================================================================================
template <int G>
__global__ __launch_bounds__(256) void k_gather_pool(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                     const float* __restrict__ table, const int64_t* __restrict__ rows,
                                                     int64_t n, int d, int mean, int vec, float* __restrict__ out) {
    constexpr int TPW = 64 / G;  // teams per wave
    const int lane = threadIdx.x & 63, sub = lane / G, sl = lane % G;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t t = wave_global * TPW + sub;
    const bool live = t < n;
    const int64_t team = live ? (rows ? rows[t] : t) : 0;
    const int64_t beg = live ? indptr[team] : 0, end = live ? indptr[team + 1] : 0;
    const int nnz = (int)(end - beg);
    const int nchunk = (d + 4 * G - 1) / (4 * G);  // float4 chunks per lane (1 when d <= 4G)
    for (int ch = 0; ch < nchunk; ++ch) {
        const int c0 = (ch * G + sl) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = 0; base < nnz; base += G) {
            const int mine = base + sl;
            const int my_idx = (mine < nnz) ? indices[beg + mine] : 0;
            const int cnt = min(G, nnz - base);
            for (int j0 = 0; j0 < cnt; j0 += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u;
                    const int s = __shfl(my_idx, sub * G + (j < cnt ? j : 0), 64);
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j < cnt) {
                        const float* rp = table + (int64_t)s * d + c0;
                        if (vec && c0 + 3 < d) v[u] = *reinterpret_cast<const float4*>(rp);
                        else { if (c0 < d) v[u].x = rp[0]; if (c0 + 1 < d) v[u].y = rp[1]; if (c0 + 2 < d) v[u].z = rp[2]; if (c0 + 3 < d) v[u].w = rp[3]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u < cnt) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
        if (live && c0 < d) {
            if (mean) { const float c = (float)nnz; acc.x /= c; acc.y /= c; acc.z /= c; acc.w /= c; }
            float* op = out + t * d + c0;
            if (vec && c0 + 3 < d) *reinterpret_cast<float4*>(op) = acc;
            else { op[0] = acc.x; if (c0 + 1 < d) op[1] = acc.y; if (c0 + 2 < d) op[2] = acc.z; if (c0 + 3 < d) op[3] = acc.w; }
        }
    }
}

void launch_gather_meanpool(hipStream_t st, const int64_t* indptr, const int32_t* indices, const float* table,
                            const int64_t* rows, int64_t n, int d, int mean, float* out) {
    if (n <= 0) return;
    const bool vec_ok = (d % 4 == 0);
    // sub-group width: smallest power of two >= d/4, within [16, 64]; unaligned d falls back to G=64 scalar tails
    int G = 16;
    while (G < 64 && G * 4 < d) G *= 2;
    if (!vec_ok) G = 64;
    const int tpw = 64 / G;
    const int64_t waves = (n + tpw - 1) / tpw;
    const int64_t blocks = (waves + 3) / 4;
    if (G == 16) hipLaunchKernelGGL(k_gather_pool<16>, dim3((unsigned)blocks), dim3(256), 0, st, indptr, indices, table, rows, n, d, mean, vec_ok ? 1 : 0, out);
    else if (G == 32) hipLaunchKernelGGL(k_gather_pool<32>, dim3((unsigned)blocks), dim3(256), 0, st, indptr, indices, table, rows, n, d, mean, vec_ok ? 1 : 0, out);
    else hipLaunchKernelGGL(k_gather_pool<64>, dim3((unsigned)blocks), dim3(256), 0, st, indptr, indices, table, rows, n, d, mean, vec_ok ? 1 : 0, out);
}

__global__ void k_gather_dense_rows(const float* __restrict__ X, int d, const int64_t* __restrict__ rows, int64_t n, float* __restrict__ out) {
    const int64_t i = blockIdx.x;
    const float* src = X + rows[i] * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[i * d + c] = src[c];
}
void launch_gather_dense_rows(hipStream_t st, const float* X, int d, const int64_t* rows, int64_t n, float* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_dense_rows, dim3((unsigned)n), dim3(d >= 256 ? 256 : 64), 0, st, X, d, rows, n, out);
}

// Multi-hot input (src/mdl/ntf.py:23: the team's skill row densified to 0/1): the first layer x W^T is the SUM of the W columns
// of the team's skills, so X is never materialised.  The engine keeps this layer's weight segments TRANSPOSED, [S, H] (in x out; the host boundary
// transposes them back to the reference's [H, S]): the H threads of a team read one contiguous 4 H-byte row per skill instead of H values
// S*4 bytes apart, and the gradient scatter below adds whole rows (the coalesced atomic form that runs at full rate).  One thread per (row, h).
//   z = b + sum_s W[h,s]  (+ (sum_s s_in(i,s) Wp[h,s] + bp[h]) * s_out(i,h) for Flipout);   act = leaky_relu(z)
__global__ void k_multihot_fwd(const int64_t* __restrict__ rows, int S, int H, const int64_t* __restrict__ indptr,
                               const int32_t* __restrict__ indices, const float* __restrict__ W, const float* __restrict__ b,
                               const float* __restrict__ Wp, const float* __restrict__ bp, SignSpec sin, SignSpec sout,
                               float* __restrict__ act) {
    const int i = blockIdx.x, h = blockIdx.y * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const int64_t team = rows[i], p0 = indptr[team], p1 = indptr[team + 1];
    float z = b[h];
    if (Wp) {
        float zp = 0.f;
        for (int64_t p = p0; p < p1; ++p) { const int64_t s = indices[p]; z += W[s * H + h]; zp += Wp[s * H + h] * sign_at(sin, i, s); }
        z += (zp + bp[h]) * sign_at(sout, i, h);
    } else {
        for (int64_t p = p0; p < p1; ++p) z += W[(int64_t)indices[p] * H + h];
    }
    act[(int64_t)i * H + h] = z > 0.f ? z : kLeakySlope * z;
}
void launch_multihot_fwd(hipStream_t st, const int64_t* rows, int B, int S, int H, const int64_t* indptr, const int32_t* indices,
                         const float* W, const float* b, const float* Wp, const float* bp, SignSpec sin, SignSpec sout, float* act) {
    if (B <= 0) return;
    const int bs = H >= 256 ? 256 : (H + 63) / 64 * 64;
    hipLaunchKernelGGL(k_multihot_fwd, dim3((unsigned)B, (unsigned)((H + bs - 1) / bs)), dim3(bs), 0, st, rows, S, H, indptr, indices, W, b, Wp, bp,
                       sin, sout, act);
}
// its weight gradient is a scatter: gW[s,h] += dz[i,h] for every skill s of row i (gWp[s,h] += dz*s_out(i,h)*s_in(i,s)), [S, H] like the weights;
// gW/gWp zeroed by the caller.  Hardware f32 atomics: the per-element sums have at most (batch frequency of s) terms.
__global__ void k_multihot_bwd(const int64_t* __restrict__ rows, int S, int H, const int64_t* __restrict__ indptr,
                               const int32_t* __restrict__ indices, const float* __restrict__ dZ, SignSpec sin, SignSpec sout,
                               float* __restrict__ gW, float* __restrict__ gWp, uint8_t* __restrict__ touched) {
    const int i = blockIdx.x, h = blockIdx.y * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const int64_t team = rows[i], p0 = indptr[team], p1 = indptr[team + 1];
    const float dz = dZ[(int64_t)i * H + h];
    if (touched && h == 0) for (int64_t p = p0; p < p1; ++p) touched[indices[p]] = 1;
    if (gWp) {
        const float dzs = dz * sign_at(sout, i, h);
        for (int64_t p = p0; p < p1; ++p) { const int64_t s = indices[p]; unsafeAtomicAdd(gW + s * H + h, dz); unsafeAtomicAdd(gWp + s * H + h, dzs * sign_at(sin, i, s)); }
    } else {
        for (int64_t p = p0; p < p1; ++p) unsafeAtomicAdd(gW + (int64_t)indices[p] * H + h, dz);
    }
}
void launch_multihot_bwd(hipStream_t st, const int64_t* rows, int B, int S, int H, const int64_t* indptr, const int32_t* indices,
                         const float* dZ, SignSpec sin, SignSpec sout, float* gW, float* gWp, uint8_t* touched) {
    if (B <= 0) return;
    const int bs = H >= 256 ? 256 : (H + 63) / 64 * 64;
    hipLaunchKernelGGL(k_multihot_bwd, dim3((unsigned)B, (unsigned)((H + bs - 1) / bs)), dim3(bs), 0, st, rows, S, H, indptr, indices, dZ, sin, sout,
                       gW, gWp, touched);
}

// =====================================================================================
// Flipout operand producer / gradient finaliser / KL  (bayesian-torch LinearFlipout + kl_div, restated)
// =====================================================================================
// out = softplus(rho) * eps; when mu is given, also adds this tensor's KL(N(mu, sigma^2) || N(0,1)) * w to kl_out.
// Grid-stride over quads with a bounded grid, so that the KL costs one double atomic per workgroup (<= 2048 in all).
// planes_w / planes_mu (output layer, bf16x6 arithmetic): also the bf16 split planes of out and of pmu for the forward kernel.
__global__ __launch_bounds__(256) void k_flipout_perturb(const float* __restrict__ rho, const float* __restrict__ mu, int64_t n, NormalSpec eps,
                                                         float* __restrict__ out, double w, double* kl_out, uint16_t* __restrict__ planes_w,
                                                         uint16_t* __restrict__ planes_mu, const float* __restrict__ pmu, int H, int np, float pscale,
                                                         int* __restrict__ rflag, const int* __restrict__ only_if, PerturbChain ch) {
    if (only_if && __builtin_nontemporal_load(only_if) == 0) return;   // (the f32 copy of sigma * eps for a step that fell back to the exact-f32 kernels, see launch_flipout_perturb)
    const int64_t quads = (n + 3) / 4;
    float kl = 0.f, amax = 0.f, amax_mu = 0.f;
    if (rflag && ch.raise_if && blockIdx.x == 0 && threadIdx.x == 0 && *ch.raise_if) *rflag = 1;      // (the planes of mu - not rewritten by this launch - left the fp16 window when they were made)
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = q * 4;
        float z[4];
        normal4(eps, q, e0, n, z);
        if (e0 + 3 < n) {
            const float4 r4 = *reinterpret_cast<const float4*>(rho + e0);
            const float rv[4] = {r4.x, r4.y, r4.z, r4.w};
            float ov[4], mv[4] = {0.f, 0.f, 0.f, 0.f};
            if (mu) { const float4 m4 = *reinterpret_cast<const float4*>(mu + e0); mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float ls;
                const float sigma = softplus_rho_fast(rv[j], ls);
                ov[j] = sigma * z[j];
                if (mu) kl += -ls + 0.5f * (sigma * sigma + mv[j] * mv[j]) - 0.5f;
            }
            if (out) *reinterpret_cast<float4*>(out + e0) = make_float4(ov[0], ov[1], ov[2], ov[3]);      // (null: the planes alone - a lean evaluation step; a range fallback makes the f32 copy itself)
            if (planes_w) {   // H % 4 == 0: the quad lies in one row
                // (H a power of two - 128 on every fused path: a shift; the 64-bit division costs ~80 vector instructions of this loop's ~350)
                const bool p2 = (H & (H - 1)) == 0;
                const int64_t row = p2 ? (e0 >> (31 - __builtin_clz(H))) : e0 / H; const int j = p2 ? (int)(e0 & (H - 1)) : (int)(e0 - row * H);
                const float4 m4 = planes_mu ? *reinterpret_cast<const float4*>(pmu + e0) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (np == 3) {
                    planes_store_quad<3>(planes_w, row, j, H, ov[0], ov[1], ov[2], ov[3], 1.f);
                    if (planes_mu) planes_store_quad<3>(planes_mu, row, j, H, m4.x, m4.y, m4.z, m4.w, 1.f);
                } else {
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(ov[0]), fabsf(ov[1])), fmaxf(fabsf(ov[2]), fabsf(ov[3]))));
                    if (planes_mu) amax_mu = fmaxf(amax_mu, fmaxf(fmaxf(fabsf(m4.x), fabsf(m4.y)), fmaxf(fabsf(m4.z), fabsf(m4.w))));
                    planes_store_quad<2>(planes_w, row, j, H, ov[0], ov[1], ov[2], ov[3], pscale);
                    if (planes_mu) planes_store_quad<2>(planes_mu, row, j, H, m4.x, m4.y, m4.z, m4.w, pscale);
                }
            }
        } else {
            for (int j = 0; j < 4 && e0 + j < n; ++j) {
                const float sigma = softplus_rho(rho[e0 + j]);
                if (out) out[e0 + j] = sigma * z[j];
                if (mu) { const float m = mu[e0 + j]; kl += -logf(sigma) + 0.5f * (sigma * sigma + m * m) - 0.5f; }
            }
        }
    }
    if (rflag && !(fmaxf(amax, amax_mu) * pscale <= 65504.f)) *rflag = 1;   // an operand of the fp16x3 products leaves the fp16 window (or is NaN)
    if (ch.mu_flag_out && !(amax_mu * pscale <= 65504.f)) *ch.mu_flag_out = 1;
    if (mu) {
        const double s = block_reduce_sum_d((double)kl);
        if (threadIdx.x == 0) { atomicAdd(kl_out, s * w); if (ch.kl_out2) atomicAdd(ch.kl_out2, s * w); }
    }
}
void launch_flipout_perturb(hipStream_t st, const float* rho, const float* mu, int64_t n, NormalSpec eps, float* out, double w, double* kl_out,
                            uint16_t* planes_w, uint16_t* planes_mu, const float* pmu, int H, int np, float pscale, int* rflag, const int* only_if, PerturbChain ch) {
    if (n <= 0) return;
    const int64_t quads = (n + 3) / 4;
    const int blocks = (int)std::min<int64_t>((quads + 255) / 256, only_if ? 256 : 2048);   // (only_if: a no-op in all but the rarest step - one short round of workgroups)
    hipLaunchKernelGGL(k_flipout_perturb, dim3(blocks), dim3(256), 0, st, rho, mu, n, eps, out, w, kl_out, planes_w, planes_mu, pmu, H, np, pscale, (planes_w && np == 2) ? rflag : nullptr, only_if, ch);
}

__global__ void k_flipout_grad_finalize(const float* __restrict__ mu, const float* __restrict__ rho, float* __restrict__ g_mu,
                                        float* __restrict__ g_rho, int64_t n, NormalSpec eps, float klw) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e0 = q * 4;
    if (e0 >= n) return;
    float z[4];
    normal4(eps, q, e0, n, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t e = e0 + j;
        if (e >= n) break;
        const float r = rho[e];
        const float sg = 1.f / (1.f + expf(-r));  // d softplus / d rho
        const float sigma = softplus_rho(r);
        g_rho[e] = g_rho[e] * z[j] * sg + klw * (sigma - 1.f / sigma) * sg;
        g_mu[e] += klw * mu[e];
    }
}
void launch_flipout_grad_finalize(hipStream_t st, const float* mu, const float* rho, float* g_mu, float* g_rho, int64_t n,
                                  NormalSpec eps, float klw) {
    if (n <= 0) return;
    const int64_t quads = (n + 3) / 4;
    hipLaunchKernelGGL(k_flipout_grad_finalize, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, mu, rho, g_mu, g_rho, n, eps, klw);
}

__global__ void k_kl_value(const float* __restrict__ mu, const float* __restrict__ rho, int64_t n, double w, double* out) {
    double s = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float sigma = softplus_rho(rho[e]);
        const float m = mu[e];
        s += (double)(-logf(sigma) + 0.5f * (sigma * sigma + m * m) - 0.5f);
    }
    s = block_reduce_sum_d(s);
    if (threadIdx.x == 0) atomicAdd(out, s * w);
}
void launch_kl_value(hipStream_t st, const float* mu, const float* rho, int64_t n, double w, double* out) {
    if (n <= 0) return;
    int blocks = (int)std::min<int64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_kl_value, dim3(blocks), dim3(256), 0, st, mu, rho, n, w, out);
}

// =====================================================================================
// Output-layer loss with sparse labels (src/mdl/fnn.py:32-46,135): dense pass + sparse fix-up.
// =====================================================================================
constexpr int LCH = 1024;  // columns per block in the dense pass
int loss_dense_nchunk(int M) { return (M + LCH - 1) / LCH; }

__global__ __launch_bounds__(256) void k_loss_dense(const float* __restrict__ Z, int64_t ld, int M, float tnw, float inv_B,
                                                    float* __restrict__ dZ, float* __restrict__ partial, int nchunk) {
    const int i = blockIdx.y, ch = blockIdx.x;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LCH / 256; ++j) {
        const int c = ch * LCH + j * 256 + threadIdx.x;
        if (c < M) {
            const float z = Z[(int64_t)i * ld + c];
            float sp, sg, dact;
            bce_terms(z, sp, sg, dact);
            s += sp;
            if (dZ) dZ[(int64_t)i * ld + c] = tnw * sg * dact * inv_B;
        }
    }
    s = block_reduce_sum(s);
    if (threadIdx.x == 0) partial[(int64_t)i * nchunk + ch] = tnw * s;
}
void launch_loss_dense(hipStream_t st, const float* Z, int64_t ld, int B, int M, float tnw, float inv_B, float* dZ, float* partial, int nchunk) {
    hipLaunchKernelGGL(k_loss_dense, dim3(nchunk, B), dim3(256), 0, st, Z, ld, M, tnw, inv_B, dZ, partial, nchunk);
}

// one wave per row; lane j takes special j (positives first, then the selected negatives)
__global__ __launch_bounds__(64) void k_loss_special(const float* __restrict__ Z, int64_t ld, int M, const int64_t* __restrict__ rows,
                                                     const int64_t* __restrict__ m_indptr, const int32_t* __restrict__ m_indices,
                                                     const int64_t* __restrict__ neg, int ns, float tpw, float tnw, float inv_B,
                                                     float* __restrict__ dZ, float* __restrict__ row_fix) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int64_t team = rows[i];
    const int64_t pb = m_indptr[team];
    const int npos = (int)(m_indptr[team + 1] - pb);
    const int total = npos + (neg ? ns : 0);
    float fix = 0.f;
    for (int j = lane; j < total; j += 64) {
        int c; float y; bool skip = false;
        if (j < npos) { c = m_indices[pb + j]; y = 1.f; }
        else {
            const int q = j - npos;
            c = (int)neg[(int64_t)i * ns + q]; y = 0.f;
            for (int p = 0; p < npos; ++p) if (m_indices[pb + p] == c) skip = true;       // selected a positive: stays y=1, handled above
            for (int p = 0; p < q; ++p) if ((int)neg[(int64_t)i * ns + p] == c) skip = true;  // duplicate pick
        }
        if (skip || c < 0 || c >= M) continue;
        const float z = Z[(int64_t)i * ld + c];
        float sp, sg, dact;
        bce_terms(z, sp, sg, dact);
        const float l = z > 0.f ? z : z * kLeakySlope;
        fix += tpw * (sp - l * y) - tnw * sp;
        if (dZ) dZ[(int64_t)i * ld + c] = tpw * (sg - y) * dact * inv_B;
    }
    fix = wave_reduce_sum(fix);
    if (lane == 0) row_fix[i] = fix;
}
void launch_loss_special(hipStream_t st, const float* Z, int64_t ld, int B, int M, const int64_t* rows, const int64_t* m_indptr,
                         const int32_t* m_indices, const int64_t* neg, int ns, float tpw, float tnw, float inv_B, float* dZ, float* row_fix) {
    hipLaunchKernelGGL(k_loss_special, dim3(B), dim3(64), 0, st, Z, ld, M, rows, m_indptr, m_indices, neg, ns, tpw, tnw, inv_B, dZ, row_fix);
}

__global__ __launch_bounds__(256) void k_loss_finalize(const float* __restrict__ partial, int nchunk, const float* __restrict__ row_fix, int B,
                                                       float inv_B, const double* __restrict__ kl, double kl_scale, float* out,
                                                       double* acc, int64_t* acc_steps) {
    double s = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) {
        float r = 0.f;
        for (int c = 0; c < nchunk; ++c) r += partial[(int64_t)i * nchunk + c];
        s += (double)(r + row_fix[i]);
    }
    s = block_reduce_sum_d(s);
    if (threadIdx.x == 0) {
        double loss = s * (double)inv_B + (kl ? kl[0] * kl_scale : 0.0);
        out[0] = (float)loss;
        if (acc) { acc[0] += (double)(float)loss; acc_steps[0] += 1; }
    }
}
void launch_loss_finalize(hipStream_t st, const float* partial, int nchunk, const float* row_fix, int B, float inv_B, const double* kl,
                          double kl_scale, float* out, double* acc, int64_t* acc_steps) {
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, st, partial, nchunk, row_fix, B, inv_B, kl, kl_scale, out, acc, acc_steps);
}

__global__ __launch_bounds__(256) void k_bias_grad(const float* __restrict__ dZ, int64_t ld, int B, int N, SignSpec sout,
                                                   float* __restrict__ g_b, float* __restrict__ g_pert) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) {
        const float v = dZ[(int64_t)b * ld + n];
        s1 += v;
        if (g_pert) s2 += v * sign_at(sout, b, n);
    }
    g_b[n] = s1;
    if (g_pert) g_pert[n] = s2;
}
// narrow layers (hidden): one wave per column, lanes stride over the batch rows, fixed-order shuffle reduction
__global__ __launch_bounds__(256) void k_bias_grad_wave(const float* __restrict__ dZ, int64_t ld, int B, int N, SignSpec sout,
                                                        float* __restrict__ g_b, float* __restrict__ g_pert) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = lane; b < B; b += 64) {
        const float v = dZ[(int64_t)b * ld + n];
        s1 += v;
        if (g_pert) s2 += v * sign_at(sout, b, n);
    }
    s1 = wave_reduce_sum(s1); s2 = wave_reduce_sum(s2);
    if (lane == 0) { g_b[n] = s1; if (g_pert) g_pert[n] = s2; }
}
// narrow layers under a wide minibatch (an expert shard steps G x b rows): row chunks of 32 read coalesced, one partial row per chunk, then a
// fixed-order sum over the chunks (the one-wave-per-column form above reads a column with a stride of ld floats: 64 bytes fetched per element used)
__global__ __launch_bounds__(256) void k_bias_grad_rows(const float* __restrict__ dZ, int64_t ld, int B, int N, SignSpec sout, float* __restrict__ part, int has_pert) {
    const int b0 = blockIdx.x * 32, b1 = min(B, b0 + 32);
    for (int n = threadIdx.x; n < N; n += 256) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) {
            const float v = dZ[(int64_t)b * ld + n];
            s1 += v;
            if (has_pert) s2 += v * sign_at(sout, b, n);
        }
        part[((int64_t)blockIdx.x * 2) * N + n] = s1;
        if (has_pert) part[((int64_t)blockIdx.x * 2 + 1) * N + n] = s2;
    }
}
__global__ __launch_bounds__(256) void k_bias_grad_sum(const float* __restrict__ part, int nchunk, int N, float* __restrict__ g_b, float* __restrict__ g_pert) {
    // 64 columns per workgroup, four waves each summing every fourth chunk (independent loads), then a fixed-order sum of the four
    __shared__ float sh[2][4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    float s1 = 0.f, s2 = 0.f;
    if (n < N) {
#pragma unroll 8
        for (int c = sub; c < nchunk; c += 4) { s1 += part[((int64_t)c * 2) * N + n]; if (g_pert) s2 += part[((int64_t)c * 2 + 1) * N + n]; }
    }
    sh[0][sub][threadIdx.x & 63] = s1; sh[1][sub][threadIdx.x & 63] = s2;
    __syncthreads();
    if (sub == 0 && n < N) {
        const int t = threadIdx.x;
        g_b[n] = ((sh[0][0][t] + sh[0][1][t]) + sh[0][2][t]) + sh[0][3][t];
        if (g_pert) g_pert[n] = ((sh[1][0][t] + sh[1][1][t]) + sh[1][2][t]) + sh[1][3][t];
    }
}
void launch_bias_grad(hipStream_t st, const float* dZ, int64_t ld, int B, int N, SignSpec sout, float* g_b, float* g_pert, float* scratch, int64_t scratch_floats) {
    const int nchunk = (B + 31) / 32;
    if (scratch && B >= 1024 && N <= 4096 && (int64_t)nchunk * 2 * N <= scratch_floats) {
        hipLaunchKernelGGL(k_bias_grad_rows, dim3(nchunk), dim3(256), 0, st, dZ, ld, B, N, sout, scratch, g_pert ? 1 : 0);
        hipLaunchKernelGGL(k_bias_grad_sum, dim3((N + 63) / 64), dim3(256), 0, st, scratch, nchunk, N, g_b, g_pert);
        return;
    }
    if (N <= 4096) hipLaunchKernelGGL(k_bias_grad_wave, dim3((N + 3) / 4), dim3(256), 0, st, dZ, ld, B, N, sout, g_b, g_pert);
    else hipLaunchKernelGGL(k_bias_grad, dim3((N + 255) / 256), dim3(256), 0, st, dZ, ld, B, N, sout, g_b, g_pert);
}

// =====================================================================================
// Negative samplers as index generators (src/mdl/fnn.py:48-76).  One thread per row.
// Sequential sampling without replacement == draw with replacement and reject members / repeats, which
// is the distribution of rand+topk (uniform) and of multinomial(replacement=False) (weighted).
// =====================================================================================
__device__ __forceinline__ bool is_member(const int32_t* mi, int npos, int c) {
    for (int p = 0; p < npos; ++p) if (mi[p] == c) return true;
    return false;
}
// The sampler threads are one dependent chain per row (rows -> indptr -> indices -> candidates): what they re-read - the row's positives and the picks
// made so far - stays in registers (statically indexed, so not in scratch); rows with more than 8 positives / ns > 8 read the rest from memory.
constexpr int NS_REG = 8;
struct RowSet {
    int pos[NS_REG], pick[NS_REG];
    const int32_t* mi; int npos; int64_t* o;
    __device__ __forceinline__ void init(const int32_t* mi_, int npos_, int64_t* o_) {
        mi = mi_; npos = npos_; o = o_;
#pragma unroll
        for (int p = 0; p < NS_REG; ++p) { pos[p] = p < npos ? mi[p] : -1; pick[p] = -1; }
    }
    __device__ __forceinline__ bool member(int c) const {
        bool b = false;
#pragma unroll
        for (int p = 0; p < NS_REG; ++p) b |= (pos[p] == c);      // unused slots hold -1, candidates are >= 0
        for (int p = NS_REG; p < npos && !b; ++p) b = (mi[p] == c);
        return b;
    }
    __device__ __forceinline__ bool picked(int q, int c) const {
        bool b = false;
#pragma unroll
        for (int p = 0; p < NS_REG; ++p) b |= (pick[p] == c);     // slots >= q still hold -1
        for (int p = NS_REG; p < q && !b; ++p) b = ((int)o[p] == c);
        return b;
    }
    __device__ __forceinline__ void put(int q, int c) {
#pragma unroll
        for (int p = 0; p < NS_REG; ++p) if (p == q) pick[p] = c;
        o[q] = c;
    }
};

__global__ void k_ns_uniform(const int64_t* __restrict__ rows, int B, int M, int ns, const int64_t* __restrict__ m_indptr,
                             const int32_t* __restrict__ m_indices, uint32_t k0, uint32_t k1, uint32_t step, uint32_t row0, int64_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int64_t team = rows[i];
    const int32_t* mi = m_indices + m_indptr[team];
    const int npos = (int)(m_indptr[team + 1] - m_indptr[team]);
    int64_t* o = out + (int64_t)i * ns;
    RowSet rs; rs.init(mi, npos, o);
    uint32_t ctr = 0;
    for (int q = 0; q < ns; ++q) {
        int pick = -1;
        for (int tries = 0; tries < 4096 && pick < 0; ++tries) {
            const uint4 r = philox4x32(make_uint4((uint32_t)i + row0, ctr++, step, 0x4e533031u), make_uint2(k0, k1));
            const uint32_t cand[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (int)__umulhi(cand[u], (uint32_t)M);
                if (pick < 0 && !rs.member(c) && !rs.picked(q, c)) pick = c;
            }
        }
        if (pick < 0) {  // fewer than ns negatives in the row: topk then returns positives (fnn.py:54); any unused column
            for (int c = 0; c < M && pick < 0; ++c) if (!rs.picked(q, c) && !rs.member(c)) pick = c;
            for (int c = 0; c < M && pick < 0; ++c) if (!rs.picked(q, c)) pick = c;
        }
        rs.put(q, pick);
    }
}
void launch_ns_uniform(hipStream_t st, const int64_t* rows, int B, int M, int ns, const int64_t* m_indptr, const int32_t* m_indices,
                       uint32_t k0, uint32_t k1, uint32_t step, uint32_t row0, int64_t* out) {
    hipLaunchKernelGGL(k_ns_uniform, dim3((B + 63) / 64), dim3(64), 0, st, rows, B, M, ns, m_indptr, m_indices, k0, k1, step, row0, out);
}

__global__ void k_ns_alias(const int64_t* __restrict__ rows, int B, int M, int ns, const int64_t* __restrict__ m_indptr,
                           const int32_t* __restrict__ m_indices, const float* __restrict__ prob, const int32_t* __restrict__ alias,
                           const double* __restrict__ weight, double total_weight, uint32_t k0, uint32_t k1, uint32_t step, uint32_t row0,
                           int64_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int64_t team = rows[i];
    const int32_t* mi = m_indices + m_indptr[team];
    const int npos = (int)(m_indptr[team + 1] - m_indptr[team]);
    int64_t* o = out + (int64_t)i * ns;
    RowSet rs; rs.init(mi, npos, o);
    double negw = total_weight;
    for (int p = 0; p < npos; ++p) negw -= weight[mi[p]];
    const bool fallback = !(negw > 1e-12 * total_weight);  // all sampling weight sits on the row's members (fnn.py:67-69)
    uint32_t ctr = 0;
    for (int q = 0; q < ns; ++q) {
        int pick = -1;
        for (int tries = 0; tries < 8192 && pick < 0; ++tries) {
            const uint4 r = philox4x32(make_uint4((uint32_t)i + row0, ctr++, step, 0x4e533032u), make_uint2(k0, k1));
            for (int u = 0; u < 2 && pick < 0; ++u) {
                const uint32_t a = u ? r.z : r.x, b = u ? r.w : r.y;
                int c = (int)__umulhi(a, (uint32_t)M);
                bool bad;
                if (fallback) bad = false;  // uniform over ALL columns, members included
                else {
                    if (u01(b) >= prob[c]) c = alias[c];
                    bad = rs.member(c) || !(weight[c] > 0.0);
                }
                if (!bad) bad = rs.picked(q, c);
                if (!bad) pick = c;
            }
        }
        if (pick < 0) {  // fewer than ns columns with weight: multinomial would raise; take any unused column
            for (int c = 0; c < M && pick < 0; ++c) if (!rs.picked(q, c)) pick = c;
        }
        rs.put(q, pick);
    }
}
void launch_ns_alias(hipStream_t st, const int64_t* rows, int B, int M, int ns, const int64_t* m_indptr, const int32_t* m_indices,
                     const float* prob, const int32_t* alias, const double* weight, double total_weight, uint32_t k0, uint32_t k1,
                     uint32_t step, uint32_t row0, int64_t* out) {
    hipLaunchKernelGGL(k_ns_alias, dim3((B + 63) / 64), dim3(64), 0, st, rows, B, M, ns, m_indptr, m_indices, prob, alias, weight,
                       total_weight, k0, k1, step, row0, out);
}

// unigram_b: the per-batch frequency table has support only on the experts of the batch (<= a few thousand of M): the alias
// table is built over that support; cols[] maps a support slot to its expert (sorted, so a row's positives are found by bisection).
__global__ void k_ns_alias_sparse(const int64_t* __restrict__ rows, int B, int M, int ns, const int64_t* __restrict__ m_indptr,
                                  const int32_t* __restrict__ m_indices, const int32_t* __restrict__ cols, const float* __restrict__ prob,
                                  const int32_t* __restrict__ alias, const float* __restrict__ weight, int nsup, double total_weight,
                                  uint32_t k0, uint32_t k1, uint32_t step, uint32_t row0, int64_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int64_t team = rows[i];
    const int32_t* mi = m_indices + m_indptr[team];
    const int npos = (int)(m_indptr[team + 1] - m_indptr[team]);
    int64_t* o = out + (int64_t)i * ns;
    RowSet rs; rs.init(mi, npos, o);
    double negw = total_weight;
    for (int p = 0; p < npos; ++p) {
        int lo = 0, hi = nsup;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cols[mid] < mi[p]) lo = mid + 1; else hi = mid; }
        if (lo < nsup && cols[lo] == mi[p]) negw -= (double)weight[lo];
    }
    const bool fallback = !(negw > 1e-9 * total_weight) || nsup == 0;  // every sampling weight sits on the row's members (fnn.py:67-69)
    uint32_t ctr = 0;
    for (int q = 0; q < ns; ++q) {
        int pick = -1;
        for (int tries = 0; tries < 8192 && pick < 0; ++tries) {
            const uint4 r = philox4x32(make_uint4((uint32_t)i + row0, ctr++, step, 0x4e533033u), make_uint2(k0, k1));
            for (int u = 0; u < 2 && pick < 0; ++u) {
                const uint32_t a = u ? r.z : r.x, b = u ? r.w : r.y;
                int c; bool bad;
                if (fallback) { c = (int)__umulhi(a, (uint32_t)M); bad = false; }  // uniform over ALL columns, members included
                else {
                    int sl = (int)__umulhi(a, (uint32_t)nsup);
                    if (u01(b) >= prob[sl]) sl = alias[sl];
                    c = cols[sl];
                    bad = rs.member(c) || !(weight[sl] > 0.f);
                }
                if (!bad) bad = rs.picked(q, c);
                if (!bad) pick = c;
            }
        }
        if (pick < 0) {  // fewer than ns admissible experts: take any unused column (multinomial would raise)
            for (int c = 0; c < M && pick < 0; ++c) if (!rs.picked(q, c)) pick = c;
        }
        rs.put(q, pick);
    }
}
void launch_ns_alias_sparse(hipStream_t st, const int64_t* rows, int B, int M, int ns, const int64_t* m_indptr, const int32_t* m_indices,
                            const int32_t* cols, const float* prob, const int32_t* alias, const float* weight, int nsup, double total_weight,
                            uint32_t k0, uint32_t k1, uint32_t step, uint32_t row0, int64_t* out) {
    hipLaunchKernelGGL(k_ns_alias_sparse, dim3((B + 63) / 64), dim3(64), 0, st, rows, B, M, ns, m_indptr, m_indices, cols, prob, alias, weight,
                       nsup, total_weight, k0, k1, step, row0, out);
}

// =====================================================================================
// Adam (torch.optim.Adam defaults, src/mdl/fnn.py:104,139) over the flat parameter buffer
// =====================================================================================
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_over_bc1, float b1, float b2, float eps, float bc2_sqrt) {
    adam_step(p, g, m, v, lr_over_bc1, b1, b2, eps, __builtin_amdgcn_rcpf(bc2_sqrt));   // (ntf_device.h; the reciprocal of the uniform bc2_sqrt is hoisted out of the callers' loops)
}
// 16-byte accesses (the segments of the flat buffers are 256-byte aligned and a multiple of 4 floats long up to a scalar tail): a quarter of
// the memory instructions, which matters when this kernel runs on the side stream beside the dW kernel and competes for issue slots
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                       float lr_over_bc1, float b1, float b2, float eps, float bc2_sqrt, int vec) {
    const int64_t nq = vec ? (n >> 2) : 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[q], mm = reinterpret_cast<float4*>(m)[q], vv = reinterpret_cast<float4*>(v)[q];
        const float4 gg = reinterpret_cast<const float4*>(g)[q];
        adam_one(pp.x, gg.x, mm.x, vv.x, lr_over_bc1, b1, b2, eps, bc2_sqrt); adam_one(pp.y, gg.y, mm.y, vv.y, lr_over_bc1, b1, b2, eps, bc2_sqrt);
        adam_one(pp.z, gg.z, mm.z, vv.z, lr_over_bc1, b1, b2, eps, bc2_sqrt); adam_one(pp.w, gg.w, mm.w, vv.w, lr_over_bc1, b1, b2, eps, bc2_sqrt);
        reinterpret_cast<float4*>(p)[q] = pp; reinterpret_cast<float4*>(m)[q] = mm; reinterpret_cast<float4*>(v)[q] = vv;
    }
    for (int64_t e = (nq << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float pe = p[e], me = m[e], ve = v[e];
        adam_one(pe, g[e], me, ve, lr_over_bc1, b1, b2, eps, bc2_sqrt);
        p[e] = pe; m[e] = me; v[e] = ve;
    }
}
void launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                 float bc1, float bc2_sqrt) {
    if (n <= 0) return;
    const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    const int blocks = (int)std::min<int64_t>(((aligned ? n / 4 : n) + 255) / 256 + 1, 256 * 8);   // unaligned: never the case for the engine's segments
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, lr / bc1, b1, b2, eps, bc2_sqrt, aligned ? 1 : 0);
}

// per-step scalars behind d_kl (ntf_engine.hip): [0] the step's KL sum (double), int32 view [2] its fp16x3 range flag, [3] the fallback counter (kept);
// [2] (double) and int32 [6]: the same two for the NEXT step, written by the dW + Adam epilogue when it also produced that step's operands (FusedDw.produce)
__global__ void k_step_scalars(double* kl, int take_next, const double* start_from) {
    int32_t* w = reinterpret_cast<int32_t*>(kl);
    kl[0] = take_next ? kl[2] : (start_from ? *start_from : 0.0);
    w[2] = take_next ? w[6] : 0;
    kl[2] = 0.0; w[6] = 0;
}
void launch_step_scalars(hipStream_t st, double* kl, int take_next, const double* start_from) { hipLaunchKernelGGL(k_step_scalars, dim3(1), dim3(1), 0, st, kl, take_next, start_from); }

// the same update over up to four [lo, lo + n) ranges of the flat buffers in ONE launch (the rest of the model beside the dW kernel's in-epilogue
// Adam: hidden layers, biases, rho biases - three short ranges, three dependent launches before).  Two more small launches ride here:
//   fin[k] = 1 / 2: the range is the output layer's bias / rho_bias, whose raw gradients (sums of dz, of dz * s_out: the dW kernel's by-product) still need the
//                   Flipout chain rule and the KL terms - k_flipout_grad_finalize's arithmetic, applied (and written back to G) before the update;
//   rotate != null: thread 0 moves the NEXT step's KL sum and range flag, written by this step's dW epilogue (FusedDw.produce), to the current slots.
//   nx.bp != null (round 5; the ranges with fin = 1 / 2): the launch also is the operand producer of the NEXT step's output bias - from the updated rho_b' the range with
//                   fin = 2 writes bp' = softplus(rho_b') eps_b' (nx.eps: the generator of step + 1) and both ranges add their parts of KL(mu_b', rho_b') * nx.klw to
//                   the NEXT step's KL slot (k_head's bias workgroups did this in a launch of their own behind this one).  The rotation then cannot be thread 0's first
//                   act - adds to the next slot are still coming: the workgroup that finishes LAST (a ticket in the int32 behind the next flag) rotates.
struct AdamRanges { int64_t lo[4], n[4]; int blk0[5]; int cnt; int fin[4]; NormalSpec eps; float klw; double* rotate;
                    struct { float* bp; NormalSpec eps; double klw; } nx;
                    // the f32 copy of the NEXT step's sigma * eps - read only by a step that falls back to the exact-f32 kernels (launch_flipout_perturb(only_if) otherwise issues it as
                    // a launch of its own in front of that step): workgroups [blk0, blk0 + nblk) of THIS launch, which leave at once unless *only_if (the next step's range flag,
                    // complete by now: the dW epilogue and the prefetched head are behind this launch) is raised.  n == 0: no such job
                    struct { const float* rho; float* out; int64_t n; NormalSpec eps; const int* only_if; int blk0, nblk; } f32c; };
__device__ __forceinline__ void fin_mu(float& g, float p, float klw) { g += klw * p; }
__device__ __forceinline__ void fin_rho(float& g, float r, float z, float klw) {
    const float sg = 1.f / (1.f + expf(-r));  // d softplus / d rho
    const float sigma = softplus_rho(r);
    g = g * z * sg + klw * (sigma - 1.f / sigma) * sg;
}
__device__ __forceinline__ void rotate_scalars(double* kl) {      // next step's KL sum and range flag -> current (see k_step_scalars)
    int32_t* w = reinterpret_cast<int32_t*>(kl);
    kl[0] = kl[2]; w[2] = w[6]; kl[2] = 0.0; w[6] = 0;
}
__global__ void k_adam_ranges(float* __restrict__ P, float* __restrict__ G, float* __restrict__ M1, float* __restrict__ V2, AdamRanges r,
                              float lr_over_bc1, float b1, float b2, float eps, float bc2_sqrt) {
    const bool produce = r.nx.bp != nullptr;
    if (r.rotate && !produce && blockIdx.x == 0 && threadIdx.x == 0) rotate_scalars(r.rotate);
    const bool f32c_wg = r.f32c.n > 0 && (int)blockIdx.x >= r.f32c.blk0;
    if (f32c_wg && __builtin_nontemporal_load(r.f32c.only_if) != 0) {      // k_flipout_perturb's arithmetic, f32 output only
        const int64_t quads = (r.f32c.n + 3) / 4;
        for (int64_t q = (int64_t)((int)blockIdx.x - r.f32c.blk0) * blockDim.x + threadIdx.x; q < quads; q += (int64_t)r.f32c.nblk * blockDim.x) {
            const int64_t e0 = q * 4;
            float z[4];
            normal4(r.f32c.eps, q, e0, r.f32c.n, z);
            if (e0 + 3 < r.f32c.n) {
                const float4 r4 = *reinterpret_cast<const float4*>(r.f32c.rho + e0);
                float ls;
                *reinterpret_cast<float4*>(r.f32c.out + e0) = make_float4(softplus_rho_fast(r4.x, ls) * z[0], softplus_rho_fast(r4.y, ls) * z[1], softplus_rho_fast(r4.z, ls) * z[2], softplus_rho_fast(r4.w, ls) * z[3]);
            } else for (int j = 0; j < 4 && e0 + j < r.f32c.n; ++j) r.f32c.out[e0 + j] = softplus_rho(r.f32c.rho[e0 + j]) * z[j];
        }
    }
    int k = 0;
    while (k + 1 < r.cnt && (int)blockIdx.x >= r.blk0[k + 1]) ++k;
    const int64_t lo = r.lo[k], n = f32c_wg ? 0 : r.n[k];
    const int fin = r.fin[k];
    const int64_t first = (int64_t)((int)blockIdx.x - r.blk0[k]) * blockDim.x + threadIdx.x, stride = (int64_t)(r.blk0[k + 1] - r.blk0[k]) * blockDim.x;
    float *p = P + lo, *m = M1 + lo, *v = V2 + lo, *g = G + lo;
    const int64_t nq = n >> 2;    // lo is a multiple of 4 floats (checked by the launcher): 16-byte accesses
    float kl_nx = 0.f;
    for (int64_t q = first; q < nq; q += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[q], mm = reinterpret_cast<float4*>(m)[q], vv = reinterpret_cast<float4*>(v)[q];
        float4 gg = reinterpret_cast<const float4*>(g)[q];
        if (fin == 1) { fin_mu(gg.x, pp.x, r.klw); fin_mu(gg.y, pp.y, r.klw); fin_mu(gg.z, pp.z, r.klw); fin_mu(gg.w, pp.w, r.klw); reinterpret_cast<float4*>(g)[q] = gg; }
        else if (fin == 2) {
            float z[4];
            normal4(r.eps, q, q * 4, n, z);
            fin_rho(gg.x, pp.x, z[0], r.klw); fin_rho(gg.y, pp.y, z[1], r.klw); fin_rho(gg.z, pp.z, z[2], r.klw); fin_rho(gg.w, pp.w, z[3], r.klw);
            reinterpret_cast<float4*>(g)[q] = gg;
        }
        adam_one(pp.x, gg.x, mm.x, vv.x, lr_over_bc1, b1, b2, eps, bc2_sqrt); adam_one(pp.y, gg.y, mm.y, vv.y, lr_over_bc1, b1, b2, eps, bc2_sqrt);
        adam_one(pp.z, gg.z, mm.z, vv.z, lr_over_bc1, b1, b2, eps, bc2_sqrt); adam_one(pp.w, gg.w, mm.w, vv.w, lr_over_bc1, b1, b2, eps, bc2_sqrt);
        reinterpret_cast<float4*>(p)[q] = pp; reinterpret_cast<float4*>(m)[q] = mm; reinterpret_cast<float4*>(v)[q] = vv;
        if (produce && fin == 1) kl_nx += 0.5f * (pp.x * pp.x) + 0.5f * (pp.y * pp.y) + 0.5f * (pp.z * pp.z) + 0.5f * (pp.w * pp.w);
        else if (produce && fin == 2) {
            float z[4], ov[4];
            const float rv[4] = {pp.x, pp.y, pp.z, pp.w};
            normal4(r.nx.eps, q, q * 4, n, z);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float ls;
                const float sigma = softplus_rho_fast(rv[j], ls);
                ov[j] = sigma * z[j];
                kl_nx += -ls + 0.5f * (sigma * sigma) - 0.5f;
            }
            *reinterpret_cast<float4*>(r.nx.bp + q * 4) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        }
    }
    for (int64_t e = (nq << 2) + first; e < n; e += stride) {
        float pe = p[e], me = m[e], ve = v[e], ge = g[e];
        if (fin == 1) { fin_mu(ge, pe, r.klw); g[e] = ge; }
        else if (fin == 2) {
            float z[4];
            normal4(r.eps, e >> 2, e & ~(int64_t)3, n, z);
            const int j = (int)(e & 3);
            fin_rho(ge, pe, j == 0 ? z[0] : j == 1 ? z[1] : j == 2 ? z[2] : z[3], r.klw); g[e] = ge;
        }
        adam_one(pe, ge, me, ve, lr_over_bc1, b1, b2, eps, bc2_sqrt);
        p[e] = pe; m[e] = me; v[e] = ve;
        if (produce && fin == 1) kl_nx += 0.5f * (pe * pe);
        else if (produce && fin == 2) {
            float z[4];
            normal4(r.nx.eps, e >> 2, e & ~(int64_t)3, n, z);
            const int j = (int)(e & 3);
            const float sigma = softplus_rho(pe);
            r.nx.bp[e] = sigma * (j == 0 ? z[0] : j == 1 ? z[1] : j == 2 ? z[2] : z[3]);
            kl_nx += -logf(sigma) + 0.5f * (sigma * sigma) - 0.5f;
        }
    }
    if (produce) {      // (uniform over the launch: every workgroup takes a ticket)
        __shared__ double red[4];
        __shared__ int last;
        const double sred = wave_reduce_sum_d((double)kl_nx);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sred;
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t* w = reinterpret_cast<int32_t*>(r.rotate);
            if (fin && !f32c_wg) atomicAdd(r.rotate + 2, (red[0] + red[1] + red[2] + red[3]) * r.nx.klw);
            __threadfence();                                        // the add (a device-scope atomic) has been performed before the ticket is taken
            last = atomicAdd(w + 7, 1) == (int)gridDim.x - 1;
            if (last) {
                // every workgroup's add is in: take the next slot's value through the atomic unit (a plain load could read a stale line of this XCD's L2) and rotate
                const unsigned long long bits = atomicExch(reinterpret_cast<unsigned long long*>(r.rotate + 2), 0ull);
                r.rotate[0] = __longlong_as_double((long long)bits);
                w[2] = atomicExch(w + 6, 0);
                atomicExch(w + 7, 0);
            }
        }
    }
}
// lo_hi: n pairs [lo, hi) of float offsets into the flat buffers (P, G, M1, V2 are the buffers' bases, 16-byte aligned); fin (nullable): per range 0 / 1 / 2, see above
void launch_adam_ranges(hipStream_t st, float* P, float* G, float* M1, float* V2, const int64_t* lo_hi, int n, float lr, float b1, float b2,
                        float eps, float bc1, float bc2_sqrt, const int* fin, const NormalSpec* fin_eps, float fin_klw, double* rotate,
                        float* nx_bp, const NormalSpec* nx_eps, double nx_klw, const F32CopyJob* f32c) {
    int k = 0;
    bool launched = false;
    while (k < n) {
        AdamRanges r; r.cnt = 0; int blocks = 0;
        r.klw = fin_klw; if (fin_eps) r.eps = *fin_eps; r.rotate = launched ? nullptr : rotate;
        r.nx.bp = (rotate && nx_bp && nx_eps && n <= 4) ? nx_bp : nullptr; if (nx_eps) r.nx.eps = *nx_eps; r.nx.klw = nx_klw;      // (one launch holds every range: the ticket counts its workgroups)
        for (; k < n && r.cnt < 4; ++k) {
            const int64_t lo = lo_hi[2 * k], cnt = lo_hi[2 * k + 1] - lo;
            if (cnt <= 0) continue;
            if ((lo & 3) && !(fin && fin[k])) {   // not 16-byte aligned (never the case for the engine's segments): its own plain launch
                launch_adam(st, P + lo, G + lo, M1 + lo, V2 + lo, cnt, lr, b1, b2, eps, bc1, bc2_sqrt);
                continue;
            }
            r.lo[r.cnt] = lo; r.n[r.cnt] = cnt; r.blk0[r.cnt] = blocks; r.fin[r.cnt] = fin ? fin[k] : 0;
            blocks += (int)std::min<int64_t>((cnt / 4 + 255) / 256 + 1, 256 * 8);
            r.cnt += 1;
        }
        if (r.cnt == 0) continue;
        for (int j = r.cnt; j <= 4; ++j) r.blk0[j] = blocks;
        for (int j = r.cnt; j < 4; ++j) { r.lo[j] = 0; r.n[j] = 0; r.fin[j] = 0; }
        r.f32c.n = 0;
        if (f32c && f32c->n > 0 && r.nx.bp && !launched) {      // (with the ticketed launch only: its last workgroup rotates the flag this job reads)
            r.f32c.rho = f32c->rho; r.f32c.out = f32c->out; r.f32c.n = f32c->n; r.f32c.eps = f32c->eps; r.f32c.only_if = f32c->only_if;
            r.f32c.blk0 = blocks; r.f32c.nblk = 64; blocks += 64;      // (few: each takes a ticket; the copy itself is made in all but never)
        }
        hipLaunchKernelGGL(k_adam_ranges, dim3(blocks), dim3(256), 0, st, P, G, M1, V2, r, lr / bc1, b1, b2, eps, bc2_sqrt);
        launched = true;
    }
    if (!launched && rotate) launch_step_scalars(st, rotate, 1);
}

// ---- launch_flipout_sweep (ntf_kernels.h): finalize + Adam + next-step operand of one Flipout weight tensor in one pass
// fin_rho on the hardware transcendentals, as the output layer's dW epilogue takes it (ntf_fused_dw.hip dw_finish_ops): sigma by the series below e = 2^-6, log(1 + e) above
__device__ __forceinline__ void fin_rho_fast(float& g, float r, float z, float klw) {
    const float e = __builtin_amdgcn_exp2f(fminf(r, 80.f) * 1.44269504f), t = 1.f + e;
    const float sigma = e < 0.015625f ? e * (1.f - e * (0.5f - e * (0.33333333f - 0.25f * e))) : __builtin_amdgcn_logf(t) * 0.69314718f;
    const float sg = e * __builtin_amdgcn_rcpf(t), isig = __builtin_amdgcn_rcpf(sigma);
    g = g * z * sg + klw * (sigma - isig) * sg;
}
__global__ __launch_bounds__(256) void k_flipout_sweep(FlipoutSweep a, float lr_over_bc1) {
    const int64_t quads = a.n >> 2;
    float kl = 0.f;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = q * 4;
        float4 pm = reinterpret_cast<float4*>(a.mu)[q], pr = reinterpret_cast<float4*>(a.rho)[q];
        float4 m1 = reinterpret_cast<float4*>(a.m_mu)[q], v1 = reinterpret_cast<float4*>(a.v_mu)[q], m2 = reinterpret_cast<float4*>(a.m_rho)[q], v2 = reinterpret_cast<float4*>(a.v_rho)[q];
        const bool hp2 = (a.H & (a.H - 1)) == 0;
        const bool t = a.touched ? a.touched[hp2 ? (uint64_t)e0 >> (31 - __builtin_clz(a.H)) : (uint64_t)e0 / (uint32_t)a.H] != 0 : true;      // (the quad lies in one row: H % 4 == 0)
        float4 gm = zero4, gr = zero4;
        if (t) {
            gm = reinterpret_cast<float4*>(a.g_mu)[q]; gr = reinterpret_cast<float4*>(a.g_rho)[q];
            reinterpret_cast<float4*>(a.g_mu)[q] = zero4; reinterpret_cast<float4*>(a.g_rho)[q] = zero4;
        }
        float z[4];
        normal4(a.eps, q, e0, a.n, z);
        fin_mu(gm.x, pm.x, a.klw); fin_mu(gm.y, pm.y, a.klw); fin_mu(gm.z, pm.z, a.klw); fin_mu(gm.w, pm.w, a.klw);
        fin_rho_fast(gr.x, pr.x, z[0], a.klw); fin_rho_fast(gr.y, pr.y, z[1], a.klw); fin_rho_fast(gr.z, pr.z, z[2], a.klw); fin_rho_fast(gr.w, pr.w, z[3], a.klw);
        adam_one(pm.x, gm.x, m1.x, v1.x, lr_over_bc1, a.b1, a.b2, a.adam_eps, a.bc2_sqrt); adam_one(pm.y, gm.y, m1.y, v1.y, lr_over_bc1, a.b1, a.b2, a.adam_eps, a.bc2_sqrt);
        adam_one(pm.z, gm.z, m1.z, v1.z, lr_over_bc1, a.b1, a.b2, a.adam_eps, a.bc2_sqrt); adam_one(pm.w, gm.w, m1.w, v1.w, lr_over_bc1, a.b1, a.b2, a.adam_eps, a.bc2_sqrt);
        adam_one(pr.x, gr.x, m2.x, v2.x, lr_over_bc1, a.b1, a.b2, a.adam_eps, a.bc2_sqrt); adam_one(pr.y, gr.y, m2.y, v2.y, lr_over_bc1, a.b1, a.b2, a.adam_eps, a.bc2_sqrt);
        adam_one(pr.z, gr.z, m2.z, v2.z, lr_over_bc1, a.b1, a.b2, a.adam_eps, a.bc2_sqrt); adam_one(pr.w, gr.w, m2.w, v2.w, lr_over_bc1, a.b1, a.b2, a.adam_eps, a.bc2_sqrt);
        reinterpret_cast<float4*>(a.mu)[q] = pm; reinterpret_cast<float4*>(a.rho)[q] = pr;
        reinterpret_cast<float4*>(a.m_mu)[q] = m1; reinterpret_cast<float4*>(a.v_mu)[q] = v1; reinterpret_cast<float4*>(a.m_rho)[q] = m2; reinterpret_cast<float4*>(a.v_rho)[q] = v2;
        // the next step's operand from the updated parameters (k_flipout_perturb's arithmetic)
        normal4(a.nx_eps, q, e0, a.n, z);
        const float rv[4] = {pr.x, pr.y, pr.z, pr.w}, mv[4] = {pm.x, pm.y, pm.z, pm.w};
        float ov[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float ls;
            const float sigma = softplus_rho_fast(rv[j], ls);
            ov[j] = sigma * z[j];
            kl += -ls + 0.5f * (sigma * sigma + mv[j] * mv[j]) - 0.5f;
        }
        reinterpret_cast<float4*>(a.nx_wp)[q] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
    const double s = block_reduce_sum_d((double)kl);
    if (threadIdx.x == 0 && a.nx_kl) atomicAdd(a.nx_kl, s * a.nx_klw);
}
void launch_flipout_sweep(hipStream_t st, const FlipoutSweep& a) {
    if (a.n <= 0) return;
    // (grid size measured beside the dW kernel at config 3, two interleaved rounds: 2048 blocks 1.489 / 1.439 ms a step, 1024 1.446 / 1.443, 512 1.433 / 1.438, 256 1.446 / 1.448 -
    //  inside the noise: the pass and the HBM-bound kernel beside it share the same ~4.4 TB/s whatever the split)
    const int blocks = (int)std::min<int64_t>(((a.n >> 2) + 255) / 256, 2048);
    hipLaunchKernelGGL(k_flipout_sweep, dim3(blocks), dim3(256), 0, st, a, a.lr / a.bc1);
}

__global__ void k_fill(float* p, int64_t n, float v) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = v;
}
void launch_fill(hipStream_t st, float* p, int64_t n, float v) {
    if (n <= 0) return;
    int blocks = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, st, p, n, v);
}

// =====================================================================================
// Inference helpers (src/mdl/fnn.py:200-211): sigmoid, MC mean, predictive entropy / mutual information
// =====================================================================================
__global__ __launch_bounds__(256) void k_sigmoid_acc(const float* __restrict__ Act, int M, float scale, int accumulate,
                                                     float* __restrict__ out, float* __restrict__ ent_rows) {
    const int64_t i = blockIdx.y;
    float ent = 0.f;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < M; c += gridDim.x * 256) {
        const float l = Act[i * M + c];
        const float p = 1.f / (1.f + expf(-l));
        const float prev = accumulate ? out[i * M + c] : 0.f;
        out[i * M + c] = prev + p * scale;
        ent -= p * logf(p + 1e-15f);
    }
    if (ent_rows) {
        ent = block_reduce_sum(ent);
        if (threadIdx.x == 0) atomicAdd(&ent_rows[i], ent * scale);
    }
}
void launch_sigmoid_acc(hipStream_t st, const float* Act, int64_t n_rows, int M, float scale, int accumulate, float* out, float* ent_rows) {
    int bx = min((M + 255) / 256, 64);
    hipLaunchKernelGGL(k_sigmoid_acc, dim3(bx, (unsigned)n_rows), dim3(256), 0, st, Act, M, scale, accumulate, out, ent_rows);
}
__global__ __launch_bounds__(256) void k_row_entropy(const float* __restrict__ P, int M, float* __restrict__ ent) {
    const int64_t i = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < M; c += 256) { const float p = P[i * M + c]; s -= p * logf(p + 1e-15f); }
    s = block_reduce_sum(s);
    if (threadIdx.x == 0) ent[i] = s;
}
void launch_row_entropy(hipStream_t st, const float* P, int n_rows, int M, float* ent) {
    hipLaunchKernelGGL(k_row_entropy, dim3(n_rows), dim3(256), 0, st, P, M, ent);
}

// =====================================================================================
// Row-wise top-K of the probabilities (src/pkgmgr.py:125-134 topk): per row a 3-pass radix select on the
// f32 bit pattern (probabilities are positive, so the unsigned order is the value order) finds the K-th
// value, then the row is compacted and the K survivors are sorted (descending value, ascending index for
// ties) by a bitonic sort in LDS.  One workgroup per row.
// =====================================================================================
constexpr int TK_MAX = 2048;
constexpr int TK_SAMPLE = 4096;
size_t topk_workspace_bytes(int, int, int) { return 0; }

// descending bitonic sort of n2 (power of two) 64-bit keys in LDS by the whole workgroup
__device__ __forceinline__ void bitonic_desc(unsigned long long* keys, int n2, int tid, int nthreads) {
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < n2 / 2; t += nthreads) {
                const int lo = (t / stride) * stride * 2 + (t % stride), hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
}

// Probabilities of one row cluster in a few exponent bins, so an MSB-first radix histogram serialises on LDS atomics (measured: 19 ms
// per 1000 rows of 233 629).  Instead: (1) a strided sample of TK_SAMPLE values is sorted in LDS and its r-th largest taken as a
// threshold, r chosen so that ~4.5 K values of the row are expected above it; (2) one pass collects the values above the threshold
// (a few hundred appends); (3) if at least K and at most TK_MAX were collected, the top K of the row are among them — every value >= the
// K-th largest is above the threshold — and a bitonic sort on (value desc, column asc) finishes.  Otherwise (ties, tiny rows, unlucky
// sample) the exact radix selection below runs.  The result is the same deterministic ranking either way.
__global__ __launch_bounds__(1024) void k_topk_rows(const float* __restrict__ P, int M, int K, float* __restrict__ vals, int32_t* __restrict__ idx) {
    __shared__ unsigned long long keys[TK_SAMPLE];      // sample sort, then the candidates (TK_MAX <= TK_SAMPLE)
    __shared__ unsigned hist[2048];
    __shared__ unsigned s_prefix, s_remaining, s_count;
    const int64_t i = blockIdx.x;
    const float* row = P + i * M;
    const int tid = threadIdx.x, NT = blockDim.x;
    int n2 = 1; while (n2 < K) n2 <<= 1;
    bool done = false;
    if (M >= 8 * TK_SAMPLE && K * 8 <= TK_MAX) {
        const int stride = M / TK_SAMPLE;
        for (int k = tid; k < TK_SAMPLE; k += NT) keys[k] = (unsigned long long)__float_as_uint(row[(int64_t)k * stride]);   // probabilities: non-negative, bit order = value order
        __syncthreads();
        bitonic_desc(keys, TK_SAMPLE, tid, NT);
        // expected number of row values above the r-th largest sample value: r * M / TK_SAMPLE; aim at 4.5 K (>= K with overwhelming probability)
        int r = (int)((4.5 * K * TK_SAMPLE) / M) + 1;
        if (r < 8) r = 8;
        const unsigned thr = (unsigned)keys[r < TK_SAMPLE ? r : TK_SAMPLE - 1];
        __syncthreads();
        if (tid == 0) s_count = 0;
        __syncthreads();
        for (int c = tid; c < M; c += NT) {
            const unsigned u = __float_as_uint(row[c]);
            if (u > thr) { const unsigned p = atomicAdd(&s_count, 1u); if (p < TK_MAX) keys[p] = ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - c); }
        }
        __syncthreads();
        const unsigned cnt = s_count;
        if (cnt >= (unsigned)K && cnt <= TK_MAX) {
            int m2 = n2; while (m2 < (int)cnt) m2 <<= 1;
            for (int k = cnt + tid; k < m2; k += NT) keys[k] = 0ull;
            __syncthreads();
            bitonic_desc(keys, m2, tid, NT);
            done = true;
        }
        __syncthreads();
    }
    if (!done) {
        unsigned prefix = 0, prefix_mask = 0, remaining = K;
        // bits [31:21], [20:10], [9:0]
        const int shifts[3] = {21, 10, 0};
        const int widths[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            const int sh = shifts[pass], nb = 1 << widths[pass];
            for (int b = tid; b < nb; b += NT) hist[b] = 0;
            __syncthreads();
            for (int c = tid; c < M; c += NT) {
                const unsigned u = __float_as_uint(row[c]);
                if ((u & prefix_mask) == prefix) atomicAdd(&hist[(u >> sh) & (nb - 1)], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned rem = remaining; int b = nb - 1;
                for (; b > 0; --b) { if (hist[b] >= rem) break; rem -= hist[b]; }
                s_prefix = prefix | ((unsigned)b << sh); s_remaining = rem;
            }
            __syncthreads();
            prefix = s_prefix; remaining = s_remaining;
            prefix_mask |= (unsigned)(nb - 1) << sh;
            __syncthreads();
        }
        // prefix = bit pattern of the K-th largest value; `remaining` = how many copies of it belong to the top K
        if (tid == 0) s_count = 0;
        for (int k = tid; k < TK_MAX; k += NT) keys[k] = 0ull;
        __syncthreads();
        const unsigned thr = prefix;
        for (int c = tid; c < M; c += NT) {
            const unsigned u = __float_as_uint(row[c]);
            if (u > thr) { const unsigned p = atomicAdd(&s_count, 1u); keys[p] = ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - c); }
        }
        __syncthreads();
        // ties at the threshold: smallest column ids first (deterministic)
        if (tid == 0) {
            unsigned need = remaining, p = s_count;
            for (int c = 0; c < M && need > 0; ++c)
                if (__float_as_uint(row[c]) == thr) { keys[p++] = ((unsigned long long)thr << 32) | (unsigned)(0x7fffffff - c); --need; }
            s_count = p;
        }
        __syncthreads();
        bitonic_desc(keys, n2, tid, NT);
    }
    for (int k = tid; k < K; k += NT) {
        vals[i * K + k] = __uint_as_float((unsigned)(keys[k] >> 32));
        idx[i * K + k] = 0x7fffffff - (int)(keys[k] & 0xffffffffu);
    }
}
void launch_topk_rows(hipStream_t st, const float* P, int n_rows, int M, int K, float* vals, int32_t* idx, void*) {
    hipLaunchKernelGGL(k_topk_rows, dim3(n_rows), dim3(1024), 0, st, P, M, K, vals, idx);
}

// =====================================================================================
// generator test hooks
// =====================================================================================
__global__ void k_fill_normal(NormalSpec s, int64_t n, float* out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e0 = q * 4;
    if (e0 >= n) return;
    float z[4];
    normal4(s, q, e0, n, z);
    for (int j = 0; j < 4; ++j) if (e0 + j < n) out[e0 + j] = z[j];
}
void launch_fill_normal(hipStream_t st, NormalSpec s, int64_t n, float* out) {
    const int64_t quads = (n + 3) / 4;
    hipLaunchKernelGGL(k_fill_normal, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, s, n, out);
}
__global__ void k_fill_sign(SignSpec s, int rows, int cols, float* out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)rows * cols) return;
    out[e] = sign_at(s, e / cols, e % cols);
}
void launch_fill_sign(hipStream_t st, SignSpec s, int rows, int cols, float* out) {
    const int64_t n = (int64_t)rows * cols;
    hipLaunchKernelGGL(k_fill_sign, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, rows, cols, out);
}

}  // namespace ntf
