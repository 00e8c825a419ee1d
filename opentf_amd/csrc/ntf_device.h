// Device-side helpers shared by the gfx950 kernels: counter-based generators, sign hash, BCE terms,
// wave64 / workgroup reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ntf_kernels.h"

namespace ntf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

// +1/-1 at (row, col) of a sign tensor: bayesian-torch draws uniform_(-1,1).sign(); here a two-stage
// murmur finaliser of (row, col) under a per-(step, layer, tensor) key, or the injected array.
// One 32-bit word carries the signs of 32 consecutive columns of a row (bit c&31 of word(r, c>>5)), so that the
// fused kernels can fetch / regenerate 32 signs at a time; every path derives its signs from this definition.
__device__ __forceinline__ uint32_t sign_word(uint32_t k0, uint32_t k1, uint32_t r, uint32_t cb) {
    uint32_t h = fmix32(r * 0x9E3779B1u + k0);
    return fmix32(h ^ (cb * 0x85EBCA77u + k1));
}
__device__ __forceinline__ float sign_hash(uint32_t k0, uint32_t k1, uint32_t r, uint32_t c) {
    return ((sign_word(k0, k1, r, c >> 5) >> (c & 31)) & 1u) ? -1.0f : 1.0f;
}
__device__ __forceinline__ float sign_at(const SignSpec& s, int64_t r, int64_t c) {
    if (s.inj) return s.inj[r * s.ld + c];
    return sign_hash(s.k0, s.k1, (uint32_t)r, (uint32_t)c);
}

// Philox4x32-10 (Salmon et al., SC'11)
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // in (0,1)

// four N(0,1) values for elements 4q .. 4q+3 of a tensor
__device__ __forceinline__ void normal4(const NormalSpec& s, int64_t q, int64_t e0, int64_t n, float z[4]) {
    if (s.inj) {
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = (e0 + j < n) ? s.inj[e0 + j] : 0.f;
        return;
    }
    q += s.qbase;
    const uint4 r = philox4x32(make_uint4((uint32_t)q, (uint32_t)(q >> 32), s.tag, s.step), make_uint2(s.k0, s.k1));
    // Box-Muller on the hardware transcendentals: v_log_f32 is log2, v_sin/v_cos take their argument in revolutions
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.x)));   // sqrt(-2 ln u) = sqrt(-2 ln2 log2 u)
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.z)));
    const float a0 = u01(r.y), a1 = u01(r.w);
    z[0] = r0 * __builtin_amdgcn_cosf(a0); z[1] = r0 * __builtin_amdgcn_sinf(a0);
    z[2] = r1 * __builtin_amdgcn_cosf(a1); z[3] = r1 * __builtin_amdgcn_sinf(a1);
}

// two f32 -> three packed bf16 pairs (element 0 in the low half)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(x0), "v"(x1));
    const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xFFFF0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(r0), "v"(r1));
    const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xFFFF0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p3) : "v"(s0), "v"(s1));
}
// two f32 -> two packed fp16 pairs, x = x1 + x2 to 22 bits (both conversions round to nearest; fp16 subnormals are honoured by the MFMA)
__device__ __forceinline__ void split_pair_h(float x0, float x1, uint32_t& p1, uint32_t& p2) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t x = {x0, x1};
    const h2_t a1 = __builtin_convertvector(x, h2_t);
    const f2_t r = x - __builtin_convertvector(a1, f2_t);
    const h2_t a2 = __builtin_convertvector(r, h2_t);
    p1 = __builtin_bit_cast(uint32_t, a1); p2 = __builtin_bit_cast(uint32_t, a2);
}
// NP = 3: bf16 three-way split (bf16x6 products);  NP = 2: fp16 two-way split of x * scale (fp16x3 products), scale an exact power of two
template <int NP> __device__ __forceinline__ void split_pair_np(float x0, float x1, float scale, uint32_t (&p)[3]) {
    if (NP == 3) split_pair(x0, x1, p[0], p[1], p[2]);
    else {   // saturate at the fp16 range instead of producing inf (weights beyond +-255, activations beyond +-4094: far outside any trained model)
        split_pair_h(__builtin_amdgcn_fmed3f(x0 * scale, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(x1 * scale, -65504.f, 65504.f), p[0], p[1]); p[2] = 0u;
    }
}
// split planes of a row-major [M, H] f32 matrix for the split-product forward kernel: [tile of 32 rows][plane 0..NP-1][row in tile][H];
// stores the planes of the element pair (row, j), (row, j+1), j even
template <int NP> __device__ __forceinline__ void planes_store_pair(uint16_t* planes, int64_t row, int j, int H, float x0, float x1, float scale) {
    uint32_t p[3];
    split_pair_np<NP>(x0, x1, scale, p);
    uint32_t* o = reinterpret_cast<uint32_t*>(planes + ((row >> 5) * (32 * NP) + (row & 31)) * H + j);
#pragma unroll
    for (int q = 0; q < NP; ++q) o[(size_t)q * 16 * H] = p[q];
}

// the same for four consecutive elements (row, j .. j+3), j % 4 == 0: one 8-byte store per plane
template <int NP> __device__ __forceinline__ void planes_store_quad(uint16_t* planes, int64_t row, int j, int H, float x0, float x1, float x2, float x3, float scale) {
    uint32_t p[3], q[3];
    split_pair_np<NP>(x0, x1, scale, p);
    split_pair_np<NP>(x2, x3, scale, q);
    uint2* o = reinterpret_cast<uint2*>(planes + ((row >> 5) * (32 * NP) + (row & 31)) * H + j);
#pragma unroll
    for (int k = 0; k < NP; ++k) o[(size_t)k * 8 * H] = make_uint2(p[k], q[k]);   // plane stride = 32 rows x H elements = 64 H bytes = 8 H uint2
}

// the same on the hardware exp2 / log2 / rcp (1 ulp each): log1p(e) = log(u) * e / (u - 1) with u = fl(1 + e) cancels the rounding of
// 1 + e (~2 ulp overall, against ~1 ulp of the library call that costs >100 vector instructions); also returns log(sigma) for the KL
__device__ __forceinline__ float softplus_rho_fast(float rho, float& log_sigma) {
    const float e = __builtin_amdgcn_exp2f(fminf(rho, 80.f) * 1.4426950408889634f);
    const float u = 1.f + e, d = u - 1.f;
    const float sigma = d == 0.f ? e : (__builtin_amdgcn_logf(u) * 0.6931471805599453f) * (e * __builtin_amdgcn_rcpf(d));
    log_sigma = __builtin_amdgcn_logf(sigma) * 0.6931471805599453f;
    return sigma;
}
__device__ __forceinline__ float softplus_rho(float rho) { return log1pf(expf(rho)); }  // sigma = log1p(exp(rho))

// terms of binary_cross_entropy_with_logits on l = leaky_relu(z) (src/mdl/fnn.py:25,46):
//   sp = softplus(l) = bce(l, y=0);  sg = sigmoid(l);  dact = d leaky_relu / dz
__device__ __forceinline__ void bce_terms(float z, float& sp, float& sg, float& dact) {
    const bool pos = z > 0.f;
    const float l = pos ? z : z * kLeakySlope;
    dact = pos ? 1.f : kLeakySlope;
    const float e = expf(-fabsf(l));
    sp = fmaxf(l, 0.f) + logf(1.f + e);
    const float inv = 1.f / (1.f + e);
    sg = (l >= 0.f) ? inv : e * inv;
}

// One Adam step of one element (torch.optim.Adam defaults, src/mdl/fnn.py:104,139: exp_avg.lerp_, exp_avg_sq.mul_().addcmul_, p.addcdiv_(exp_avg, sqrt(exp_avg_sq) /
// sqrt(bc2) + eps, -lr / bc1)).  EVERY Adam kernel of the engine calls this one (the flat kernel, the range kernel, the dW epilogues), so the paths agree bit for bit.
// Round 4: the square root and the two divisions on the hardware's v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE sequences (~10 instructions per division,
// ~12 per square root: a third of the fused dW epilogue's vector instructions).  The update term lr / bc1 * m / denom is then good to ~3e-7 relative, i.e. ~3e-10
// absolute at lr = 1e-3 - below half an ulp of any parameter larger than 5e-3; a subnormal exp_avg_sq (sqrt < 1e-19) vanishes beside eps = 1e-8 either way.
__device__ __forceinline__ void adam_step(float& p, float g, float& m, float& v, float lr_over_bc1, float b1, float b2, float eps, float inv_bc2_sqrt) {
    m = m + (1.f - b1) * (g - m);
    v = v * b2 + (1.f - b2) * g * g;
#ifdef NTF_ADAM_IEEE      // diagnostic builds only (profiles/r5_ep_tolerance.md): round 3's correctly rounded square root and division
    const float denom = fmaf(sqrtf(v), inv_bc2_sqrt, eps);
    p = p - (lr_over_bc1 * m) / denom;
#else
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), inv_bc2_sqrt, eps);
    p = p - (lr_over_bc1 * m) * __builtin_amdgcn_rcpf(denom);
#endif
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over a workgroup of up to 1024 threads; result valid in thread 0 (deterministic order)
__device__ __forceinline__ float block_reduce_sum(float v) {
    __shared__ float sh_f[16];
    v = wave_reduce_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh_f[w] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) for (int i = 0; i < nw; ++i) s += sh_f[i];
    return s;
}
__device__ __forceinline__ double block_reduce_sum_d(double v) {
    __shared__ double sh_d[16];
    v = wave_reduce_sum_d(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh_d[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < nw; ++i) s += sh_d[i];
    return s;
}

}  // namespace ntf
