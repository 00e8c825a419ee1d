// Shared by the two translation units of the fused output layer (ntf_fused.hip: forward / loss / dz / dh and the sparse fix-up; ntf_fused_dw.hip: weight gradients,
// Adam, next-step operands): tile constants, the dzT and workspace layouts, LDS-DMA from inline asm, the split-product MFMA helpers.  Internal, not part of the C ABI.
#pragma once
#include "ntf_fused.h"
#include "ntf_device.h"
#include <algorithm>
#include <vector>
#include <cstdio>
#include <type_traits>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace ntf {

constexpr int BM = 128;       // batch rows per workgroup: 4 waves x 32
constexpr int BN = 64;        // experts per tile
constexpr int NCG_MAX = 256;  // column groups (one workgroup per CU when the batch has a single row block)
#ifndef DW_PPG
#define DW_PPG 2      // DMA pieces of the next K block per MFMA group (1: 0.705, 2: 0.693, 4: 0.707 ms at config 2)
#endif
constexpr int DW_WAVES = 8;               // waves per workgroup of the dW kernel: two per SIMD
constexpr int DW_TC = 32 * DW_WAVES;      // experts per workgroup of the dW kernel

static inline int rup(int a, int b) { return (a + b - 1) / b * b; }
// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel that asks for more than 64 KiB of dynamic LDS: a property of (device, kernel) that never changes - set the first
// time that kernel is launched with that size, not once per launch (round 4 paid the runtime call on every launch of every big kernel)
inline void set_max_lds(const void* fn, int bytes) {
    static std::mutex mu; static std::unordered_map<uint64_t, int> done;
    int dev = 0; (void)hipGetDevice(&dev);
    const uint64_t key = (uint64_t)(uintptr_t)fn * 64u + (uint64_t)(dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[key] = bytes;
}
// Layout of dzT (d loss / d z of the output layer, the only dense [experts x batch] tensor of a step): tiles of DW_TC = 256 experts x 32 batch
// rows, [expert tile][32-row K block][expert in tile][row in block].  The dW kernel consumes one K block of its expert tile per stage: one
// contiguous 32 KiB piece, and a whole tile is a contiguous (Bpad / 32) * 32 KiB stream; the forward kernels' stores (32 rows = 128 B per expert)
// fall in the same granules as in a plain [expert][Bpad] matrix.
__host__ __device__ __forceinline__ int64_t dzt_index(int c, int i, int Bpad) {
    return ((int64_t)(c >> 8) * (Bpad >> 5) + (i >> 5)) * 8192 + ((c & 255) << 5) + (i & 31);
}
__host__ __device__ __forceinline__ int64_t dzt_tile_base(int c0, int Bpad) { return (int64_t)(c0 >> 8) * (Bpad >> 5) * 8192 + ((c0 & 255) << 5); }   // floats

struct Geom { int Bpad, NRB, T, NCG, nCB; };
inline Geom geom(int B, int M) {
    Geom g;
    g.Bpad = rup(B, BM); g.NRB = g.Bpad / BM; g.T = (M + BN - 1) / BN;
    g.NCG = std::max(1, std::min(NCG_MAX / g.NRB, g.T));
    g.nCB = rup((M + 31) / 32, 2);
    return g;
}
// column groups of a forward-only launch (evaluation, inference) at H = 128: twice the training step's, so that two workgroups share a CU (the loss / entropy partials have
// NCG_MAX slots a row)
inline int eval_ncg(const Geom& g) { return std::max(1, std::min(2 * NCG_MAX / g.NRB, std::min(g.T, NCG_MAX))); }
struct WsLayout { size_t sbits, sbitsT, sinbits, sinT, hs, hz, lossp, hb, total; };
inline WsLayout ws_layout(int Bmax, int H, int M) {
    const int Bpad = rup(Bmax, BM), nCB = rup((M + 31) / 32, 2);
    WsLayout w; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
    w.sbits = take((size_t)Bpad * nCB * 4);
    w.sbitsT = take((size_t)rup(M, DW_TC) * (Bpad / 32) * 4);   // [expert tile of 256][K block of 32 rows][256 experts]: word = s_out signs of the 32 rows
    w.sinbits = take((size_t)Bpad * (H / 32) * 4);
    w.sinT = take((size_t)(Bpad / 32) * H * 4);   // [K block][hidden unit]: s_in signs of the block's 32 rows (k_sin_words_T)
    w.hs = take((size_t)Bpad * H * 4);
    w.hz = take((size_t)Bpad * H * 4);
    w.lossp = take((size_t)Bpad * NCG_MAX * 4);
    w.hb = take((size_t)Bpad * H * 2 * 6);   // bf16 split planes of h and h*s_in, K-block tiled (k_prep_planes_T)
    w.total = o;
    return w;
}
// ------------------------------------------------------------------------------------------------
template <int H> __device__ __forceinline__ int swz(int row) { return H >= 64 ? (row & 15) : ((row >> 1) & 7); }
__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-DMA issued from inline asm: hipcc then neither counts it nor fences later ds_reads behind it with vmcnt(0)
// (which it does for the builtin, serialising the prefetch of tile t+1 with the compute on tile t).  The waits are
// placed by hand: a counted s_waitcnt vmcnt(N) before the barrier that publishes the buffer.
//   lds_dst = wave-uniform LDS byte address; lane l lands at lds_dst + l*size; gsrc = this lane's global source.
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);   // wave-uniform by construction; hipcc cannot always prove it (a loop-carried stage index)
#ifdef NTF_GLDS_KEEP_M0
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#else
    // M0 is DECLARED clobbered instead of saved and restored around every piece (two scalar instructions less per piece: 18 of a dW K block's ~300 issue slots):
    // hipcc then keeps nothing in M0 across the statement (it warns that M0 is a reserved register; none of these kernels uses it otherwise)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory", "m0");
#endif
}
// the same with the source as a wave-uniform base (an SGPR pair) + a 32-bit per-lane byte offset: no 64-bit vector address arithmetic per piece (hipcc spends a
// v_lshl_add_u64 on every `base + lane offset`: 17 of the forward kernel's ~950 vector instructions per tile)
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Row-loss accumulation over a column group (up to ~1.6e5 experts per lane at M = 5e6): the per-expert terms are nearly equal (softplus of a logit
// near 0), so a plain f32 running sum rounds every add the SAME way once it is large (+0.3 % per term in one binade, -0.8 % in the next: measured
// -335 ppm on the loss at M = 5 022 955).  Two levels: a tile's terms go to a fresh sum, the tile sums are added with compensation (Kahan).
struct LossAcc {
    float sum = 0.f, comp = 0.f, tile = 0.f;
    __device__ __forceinline__ void end_tile() {
        const float y = tile - comp, t = sum + y;
        comp = (t - sum) - y; sum = t; tile = 0.f;
    }
};

struct OutFwdArgs {
    int B, M, Bpad, NRB, NCG, T, nCB;
    // k_out_fwd_h3p on a RANGE of the experts (data-parallel ranks: one launch per all-gathered chunk of the parameters): the launch's NCG column groups walk the
    // 64-expert tiles [t_lo, t_hi) and write the dh slabs / loss partials cg_off .. cg_off + NCG - 1 of ncg_tot.  Whole layer: t_lo = 0, t_hi = T, cg_off = 0, ncg_tot = NCG
    int t_lo, t_hi, cg_off, ncg_tot;
    const float *h, *hs, *mu, *mu_b, *wp, *bp;
    const uint32_t *sbits, *sinbits;
    uint32_t so_k0, so_k1, si_k0, si_k1;   // native sign generators; *_inj != 0: read the packed images instead (injected signs)
    int so_inj, si_inj;
    float tnw, inv_B;
    float *dzT, *slab, *lossp;
    // fp16x3 range guard: rmode 1 (split-product kernels) = do nothing when *rflag is raised; rmode 2 (the exact-f32 kernels launched right
    // behind them) = run ONLY then, and count the step in rflag[1]; rmode 0 = unconditional
    int* rflag; int rmode;
};
__device__ __forceinline__ bool range_guard_skip(int* rflag, int rmode, bool count) {
    if (rmode == 0) return false;
    const int raised = __builtin_nontemporal_load(rflag);
    if (rmode == 1) return raised != 0;
    if (raised == 0) return true;
    if (count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(rflag + 1, 1);
    return false;
}

// f(integral_constant<int, I>) for I = I0 .. N - 1: an unrolled loop by construction
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// N consecutive floats (N = 1, 2, 4) as one access
template <int N> __device__ __forceinline__ void ld_vec(const float* p, float (&v)[N]) {
    if (N == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[N > 1 ? 1 : 0] = t.y; v[N > 2 ? 2 : 0] = t.z; v[N > 3 ? 3 : 0] = t.w; }
    else if (N == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[N > 1 ? 1 : 0] = t.y; }
    else v[0] = p[0];
}
template <int N> __device__ __forceinline__ void st_vec(float* p, const float (&v)[N]) {
    if (N == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[N > 1 ? 1 : 0], v[N > 2 ? 2 : 0], v[N > 3 ? 3 : 0]);
    else if (N == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[N > 1 ? 1 : 0]);
    else p[0] = v[0];
}

// ================================================================================================
// bf16x6 variant: every f32 operand x is split exactly into three bf16 values x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2); 24 mantissa bits in all), and a product a*b is taken as the six bf16 MFMA products
// a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 accumulated in f32.  Each bf16 x bf16 product is exact in f32 and the dropped terms are
// <= 2^-24 relative, so the result carries f32 accuracy (measured: the same error against f64 as the f32 MFMA) while the matrix
// pipe does 6 x 32 cycles per 16-deep k-step of a 32x32 tile instead of 8 x 64 (v_mfma_f32_32x32x2_f32) — and, unlike the f32 MFMA,
// v_mfma_f32_32x32x16_bf16 leaves 24 of its 32 cycles free for vector instructions of the same wave.
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 as_frag_h(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
// acc += a*b with a = a1+a2+a3, b = b1+b2+b3 (smallest terms first)
__device__ __forceinline__ f32x16 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a[2]), as_frag(b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a[0]), as_frag(b[2]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a[1]), as_frag(b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a[1]), as_frag(b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a[0]), as_frag(b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a[0]), as_frag(b[0]), acc, 0, 0, 0);
    return acc;
}

// fp16x3 variant of the same idea: x * 2^k (k per tensor, exact) split into two fp16 values (22 mantissa bits; fp16 subnormals are honoured by
// the MFMA, checked on the hardware), a product = a1b1 + a1b2 + a2b1 on v_mfma_f32_32x32x16_f16: half the matrix work of bf16x6 for an error
// against f64 1.2 x that of the f32 MFMA (5.7e-7 vs 4.6e-7 of the largest element on the forward product).
__device__ __forceinline__ f32x16 mfma3h(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(a[1]), as_frag_h(b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(a[0]), as_frag_h(b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(a[0]), as_frag_h(b[0]), acc, 0, 0, 0);
    return acc;
}
template <int NP> __device__ __forceinline__ f32x16 mfma_np(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
    if (NP == 3) return mfma6(a, b, acc); else return mfma3h(a, b, acc);
}

}  // namespace ntf
