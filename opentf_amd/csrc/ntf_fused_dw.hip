// Weight-gradient side of the fused output layer for gfx950 (split from ntf_fused.hip in round 4): dmu = dzT . h, dWp = (dzT * s_out) . (h * s_in) with K = batch, the
// Flipout rho-gradient + KL, Adam in place and the NEXT step's Flipout operands in the epilogue.
//   k_out_dw          exact-f32 MFMA (v_mfma_f32_32x32x2_f32); k_out_dw_fallback: the same as the range fallback behind a split-product kernel
//   k_out_dw_b6       fp16x3 split products on f32 dz (H = 32, 64; H = 128 without the packed forward kernel)
//   k_out_dw_q        fp16x3 on the forward kernel's packed dz planes: two 128-expert workgroups per CU, one's epilogue beside the other's main loop; <.., SPLIT>: a K range
//                     per workgroup for narrow expert shards.  (Round 2-3's one-workgroup-per-CU form k_out_dw_p2 - bit-identical sums - was retired in round 6)
//   k_out_dw_finish   sum of split-K partial slabs + the epilogue
// plus the operand images these kernels read: K-block-tiled planes of h (k_prep_planes_T), transposed s_out / s_in sign words (k_sign_words_T, k_sin_words_T).
#include "ntf_fused_common.h"

namespace ntf {

// ------------------------------------------------------------------------------------------------
struct DwArgs {
    int B, M, Bpad;
    const float *__restrict__ dzT, *__restrict__ h, *__restrict__ hs, *__restrict__ mu, *__restrict__ rho;
    const float* wp;     // (not restrict: with `produce` the epilogue overwrites the element it has just read with the next step's value)
    const uint32_t* sbits; int nCB; uint32_t so_k0, so_k1; int so_inj;   // s_out signs: packed row image (injected) or hash keys
    float *__restrict__ g_mu, *__restrict__ g_rho, *__restrict__ g_b, *__restrict__ g_bp;
    float klw;
    // fused Adam (single GPU): update mu / rho and their moments in the epilogue instead of writing the gradients
    float *__restrict__ w_mu, *__restrict__ w_rho, *__restrict__ m_mu, *__restrict__ v_mu, *__restrict__ m_rho, *__restrict__ v_rho;
    float lr_over_bc1, b1, b2, eps, bc2_sqrt;
    int wg_begin;   // first expert tile of this launch (the expert range can be launched in chunks)
    int* rflag; int rmode;   // fp16x3 range guard, see OutFwdArgs
    const uint32_t* sT;      // k_sign_words_T image (fp16x3 packed path)
    const uint32_t* sinT;    // k_sin_words_T image: s_in signs of (K block, hidden unit) over the block's 32 rows (k_out_dw_q)
    const uint16_t* hb;   // split planes of h / h*s_in (k_prep_planes_T)
    float a_scale, unscale;   // fp16x3: dz is scaled by a_scale before its split; accumulators are multiplied by unscale = 1 / (a_scale * h scale)
    // split-K (k_out_dw_q<.., SPLIT>): few expert tiles (a narrow expert shard under a wide minibatch) are launched ksplit times, each workgroup summing a
    // contiguous part of the K blocks into part[(split * 2 + matrix) * slab ..] (bias sums behind the slabs); k_out_dw_finish adds the parts and runs the epilogue
    int ksplit; float* part; int64_t slab; int part_row0;   // part_row0: first expert of the launched tile range - the slabs hold that range only
    unsigned long long* stamps;   // diagnostics (k_out_dw_q<.., STAMP>, NTF_DW_STAMP_FILE)
    int ntile, stagger;   // k_out_dw_q, unsplit: expert tiles of this launch (walked by persistent workgroups), start delay of every second workgroup (100 MHz ticks)
    // produce != 0 (fused Adam, Flipout, fp16x3 planes): the epilogue holds the UPDATED mu' / rho' of its elements - it also is the next step's operand producer:
    // eps' (Philox keyed by step + 1), Wp' = softplus(rho') eps' (f32, in place over this step's Wp), the fp16 split planes of Wp' and mu' that the forward kernel
    // streams, the layer's KL' and fp16 range flag of the next step.  Saves k_flipout_perturb's own pass over the layer (0.72 GB, 0.12 ms at config 2) and takes
    // it off the path between two steps: what it would read is in registers here.
    NormalSpec cur_eps; int lean;   // see FusedDw
    int produce; NormalSpec nx_eps; float* nx_wp; uint16_t *nx_pl_wp, *nx_pl_mu; float nx_pscale; double nx_klw; double* nx_kl; int* nx_rflag;
};

__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, float lr_over_bc1, float b1, float b2, float eps, float bc2_sqrt) {
    adam_step(p, g, m, v, lr_over_bc1, b1, b2, eps, __builtin_amdgcn_rcpf(bc2_sqrt));   // ntf_device.h: the one Adam expression of every kernel
    return p;
}

// 32x32 bit-matrix transpose across the 32 lanes of a half-wave (lane l holds row l): five butterfly stages of masked
// swaps with the lane l ^ j.  Afterwards lane l holds column l (bit k = old row k).
__device__ __forceinline__ uint32_t transpose32(uint32_t a, int il) {
    uint32_t m = 0x0000FFFFu;
#pragma unroll
    for (int j = 16; j != 0; j >>= 1) {
        const uint32_t pv = (uint32_t)__shfl_xor((int)a, j, 64);
        const bool lower = (il & j) == 0;
        // bit index = column (LSB first): swap the lower lane's bits with (bit & j) set against the upper lane's bits without
        const uint32_t lo = lower ? a : pv, hi = lower ? pv : a;
        const uint32_t t = ((lo >> j) ^ hi) & m;
        a ^= lower ? (t << j) : t;
        m ^= m << (j >> 1);
    }
    return a;
}

// DwArgs.produce: the next step's operands of the quad idx0 .. idx0 + 3 (one expert's four consecutive hidden units) from its updated parameters; the arithmetic is
// k_flipout_perturb's (ntf_kernels.hip), so a step that follows reads bit for bit what the stand-alone producer would have written
__device__ __forceinline__ void dw_produce_next(const DwArgs& p, int64_t idx0, const float (&mu4)[4], const float (&rho4)[4], float& kl, float& amax) {
    float z[4], ov[4];
#if defined(DWQ_ABL) && (DWQ_ABL & 2)      // timing-only builds (profiles/mk_variants.sh def ...: results are garbage): the epilogue without the next step's draw
    z[0] = z[1] = z[2] = z[3] = 1.f;
#else
    normal4(p.nx_eps, idx0 >> 2, idx0, INT64_MAX, z);
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float ls;
        const float sigma = softplus_rho_fast(rho4[j], ls);
        ov[j] = sigma * z[j];
        kl += -ls + 0.5f * (sigma * sigma + mu4[j] * mu4[j]) - 0.5f;
    }
    if (!p.lean) *reinterpret_cast<float4*>(p.nx_wp + idx0) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    const int64_t row = idx0 >> 7; const int j = (int)(idx0 & 127);      // H = 128
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(ov[0]), fabsf(ov[1])), fmaxf(fabsf(ov[2]), fabsf(ov[3]))));
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(mu4[0]), fabsf(mu4[1])), fmaxf(fabsf(mu4[2]), fabsf(mu4[3]))));
    planes_store_quad<2>(p.nx_pl_wp, row, j, 128, ov[0], ov[1], ov[2], ov[3], p.nx_pscale);
    planes_store_quad<2>(p.nx_pl_mu, row, j, 128, mu4[0], mu4[1], mu4[2], mu4[3], p.nx_pscale);
}
// Fnn (round 6): no Flipout operand - the next step's operand is the fp16 planes of the UPDATED mu alone (k_split_planes made them in a pass of its own over the layer at
// the head of every step: 4 B read + 4 B written per weight on the way to the forward kernel)
__device__ __forceinline__ void dw_produce_next_fnn(const DwArgs& p, int64_t idx0, const float (&mu4)[4], float& amax) {
    const int64_t row = idx0 >> 7; const int j = (int)(idx0 & 127);      // H = 128
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(mu4[0]), fabsf(mu4[1])), fmaxf(fabsf(mu4[2]), fabsf(mu4[3]))));
    planes_store_quad<2>(p.nx_pl_mu, row, j, 128, mu4[0], mu4[1], mu4[2], mu4[3], p.nx_pscale);
}
// ... and once per workgroup: the KL' sum (one double atomic, as the stand-alone producer) and the range flag.  red = 8-byte-aligned LDS scratch of >= nwaves doubles
template <bool BAYES = true>
__device__ __forceinline__ void dw_produce_finish(const DwArgs& p, float kl, float amax, double* red, int nwaves) {
    if (!(amax * p.nx_pscale <= 65504.f)) *p.nx_rflag = 1;
    if (!BAYES) return;
    const double s = wave_reduce_sum_d((double)kl);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < nwaves; ++w) t += red[w]; atomicAdd(p.nx_kl, t * p.nx_klw); }
}

template <int H, bool BAYES, bool ADAM>
__device__ __forceinline__ void out_dw_f32_body(const DwArgs& p, char* smem) {  // ADAM: see DwArgs
    constexpr int NJT = H / 32;
    constexpr int KB = 32;                  // batch rows per K block
    constexpr int HROW = 4 * H;
    constexpr int TA = DW_TC * KB * 4;      // dzT tile [256 experts][32 batch rows], 16-byte chunks XOR-swizzled
    constexpr int TH = KB * HROW;           // h tile [32][H] (and h*s_in behind it)
    constexpr int STAGE = TA + (BAYES ? 2 : 1) * TH;   // two stages: the DMA of K block b+1 runs under the MFMAs of K block b
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, half = lane >> 5;
    if (range_guard_skip(p.rflag, p.rmode, false)) return;
    const int c0 = (p.wg_begin + blockIdx.x) * DW_TC;
    const int crow = wave * 32 + il;        // this lane's expert row inside the tile
    const int c = c0 + crow;
    const int nib = p.Bpad / KB;

    f32x16 acc1[NJT], acc2[NJT];
#pragma unroll
    for (int j = 0; j < NJT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[j][r] = 0.f; acc2[j][r] = 0.f; }
    float sum1 = 0.f, sum2 = 0.f;

    const uint32_t smem_base = lds_addr(smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto stage = [&](int ib, int buf) {
        const uint32_t sb = smem_base + buf * STAGE;
#pragma unroll
        for (int n = 0; n < TA / 1024 / DW_WAVES; ++n) {      // 1 KiB wave-instructions: 8 expert rows x 128 B
            const int inst = wave_u * (TA / 1024 / DW_WAVES) + n;
            const int row = inst * 8 + (lane >> 3), pch = lane & 7;
            const int q = pch ^ ((row >> 1) & 7);
            glds16(p.dzT + ((int64_t)(c0 >> 8) * nib + ib) * 8192 + row * 32 + 4 * q, sb + inst * 1024);   // the K block of this tile: contiguous 32 KiB
        }
        constexpr int HI = TH / 1024;                           // wave-instructions per h tile (may be fewer than the waves)
#pragma unroll
        for (int n = 0; n < (HI + DW_WAVES - 1) / DW_WAVES; ++n) {
            const int inst = wave_u * ((HI + DW_WAVES - 1) / DW_WAVES) + n;
            if (inst < HI) {
                const int64_t goff = (int64_t)ib * KB * H + inst * 256 + lane * 4;  // h tile rows are contiguous in memory
                glds16(p.h + goff, sb + TA + inst * 1024);
                if (BAYES) glds16(p.hs + goff, sb + TA + TH + inst * 1024);
            }
        }
    };

    // s_out sign bits of (32 batch rows of the K block) x (this wave's 32 experts): lane il produces the row word of batch row
    // ib*32 + il for the wave's column block (one hash, or one load from the injected image), then a 32x32 bit transpose
    // across lanes leaves lane il with the word of ITS expert (bit k = batch row ib*32 + k).
    const uint32_t cb = (uint32_t)((c0 + wave * 32) >> 5);
    auto sign_col_word = [&](int ib) -> uint32_t {
        const int i = ib * KB + il;
        uint32_t w = 0u;
        if (p.so_inj) { if ((int)cb < p.nCB) w = p.sbits[(int64_t)i * p.nCB + cb]; }
        else w = sign_word(p.so_k0, p.so_k1, (uint32_t)i, cb);
        return transpose32(w, il);
    };
    uint32_t word_next = 0;
    if (BAYES) word_next = sign_col_word(0);
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ib = 0; ib < nib; ++ib) {
        const int buf = ib & 1;
        const uint32_t word = word_next >> (4 * half);
        if (BAYES && ib + 1 < nib) word_next = sign_col_word(ib + 1);
        if (ib + 1 < nib) stage(ib + 1, buf ^ 1);
        const char* sA = smem + buf * STAGE;
        const char* sH = sA + TA;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = 2 * t + half;
            const float4 a = *reinterpret_cast<const float4*>(sA + crow * 128 + 16 * (q ^ ((crow >> 1) & 7)));
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int kk = 8 * t + e4;  // + 4*half : batch row inside the K block
                const float a1 = av[e4];
                sum1 += a1;
                float a2 = a1;
                if (BAYES) { a2 = __uint_as_float(__float_as_uint(a1) ^ ((word << (31 - kk)) & 0x80000000u)); sum2 += a2; }
                // ONE wide read feeds all NJT column tiles: lane il owns hidden units j = NJT*il + jt
                const char* hb = sH + (kk + 4 * half) * HROW + 4 * NJT * il;
                float bv[NJT], bsv[NJT];
                if (NJT == 4) {
                    const float4 b4 = *reinterpret_cast<const float4*>(hb); bv[0] = b4.x; bv[1] = b4.y; bv[NJT > 2 ? 2 : 0] = b4.z; bv[NJT > 3 ? 3 : 0] = b4.w;
                    if (BAYES) { const float4 s4 = *reinterpret_cast<const float4*>(hb + TH); bsv[0] = s4.x; bsv[1] = s4.y; bsv[NJT > 2 ? 2 : 0] = s4.z; bsv[NJT > 3 ? 3 : 0] = s4.w; }
                } else if (NJT == 2) {
                    const float2 b2 = *reinterpret_cast<const float2*>(hb); bv[0] = b2.x; bv[NJT > 1 ? 1 : 0] = b2.y;
                    if (BAYES) { const float2 s2 = *reinterpret_cast<const float2*>(hb + TH); bsv[0] = s2.x; bsv[NJT > 1 ? 1 : 0] = s2.y; }
                } else {
                    bv[0] = *reinterpret_cast<const float*>(hb);
                    if (BAYES) bsv[0] = *reinterpret_cast<const float*>(hb + TH);
                }
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    acc1[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[jt], acc1[jt], 0, 0, 0);
                    if (BAYES) acc2[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, bsv[jt], acc2[jt], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next K block landed (the DMA is invisible to hipcc's own counting)
        __syncthreads();                                     // ... and this one is fully consumed
    }

    sum1 += __shfl_xor(sum1, 32, 64);
    sum2 += __shfl_xor(sum2, 32, 64);
    if (half == 0 && c < p.M) { p.g_b[c] = sum1; if (BAYES) p.g_bp[c] = sum2; }

    float nx_kl = 0.f, nx_amax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cr = c0 + wave * 32 + rowmap(r, half);
        if (cr >= p.M) continue;
        const int64_t idx0 = (int64_t)cr * H + NJT * il;  // NJT consecutive hidden units per lane: one wide access per array
        float gm[NJT], gr[NJT], pm[NJT], pr[NJT];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            pm[jt] = ADAM ? p.w_mu[idx0 + jt] : (BAYES ? p.mu[idx0 + jt] : 0.f);
            if (!BAYES) { gm[jt] = acc1[jt][r]; gr[jt] = 0.f; pr[jt] = 0.f; }
            else {
                const float rh = ADAM ? p.w_rho[idx0 + jt] : p.rho[idx0 + jt], w = p.wp[idx0 + jt];
                pr[jt] = rh;
                const float sigma = softplus_rho(rh);
                const float sg = 1.f / (1.f + expf(-rh));
                gm[jt] = acc1[jt][r] + p.klw * pm[jt];
                gr[jt] = acc2[jt][r] * (w / sigma) * sg + p.klw * (sigma - 1.f / sigma) * sg;
            }
        }
        if (!ADAM) {
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) { p.g_mu[idx0 + jt] = gm[jt]; if (BAYES) p.g_rho[idx0 + jt] = gr[jt]; }
        } else {
            float nmu[4] = {0.f, 0.f, 0.f, 0.f}, nrho[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                float m = p.m_mu[idx0 + jt], v = p.v_mu[idx0 + jt];
                nmu[jt & 3] = p.w_mu[idx0 + jt] = adam_update(pm[jt], gm[jt], m, v, p.lr_over_bc1, p.b1, p.b2, p.eps, p.bc2_sqrt);
                p.m_mu[idx0 + jt] = m; p.v_mu[idx0 + jt] = v;
                if (BAYES) {
                    float m2 = p.m_rho[idx0 + jt], v2 = p.v_rho[idx0 + jt];
                    nrho[jt & 3] = p.w_rho[idx0 + jt] = adam_update(pr[jt], gr[jt], m2, v2, p.lr_over_bc1, p.b1, p.b2, p.eps, p.bc2_sqrt);
                    p.m_rho[idx0 + jt] = m2; p.v_rho[idx0 + jt] = v2;
                }
            }
            if constexpr (H == 128) { if (p.produce) { if constexpr (BAYES) dw_produce_next(p, idx0, nmu, nrho, nx_kl, nx_amax); else dw_produce_next_fnn(p, idx0, nmu, nx_amax); } }   // (this kernel as the fp16x3 step's range fallback)
        }
    }
    if constexpr (ADAM && H == 128) { if (p.produce) dw_produce_finish<BAYES>(p, nx_kl, nx_amax, reinterpret_cast<double*>(smem), DW_WAVES); }
}

template <int H, bool BAYES, bool ADAM>
__global__ __launch_bounds__(64 * DW_WAVES, 2) void k_out_dw(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    out_dw_f32_body<H, BAYES, ADAM>(p, smem);
}
// the same as the range FALLBACK behind a split-product kernel that cannot run the f32 body itself (k_out_dw_q, the split-K launches): it runs only in a step whose
// range flag is raised - in every other step it is a no-op on the critical path, so it is launched on ONE round of workgroups that walk the tiles (p.ntile of
// them) instead of one workgroup per tile: 913 early exits of 128 KB-LDS workgroups took 5 us per step, 256 take under 2
template <int H, bool BAYES, bool ADAM>
__global__ __launch_bounds__(64 * DW_WAVES, 2) void k_out_dw_fallback(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (range_guard_skip(p.rflag, p.rmode, false)) return;
    for (int t = (int)blockIdx.x; t < p.ntile; t += (int)gridDim.x) {
        DwArgs q = p; q.rmode = 0; q.wg_begin = p.wg_begin + t - (int)blockIdx.x;      // (the body takes its tile as wg_begin + blockIdx.x)
        out_dw_f32_body<H, BAYES, ADAM>(q, smem);
        __syncthreads();
    }
}

// hb: for every 32-row K block ib of the batch, the planes [p = h1,h2,h3,(hs1,hs2,hs3)][j][r = 0..31] of bf16 — the B operand
// (k = batch row, n = hidden unit) of the dW products reads 8 consecutive batch rows of one hidden unit as one 16-byte chunk.
__global__ void k_prep_planes_T(const float* __restrict__ hz, const float* __restrict__ hs, int bayes, int Bpad, int H, int np, float scale,
                                uint16_t* __restrict__ hb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // (ib, j, r), r fastest
    if (t >= Bpad * H) return;
    const int r = t & 31, j = (t >> 5) % H, ib = t / (32 * H);
    const int npl = (bayes ? 2 : 1) * np;
    uint16_t* tile = hb + (size_t)ib * npl * H * 32;
    // slot of hidden unit j in the image: column tile jt = j % NJT, lane il = j / NJT (NJT = H / 32).  The dW kernels read slot (jt, il) as the B column of
    // lane il in tile jt, so a lane's NJT accumulators are NJT CONSECUTIVE hidden units: its epilogue moves 4 NJT-byte pieces, 32 lanes one whole row
    const int njt = H >> 5, slot = 32 * (j % njt) + j / njt;
    for (int q = 0; q < (bayes ? 2 : 1); ++q) {
        const float x = (q ? hs : hz)[(int64_t)(ib * 32 + r) * H + j];
        uint32_t p[3];
        if (np == 3) split_pair_np<3>(x, 0.f, 1.f, p); else split_pair_np<2>(x, 0.f, scale, p);
        for (int k = 0; k < np; ++k) tile[((q * np + k) * H + slot) * 32 + r] = (uint16_t)p[k];
    }
}

// dW with bf16x6 products.  Workgroup = 8 waves x 32 experts; K = batch in 32-row blocks, two LDS stages filled by LDS-DMA:
//   A (k = batch row): the f32 dzT tile [256 experts][32 rows], 16-byte chunks XOR-swizzled ((row>>1)&7) exactly as in k_out_dw; a lane reads
//     the 8 consecutive values of ITS expert (two ds_read_b128) and splits them in registers;
//   B: the bf16 planes of h / h*s_in for the K block ([plane][j][32 rows], chunks swizzled with (j>>2)&3): one ds_read_b128 per fragment.
// No compiler-visible global load sits in the loop: hipcc would wait for it with a vmcnt that, in the real in-order queue, also waits
// for the DMA issued just before.
template <int H, bool BAYES, bool ADAM, int NP>
__global__ __launch_bounds__(64 * DW_WAVES, 2) void k_out_dw_b6(DwArgs p) {   // ADAM: update mu / rho and their moments in the epilogue (see DwArgs)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NJT = H / 32;
    constexpr int NPL = (BAYES ? 2 : 1) * NP;   // NP = 3: bf16x6, NP = 2: fp16x3 (planes pre-scaled by 2^k, dz scaled by p.a_scale here)
    constexpr int TA = DW_TC * 32 * 4;            // dzT tile bytes
    constexpr int PLANE = H * 64;                 // bytes of one plane of one K block: [H][32 rows] bf16
    constexpr int TB = NPL * PLANE;
    constexpr int STAGE = TA + TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, half = lane >> 5;
    if (range_guard_skip(p.rflag, p.rmode, false)) return;
    const int c0 = (p.wg_begin + blockIdx.x) * DW_TC;
    const int crow = wave * 32 + il;
    const int c = c0 + crow;
    const int nib = p.Bpad / 32;
    const uint32_t smem_base = lds_addr(smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* hb = reinterpret_cast<const char*>(p.hb);

    f32x16 acc1[NJT], acc2[NJT];
#pragma unroll
    for (int j = 0; j < NJT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[j][r] = 0.f; acc2[j][r] = 0.f; }
    float sum1 = 0.f, sum2 = 0.f;

    auto stage = [&](int ib, int buf) {
        const uint32_t sb = smem_base + buf * STAGE;
#pragma unroll
        for (int n = 0; n < TA / 1024 / DW_WAVES; ++n) {      // 1 KiB wave-instructions: 8 expert rows x 128 B
            const int inst = wave_u * (TA / 1024 / DW_WAVES) + n;
            const int row = inst * 8 + (lane >> 3), pch = lane & 7;
            const int q = pch ^ ((row >> 1) & 7);
            glds16(p.dzT + ((int64_t)(c0 >> 8) * nib + ib) * 8192 + row * 32 + 4 * q, sb + inst * 1024);   // the K block of this tile: contiguous 32 KiB
        }
        const char* src = hb + (size_t)ib * TB;
        constexpr int NINST = TB / 1024;
#pragma unroll
        for (int n = 0; n < (NINST + DW_WAVES - 1) / DW_WAVES; ++n) {
            const int inst = wave_u * ((NINST + DW_WAVES - 1) / DW_WAVES) + n;
            if (inst < NINST) {
                const int pos = inst * 1024 + lane * 16;          // destination byte inside the plane area
                const int j = (pos % PLANE) >> 6, cd = (pos >> 4) & 3;
                glds16(src + (pos & ~63) + 16 * (cd ^ ((j >> 2) & 3)), sb + TA + inst * 1024);
            }
        }
    };
    const uint32_t cb = (uint32_t)((c0 + wave * 32) >> 5);
    auto sign_col_word = [&](int ib) -> uint32_t {   // bit k = s_out sign of (batch row ib*32 + k, this lane's expert)
        const int i = ib * 32 + il;
        uint32_t w = 0u;
        if (p.so_inj) { if ((int)cb < p.nCB) w = p.sbits[(int64_t)i * p.nCB + cb]; }
        else w = sign_word(p.so_k0, p.so_k1, (uint32_t)i, cb);
        return transpose32(w, il);
    };
    uint32_t word_next = 0;
    if (BAYES) word_next = sign_col_word(0);
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ib = 0; ib < nib; ++ib) {
        const int buf = ib & 1;
        const uint32_t word = word_next;
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + TA;
        // the K block as a flat, software-pipelined sequence of half-groups hg = (ks, jt, plain | signed): 3 fragment reads + 6 MFMAs
        // each; the reads of half-group hg+1 are in flight while the MFMAs of hg run (explicit double buffer: left to itself hipcc
        // reuses the fragment registers and waits for every ds_read right before the MFMA that needs it)
        constexpr int NHG = 2 * NJT * (BAYES ? 2 : 1);
        const char* bbase = sB + il * 64;
        const int swz = (il >> 2) & 3;
        auto load_b = [&](int hg, u32x4 (&dst)[3]) {
            const int which = BAYES ? (hg & 1) : 0, g = BAYES ? (hg >> 1) : hg, ks = g / NJT, jt = g % NJT;
            const char* bp = bbase + jt * 2048 + 16 * ((2 * ks + half) ^ swz) + which * NP * PLANE;
#pragma unroll
            for (int q = 0; q < NP; ++q) dst[q] = *reinterpret_cast<const u32x4*>(bp + q * PLANE);
        };
        u32x4 a[2][3], as[2][3];
        auto prep_a = [&](int ks) {
            const int ch = 4 * ks + 2 * half, sw = (crow >> 1) & 7;
            const float4 lo = *reinterpret_cast<const float4*>(sA + crow * 128 + 16 * (ch ^ sw));
            const float4 hi = *reinterpret_cast<const float4*>(sA + crow * 128 + 16 * ((ch + 1) ^ sw));
            const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const uint32_t w8 = word >> (ks * 16 + half * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t pq[3];
                split_pair_np<NP>(x[2 * q], x[2 * q + 1], p.a_scale, pq);
                a[ks][0][q] = pq[0]; a[ks][1][q] = pq[1]; a[ks][2][q] = pq[2];
                sum1 += x[2 * q] + x[2 * q + 1];
                if (BAYES) {
                    const uint32_t m = ((w8 << (15 - 2 * q)) & 0x8000u) | ((w8 << (30 - 2 * q)) & 0x80000000u);
                    as[ks][0][q] = pq[0] ^ m; as[ks][1][q] = pq[1] ^ m; as[ks][2][q] = pq[2] ^ m;
                    sum2 += __uint_as_float(__float_as_uint(x[2 * q]) ^ ((w8 << (31 - 2 * q)) & 0x80000000u)) +
                            __uint_as_float(__float_as_uint(x[2 * q + 1]) ^ ((w8 << (30 - 2 * q)) & 0x80000000u));
                }
            }
        };
        u32x4 bq[2][3];
        load_b(0, bq[0]);
        prep_a(0);
#pragma unroll
        for (int hg = 0; hg < NHG; ++hg) {
            if (hg + 1 < NHG) load_b(hg + 1, bq[(hg + 1) & 1]);
            asm volatile("" ::: "memory");   // keep the prefetch above this half-group's MFMAs
            const int which = BAYES ? (hg & 1) : 0, g = BAYES ? (hg >> 1) : hg, ks = g / NJT, jt = g % NJT;
            if (which) acc2[jt] = mfma_np<NP>(as[ks], bq[hg & 1], acc2[jt]);
            else acc1[jt] = mfma_np<NP>(a[ks], bq[hg & 1], acc1[jt]);
            if (hg == (NHG / 2 > 1 ? 1 : 0)) prep_a(1);   // before the first k-step-1 half-group; its vector work runs in the shadow of the following MFMAs
            // next K block: DMA issue + sign words in the middle of the MFMA phase, not in front of it — the two waves of a SIMD leave
            // every barrier in phase, and vector work bunched at the top of the iteration would meet the partner's vector work there
            if (hg == (NP == 2 ? 0 : NHG / 2) && ib + 1 < nib) { stage(ib + 1, buf ^ 1); if (BAYES && !p.so_inj) word_next = sign_col_word(ib + 1); }   // fp16x3: the K block is short, give the DMA all of it
        }
        if (BAYES && p.so_inj && ib + 1 < nib) word_next = sign_col_word(ib + 1);   // injected signs (tests): a visible load, kept out of the MFMA phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next K block (DMA) has landed
        __syncthreads();
    }

    // each half of the wave summed its 8 of every 16 batch rows
    sum1 += __shfl_xor(sum1, 32, 64);
    sum2 += __shfl_xor(sum2, 32, 64);
    if (half == 0 && c < p.M) { p.g_b[c] = sum1; if (BAYES) p.g_bp[c] = sum2; }

#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cr = c0 + wave * 32 + rowmap(r, half);
        if (cr >= p.M) continue;
        // lane il holds the NJT consecutive hidden units NJT*il .. (k_prep_planes_T's slot order): one vector access per array and row
        float v_rho[NJT], v_mu[NJT], v_wp[NJT], o_mu[NJT], o_rho[NJT];
        const int64_t idx0 = (int64_t)cr * H + NJT * il;
        if (BAYES) { ld_vec<NJT>((ADAM ? p.w_rho : p.rho) + idx0, v_rho); ld_vec<NJT>((ADAM ? p.w_mu : p.mu) + idx0, v_mu); ld_vec<NJT>(p.wp + idx0, v_wp); }
        else if (ADAM) ld_vec<NJT>(p.w_mu + idx0, v_mu);
        float a_m1[NJT], a_v1[NJT], a_m2[NJT], a_v2[NJT];
        if (ADAM) { ld_vec<NJT>(p.m_mu + idx0, a_m1); ld_vec<NJT>(p.v_mu + idx0, a_v1); if (BAYES) { ld_vec<NJT>(p.m_rho + idx0, a_m2); ld_vec<NJT>(p.v_rho + idx0, a_v2); } }
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            float gm = acc1[jt][r] * p.unscale, gr = 0.f, pm = 0.f, rh = 0.f;   // unscale: 1 / (dz scale * h scale), 1 for bf16x6
            if (BAYES) {
                rh = v_rho[jt];
                pm = v_mu[jt];
                const float w = v_wp[jt];
                // sigma = log1p(e^rho), sigmoid(rho) = e^rho / (1 + e^rho) on the hardware exp2/log2/rcp (the library expf/log1pf cost more vector
                // instructions here than the whole K loop); the short series keeps log1p accurate where 1 + e^rho rounds
                const float e = __builtin_amdgcn_exp2f(fminf(rh, 80.f) * 1.44269504f), t = 1.f + e;
                const float sigma = e < 0.015625f ? e * (1.f - e * (0.5f - e * (0.33333333f - 0.25f * e))) : __builtin_amdgcn_logf(t) * 0.69314718f;
                const float sg = e * __builtin_amdgcn_rcpf(t), isig = __builtin_amdgcn_rcpf(sigma);
                gm += p.klw * pm;
                gr = (acc2[jt][r] * p.unscale) * (w * isig) * sg + p.klw * (sigma - isig) * sg;
            } else if (ADAM) pm = v_mu[jt];
            o_mu[jt] = gm; o_rho[jt] = gr;
            if (ADAM) {   // in place: the updated parameter goes where the gradient would have gone
                o_mu[jt] = adam_update(pm, gm, a_m1[jt], a_v1[jt], p.lr_over_bc1, p.b1, p.b2, p.eps, p.bc2_sqrt);
                if (BAYES) o_rho[jt] = adam_update(rh, gr, a_m2[jt], a_v2[jt], p.lr_over_bc1, p.b1, p.b2, p.eps, p.bc2_sqrt);
            }
        }
        if (!ADAM) { st_vec<NJT>(p.g_mu + idx0, o_mu); if (BAYES) st_vec<NJT>(p.g_rho + idx0, o_rho); }
        else {
            st_vec<NJT>(p.w_mu + idx0, o_mu); st_vec<NJT>(p.m_mu + idx0, a_m1); st_vec<NJT>(p.v_mu + idx0, a_v1);
            if (BAYES) { st_vec<NJT>(p.w_rho + idx0, o_rho); st_vec<NJT>(p.m_rho + idx0, a_m2); st_vec<NJT>(p.v_rho + idx0, a_v2); }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// s_out sign words of the dW kernel: sT[expert tile of 256][K block][expert in tile], bit k = sign of (batch row 32 ib + k, expert) - the 32x32 bit
// transposes of the row words (hash, or the packed image of injected signs), made once per step instead of once per K block inside the dW kernel
constexpr int SWT_IB = 16;
__global__ __launch_bounds__(256) void k_sign_words_T(const uint32_t* __restrict__ sbits, int so_inj, uint32_t k0, uint32_t k1, int B, int nCB, int ncb_all, int nib,
                                                      uint32_t* __restrict__ sT) {
    const int lane = threadIdx.x & 63, il = lane & 31, half = lane >> 5;
    const int cb = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;   // a half-wave per 32-expert block, looping over the K blocks
    if (cb >= ncb_all) return;                                          // (wave-uniform per half: transpose32 shuffles stay inside a half)
    const int c = cb * 32 + il;
    uint32_t* dst = sT + (int64_t)(c >> 8) * nib * 256 + (c & 255);
    const int ib_end = min(nib, (int)(blockIdx.y + 1) * SWT_IB);      // grid.y: chunks of SWT_IB K blocks (a wide minibatch over few expert blocks still fills the chip)
    for (int ib = blockIdx.y * SWT_IB; ib < ib_end; ++ib) {
        const int i = ib * 32 + il;
        uint32_t w = 0u;
        if (i < B) {
            if (so_inj) { if (cb < nCB) w = sbits[(int64_t)i * nCB + cb]; }
            else w = sign_word(k0, k1, (uint32_t)i, (uint32_t)cb);
        }
        dst[(int64_t)ib * 256] = transpose32(w, il);
    }
}

// Epilogue of the output layer's dW for N consecutive hidden units of one expert (idx0 = expert * H + first unit), from the finished sums s1 = dz^T h and
// s2 = (dz s_out)^T (h s_in): Flipout chain rule for rho (eps recovered as Wp / sigma) + the KL terms, then either the gradients or, with ADAM, the update in place.
// operands of that epilogue for one run of N hidden units: what it reads from memory, so that a caller can have the next runs' loads in flight (k_out_dw_q)
template <int N> struct DwOps { float rho[N], mu[N], m1[N], v1[N], m2[N], v2[N]; };
template <bool BAYES, bool ADAM, int N>
__device__ __forceinline__ void dw_ops_load(const DwArgs& p, int64_t idx0, DwOps<N>& o) {
    if (BAYES) { ld_vec<N>((ADAM ? p.w_rho : p.rho) + idx0, o.rho); ld_vec<N>((ADAM ? p.w_mu : p.mu) + idx0, o.mu); }
    else if (ADAM) ld_vec<N>(p.w_mu + idx0, o.mu);
    if (ADAM) { ld_vec<N>(p.m_mu + idx0, o.m1); ld_vec<N>(p.v_mu + idx0, o.v1); if (BAYES) { ld_vec<N>(p.m_rho + idx0, o.m2); ld_vec<N>(p.v_rho + idx0, o.v2); } }
}
template <bool BAYES, bool ADAM, int N>
__device__ __forceinline__ void dw_finish_ops(const DwArgs& p, int64_t idx0, const float (&s1)[N], const float (&s2)[N], DwOps<N>& o, float& nx_kl, float& nx_amax) {
    static_assert(N == 4, "one Philox quad per call");
    float o_mu[N], o_rho[N], z[4] = {0.f, 0.f, 0.f, 0.f};
    // d(sigma eps)/d rho = eps sigmoid(rho): eps of THIS step drawn again from its counter (or read from the injected tensor) - round 3 recovered it as wp / sigma from an
    // f32 copy of sigma eps that every step wrote (4 B) and read (4 B) per element for this one use
#if defined(DWQ_ABL) && (DWQ_ABL & 1)      // timing-only builds: without this step's eps drawn again
    z[0] = z[1] = z[2] = z[3] = 1.f;
#else
    if (BAYES) normal4(p.cur_eps, idx0 >> 2, idx0, INT64_MAX, z);
#endif
#pragma unroll
    for (int jt = 0; jt < N; ++jt) {
        float gm = s1[jt], gr = 0.f, pm = 0.f, rh = 0.f;
        if (BAYES) {
            rh = o.rho[jt];
            pm = o.mu[jt];
            const float e = __builtin_amdgcn_exp2f(fminf(rh, 80.f) * 1.44269504f), t = 1.f + e;
            const float sigma = e < 0.015625f ? e * (1.f - e * (0.5f - e * (0.33333333f - 0.25f * e))) : __builtin_amdgcn_logf(t) * 0.69314718f;
            const float sg = e * __builtin_amdgcn_rcpf(t), isig = __builtin_amdgcn_rcpf(sigma);
            gm += p.klw * pm;
            gr = s2[jt] * z[jt] * sg + p.klw * (sigma - isig) * sg;
        } else if (ADAM) pm = o.mu[jt];
        o_mu[jt] = gm; o_rho[jt] = gr;
        if (ADAM) {   // in place: the updated parameter goes where the gradient would have gone
            o_mu[jt] = adam_update(pm, gm, o.m1[jt], o.v1[jt], p.lr_over_bc1, p.b1, p.b2, p.eps, p.bc2_sqrt);
            if (BAYES) o_rho[jt] = adam_update(rh, gr, o.m2[jt], o.v2[jt], p.lr_over_bc1, p.b1, p.b2, p.eps, p.bc2_sqrt);
        }
    }
    if (!ADAM) { st_vec<N>(p.g_mu + idx0, o_mu); if (BAYES) st_vec<N>(p.g_rho + idx0, o_rho); }
    else {
        st_vec<N>(p.w_mu + idx0, o_mu); st_vec<N>(p.m_mu + idx0, o.m1); st_vec<N>(p.v_mu + idx0, o.v1);
        if (BAYES) { st_vec<N>(p.w_rho + idx0, o_rho); st_vec<N>(p.m_rho + idx0, o.m2); st_vec<N>(p.v_rho + idx0, o.v2); }
        if constexpr (N == 4) { if (p.produce) { if constexpr (BAYES) dw_produce_next(p, idx0, o_mu, o_rho, nx_kl, nx_amax); else dw_produce_next_fnn(p, idx0, o_mu, nx_amax); } }
    }
}
template <bool BAYES, bool ADAM, int N>
__device__ __forceinline__ void dw_finish_vec(const DwArgs& p, int64_t idx0, const float (&s1)[N], const float (&s2)[N], float& nx_kl, float& nx_amax) {
    DwOps<N> o;
    dw_ops_load<BAYES, ADAM, N>(p, idx0, o);
    dw_finish_ops<BAYES, ADAM, N>(p, idx0, s1, s2, o, nx_kl, nx_amax);
}

// split-K dW: sum of the K ranges' partial slabs, then the epilogue (one thread per four hidden units; the first M threads also finish the bias gradients)
template <bool BAYES, bool ADAM>
__global__ __launch_bounds__(256) void k_out_dw_finish(DwArgs p) {
    if (p.rflag && *p.rflag) return;           // an operand left the fp16 window: the exact-f32 dW kernel (unsplit, own epilogue) ran instead
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, il0 = q * 4, idx0 = il0 + (int64_t)p.part_row0 * 128;   // il0: offset inside the slabs (the launched tile range)
    const int64_t Mp = p.slab / 128;
    if (q < Mp && q + p.part_row0 < p.M) {
        const float* pb = p.part + (int64_t)p.ksplit * 2 * p.slab;
        float b1 = 0.f, b2 = 0.f;
        for (int s = 0; s < p.ksplit; ++s) { b1 += pb[(int64_t)(s * 2) * Mp + q]; if (BAYES) b2 += pb[(int64_t)(s * 2 + 1) * Mp + q]; }
        p.g_b[q + p.part_row0] = b1; if (BAYES) p.g_bp[q + p.part_row0] = b2;
    }
    float nx_kl = 0.f, nx_amax = 0.f;
    const bool produce = ADAM && p.produce;
    const bool live = il0 < p.slab && idx0 < (int64_t)p.M * 128;
    if (!live && !produce) return;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
    for (int s = 0; s < p.ksplit; ++s) {
        float t[4];
        ld_vec<4>(p.part + (int64_t)(s * 2) * p.slab + il0, t);
#pragma unroll
        for (int k = 0; k < 4; ++k) s1[k] += t[k];
        if (BAYES) {
            ld_vec<4>(p.part + (int64_t)(s * 2 + 1) * p.slab + il0, t);
#pragma unroll
            for (int k = 0; k < 4; ++k) s2[k] += t[k];
        }
    }
    dw_finish_vec<BAYES, ADAM, 4>(p, idx0, s1, s2, nx_kl, nx_amax);
    }
    if (produce) { __shared__ double red[4]; dw_produce_finish<BAYES>(p, nx_kl, nx_amax, red, 4); }
}

// ------------------------------------------------------------------------------------------------
// s_in sign words of the dW kernel k_out_dw_q: sinT[K block ib][hidden unit j] = the signs of (batch rows 32 ib .. 32 ib + 31, j), in the bit order that kernel's B
// fragments take their masks from - the fragment (k step ks, lane half hf) of a hidden unit holds rows 8 g .. 8 g + 7, g = 2 ks + hf, dword q = rows 8 g + 2 q (low
// fp16) and 8 g + 2 q + 1 (high): row 8 g + 2 q sits at bit 4 g + 3 - q, row 8 g + 2 q + 1 at bit 16 + 4 g + 3 - q, so that (word << (12 - 4 g + q)) & 0x80008000
// is the XOR mask of dword q.
__host__ __device__ __forceinline__ int sin_word_bit(int r) { const int g = r >> 3, q = (r & 7) >> 1, o = r & 1; return 16 * o + 4 * g + 3 - q; }
__global__ void k_sin_words_T(const uint32_t* __restrict__ sinbits, int Bpad, uint32_t* __restrict__ sinT) {   // sinbits[i][4]: bit j & 31 of word j >> 5 (k_prep_h)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (Bpad >> 5) * 128) return;
    const int ib = t >> 7, j = t & 127;
    uint32_t w = 0u;
    for (int r = 0; r < 32; ++r) w |= ((sinbits[(int64_t)(ib * 32 + r) * 4 + (j >> 5)] >> (j & 31)) & 1u) << sin_word_bit(r);
    sinT[t] = w;
}

// dW + Adam (+ the next step's operands) of the fp16x3 training step, second form (round 4).  k_out_dw_p2's tile is a main loop (LDS-fed MFMAs, HBM a quarter used)
// followed by an HBM-bound epilogue (64 B per mu / rho pair, the matrix pipe idle), and with one 130 KB-LDS workgroup per CU neither hides the other: 0.33 + 0.32 ms.
// Here a workgroup is HALF of that tile - 4 waves x 32 experts, the same 32-row K blocks, the same MFMA sequence per accumulator (bit-identical sums) - and its LDS
// stage is 33 KB instead of 65: the A half (16 KB of packed dz), the h planes (16 KB) and two 512-B sign images; the planes of h * s_in are NOT staged - the
// signed B fragment is the plain one XOR a mask made from the transposed s_in words (k_sin_words_T: 2 vector instructions per mask dword), which also halves the
// LDS bytes a wave reads per MFMA (20 KB instead of 36 KB per K block).  Two such workgroups fit a CU (2 x 66 KB of LDS, 2 x 4 waves x 256 registers); the second
// one of every CU starts half a tile late (p.stagger), so that from then on one is in its main loop while the other streams its epilogue: matrix pipe and HBM at
// the same time.  (A workgroup that exits is replaced at once, which keeps the offset.)  The epilogue keeps the operand rows of DW_EPI_PD accumulator rows in flight
// ahead of the row it works on: four waves must sustain what eight did.
#ifndef DW_EPI_PD
#define DW_EPI_PD 1
#endif
#ifndef DWQ_PRIO
#define DWQ_PRIO 2    // s_setprio of the main loop (0: 0.634, 2: 0.604 ms on one box; 3 = 2)
#endif
#ifndef DWQ_BRING
#define DWQ_BRING 2
#endif
#ifndef DWQ_PPG
#define DWQ_PPG 3     // DMA pieces of the next K block per half-group of 3 MFMAs (with DWQ_PRIO 2, relative to k_out_dw_p2 on the same box: 1 -5.0 %, 2 -5.6 %, 3 -6.7 %)
#endif
constexpr int QW = 4;            // waves per workgroup
constexpr int QTC = 32 * QW;     // experts per workgroup
// SPLIT (round 5): the split-K form for few expert tiles - a narrow expert shard under a wide minibatch: workgroup = (tile, K range), raw partial sums into the slabs
// k_out_dw_finish adds up (the layout k_out_dw_p2's split launch writes: an accumulator's element is addressed by its expert row and hidden unit, not by the kernel).
template <bool BAYES, bool ADAM, bool STAMP = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * QW, 2) void k_out_dw_q(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = 128, NJT = 4, NP = 2;
    constexpr int TA = QTC * 32 * 4;              // packed dz [128 experts][32 rows] dwords, 16-byte chunks XOR-swizzled ((row>>1)&7)
    constexpr int PLANE = H * 64;                 // [H slots][32 rows] fp16
    constexpr int TB = NP * PLANE;                // the two planes of h
    constexpr int SRC_TB = (BAYES ? 2 : 1) * TB;  // hb holds the planes of h * s_in behind them (k_out_dw_p2's operands)
    constexpr int TW = BAYES ? 1024 : 0;          // s_out words of the tile's experts (512 B), s_in words of the hidden units (512 B)
    constexpr int STAGE = TA + TB + TW;
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, half = lane >> 5;
    if (p.rmode == 1 && __builtin_nontemporal_load(p.rflag) != 0) return;   // the step runs in exact f32: the kernel launched behind this one
    const int ntile_s = SPLIT ? (int)gridDim.x / p.ksplit : 0;
    const int ksi = SPLIT ? (int)blockIdx.x / ntile_s : 0;                  // which K range (the splits of one tile sit ntile_s workgroups apart)
    const int c0 = (p.wg_begin + (SPLIT ? (int)blockIdx.x % ntile_s : (int)blockIdx.x)) * QTC;
    const int crow = wave * 32 + il, c = c0 + crow;
    const int nib = p.Bpad / 32;
    const int ib0 = SPLIT ? (int)((int64_t)ksi * nib / p.ksplit) : 0, ib1 = SPLIT ? (int)((int64_t)(ksi + 1) * nib / p.ksplit) : nib;
    const uint32_t smem_base = lds_addr(smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* hb = reinterpret_cast<const char*>(p.hb);
    unsigned long long st_t[4] = {0, 0, 0, 0};
    if (STAMP) st_t[0] = __builtin_amdgcn_s_memrealtime();
    uint32_t lds_base = 0u;
    if (STAMP || (p.stagger > 0 && (int)blockIdx.x < p.ntile)) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC, 0, 8)" : "=s"(lds_base));
    if (p.stagger > 0 && (int)blockIdx.x < p.ntile) {
        // first round of workgroups: the one that shares its CU with an earlier one (its LDS allocation does not start at 0) starts late
        if (lds_base != 0u) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)p.stagger) __builtin_amdgcn_s_sleep(64);
        }
    }

    const h2_t ones = {(_Float16)1.0f, (_Float16)1.0f};
    constexpr int NA = TA / 1024 / QW, NB = TB / 1024 / QW;   // DMA pieces per wave and K block: 4 of the dz tile, 4 of the planes (+ the sign words with the last one)
    const float* dz_tile = p.dzT + (int64_t)(c0 >> 8) * nib * 8192 + ((c0 & 255) << 5);
    const uint32_t* sw_src = BAYES ? (half == 0 ? p.sT + (int64_t)(c0 >> 8) * nib * 256 + (c0 & 255) + il * 4 : p.sinT + il * 4) : nullptr;
    const int sw_step = half == 0 ? 256 : 128;    // words per K block in the two images
    auto stage_piece = [&](int ib, int buf, int n) {
        const uint32_t sb = smem_base + buf * STAGE;
        if (n < NA) {
            const int inst = wave_u * NA + n;
            const int row = inst * 8 + (lane >> 3), pch = lane & 7;
            const int q = pch ^ ((row >> 1) & 7);
            glds16(dz_tile + (int64_t)ib * 8192 + row * 32 + 4 * q, sb + inst * 1024);
        } else {
            const char* src = hb + (size_t)ib * SRC_TB;
            const int inst = wave_u * NB + (n - NA);
            const int pos = inst * 1024 + lane * 16;
            const int j = (pos % PLANE) >> 6, cd = (pos >> 4) & 3;
            glds16(src + (pos & ~63) + 16 * (cd ^ ((j >> 2) & 3)), sb + TA + inst * 1024);
            // the K block's sign words, 1 KiB: fetched by EVERY wave (the same bytes to the same place; a wave-uniform branch would split the K block's body)
            if (BAYES && n == NA + NB - 1) glds16(sw_src + (int64_t)ib * sw_step, sb + TA + TB);
        }
    };
    float nx_kl = 0.f, nx_amax = 0.f;
    unsigned long long st_c[3] = {0, 0, 0}, c_prev = 0;     // STAMP: shader-clock sums over the K blocks - body, DMA wait, barrier
    auto cyc = [&]() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
#pragma unroll
    for (int n = 0; n < NA + NB; ++n) stage_piece(ib0, 0, n);
    f32x16 acc1[NJT], acc2[NJT];
#pragma unroll
    for (int j = 0; j < NJT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[j][r] = 0.f; acc2[j][r] = 0.f; }
    float sum1 = 0.f, sum2 = 0.f;
    const int shl0 = 12 - 4 * half, shl1 = 4 - 4 * half;      // mask shifts of k step 0 / 1 (fragment row group g = 2 ks + half)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto k_block = [&](int ib, auto more_c) {
        constexpr bool MORE = decltype(more_c)::value;
        const int buf = (ib - ib0) & 1;
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + TA;
        uint32_t word = 0u;
        u32x4 iw = {0u, 0u, 0u, 0u};
        if (BAYES) { word = *reinterpret_cast<const uint32_t*>(sA + TA + TB + crow * 4); iw = *reinterpret_cast<const u32x4*>(sA + TA + TB + 512 + il * 16); }
        constexpr int NG = 2 * NJT;                 // (ks, jt) groups: one B fragment each, used plain and (Flipout) signed
        const char* bbase = sB + il * 64;
        const int swz = (il >> 2) & 3;
        auto load_b = [&](int g, u32x4 (&dst)[3]) {
            const int ks = g / NJT, jt = g % NJT;
            const char* bp = bbase + jt * 2048 + 16 * ((2 * ks + half) ^ swz);
#pragma unroll
            for (int q = 0; q < NP; ++q) dst[q] = *reinterpret_cast<const u32x4*>(bp + q * PLANE);
            dst[2] = u32x4{0u, 0u, 0u, 0u};
        };
        u32x4 a[2][3], as[2][3];
        auto prep_a = [&](int ks) {      // rows 16 ks + 8 half .. + 7 of this lane's expert: 8 packed dwords -> the hi and the lo plane fragment (k_out_dw_p2)
            const int ch = 4 * ks + 2 * half, sw = (crow >> 1) & 7;
            const u32x4 lo = *reinterpret_cast<const u32x4*>(sA + crow * 128 + 16 * (ch ^ sw));
            const u32x4 hi = *reinterpret_cast<const u32x4*>(sA + crow * 128 + 16 * ((ch + 1) ^ sw));
            const uint32_t x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const uint32_t w8 = word >> (ks * 16 + half * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t p1 = __builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x05040100u), p2 = __builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x07060302u);
                a[ks][0][q] = p1; a[ks][1][q] = p2; a[ks][2][q] = 0u;
                sum1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, p1), ones, sum1, false);
                sum1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, p2), ones, sum1, false);
                if (BAYES) {
                    const uint32_t m = ((w8 << (15 - 2 * q)) & 0x8000u) | ((w8 << (30 - 2 * q)) & 0x80000000u);
                    const uint32_t s1 = p1 ^ m, s2 = p2 ^ m;
                    as[ks][0][q] = s1; as[ks][1][q] = s2; as[ks][2][q] = 0u;
                    sum2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, s1), ones, sum2, false);
                    sum2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, s2), ones, sum2, false);
                }
            }
        };
        constexpr int BR = DWQ_BRING;           // B fragments in flight: the fragment of group g + BR - 1 is fetched while group g's MFMAs run
        u32x4 bq[BR][3], bs[3];
#pragma unroll
        for (int g = 0; g < BR - 1; ++g) load_b(g, bq[g]);
        prep_a(0);
        int piece = 0;
        auto dma = [&]() {
            if (MORE) {
#pragma unroll
                for (int q = 0; q < DWQ_PPG; ++q) { if (piece < NA + NB) stage_piece(ib + 1, buf ^ 1, piece); ++piece; }
            }
        };
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + BR - 1 < NG) load_b(g + BR - 1, bq[(g + BR - 1) % BR]);
            asm volatile("" ::: "memory");
            const int ks = g / NJT, jt = g % NJT;
            acc1[jt] = mfma_np<NP>(a[ks], bq[g % BR], acc1[jt]);
            dma();
            if (BAYES) {
                const uint32_t v = iw[jt] << (ks ? shl1 : shl0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t m = (v << q) & 0x80008000u;
                    bs[0][q] = bq[g % BR][0][q] ^ m; bs[1][q] = bq[g % BR][1][q] ^ m;
                }
                bs[2] = u32x4{0u, 0u, 0u, 0u};
                acc2[jt] = mfma_np<NP>(as[ks], bs, acc2[jt]);
                dma();
            }
            if (g == 0) prep_a(1);
        }
        if (STAMP) { st_c[0] += cyc() - c_prev; c_prev = cyc(); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (STAMP) { st_c[1] += cyc() - c_prev; c_prev = cyc(); }
        __syncthreads();
        if (STAMP) { st_c[2] += cyc() - c_prev; c_prev = cyc(); }
    };
    if (STAMP) { st_t[1] = __builtin_amdgcn_s_memrealtime(); c_prev = cyc(); }
    if (DWQ_PRIO) __builtin_amdgcn_s_setprio(DWQ_PRIO);   // the main loop is the latency-bound one of a SIMD's two waves (one in-order MFMA stream); its partner streams an epilogue
    for (int ib = ib0; ib < ib1 - 1; ++ib) k_block(ib, std::true_type{});
    k_block(ib1 - 1, std::false_type{});
    if (DWQ_PRIO) __builtin_amdgcn_s_setprio(0);
    if (STAMP) st_t[2] = __builtin_amdgcn_s_memrealtime();
    sum1 += __shfl_xor(sum1, 32, 64);
    sum2 += __shfl_xor(sum2, 32, 64);
    const float inv_a = 1.f / p.a_scale;
    if constexpr (SPLIT) {    // raw partial sums of this K range (bias sums behind the slabs); k_out_dw_finish adds the ranges and runs the epilogue
        if (half == 0 && c < p.M) {
            float* pb = p.part + (int64_t)p.ksplit * 2 * p.slab; const int64_t Mp = p.slab / 128; const int cl = c - p.part_row0;
            pb[(int64_t)(ksi * 2) * Mp + cl] = sum1 * inv_a; if (BAYES) pb[(int64_t)(ksi * 2 + 1) * Mp + cl] = sum2 * inv_a;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cr = c0 + wave * 32 + rowmap(r, half);
            if (cr >= p.M) continue;
            const int64_t il0 = (int64_t)(cr - p.part_row0) * H + NJT * il;
            float s1[NJT], s2[NJT];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) { s1[jt] = acc1[jt][r] * p.unscale; s2[jt] = acc2[jt][r] * p.unscale; }
            st_vec<NJT>(p.part + (int64_t)(ksi * 2) * p.slab + il0, s1);
            if (BAYES) st_vec<NJT>(p.part + (int64_t)(ksi * 2 + 1) * p.slab + il0, s2);
        }
        return;
    }
    if (half == 0 && c < p.M) { p.g_b[c] = sum1 * inv_a; if (BAYES) p.g_bp[c] = sum2 * inv_a; }

    // epilogue: accumulator row r of lane (il, half) = expert c0 + 32 wave + rowmap(r, half), hidden units 4 il .. 4 il + 3
    constexpr int PD = DW_EPI_PD;
    DwOps<NJT> ops[PD + 1];
    auto row_idx = [&](int r, int64_t& idx0) -> bool {
        const int cr = c0 + wave * 32 + rowmap(r, half);
        idx0 = (int64_t)cr * H + NJT * il;
        return cr < p.M;
    };
    static_for<0, PD>([&](auto rc) { constexpr int r = decltype(rc)::value; int64_t idx0; if (row_idx(r, idx0)) dw_ops_load<BAYES, ADAM, NJT>(p, idx0, ops[r % (PD + 1)]); });
    static_for<0, 16>([&](auto rc) {      // (a compile-time r: left as a loop hipcc keeps it rolled and indexes the operand sets through scratch)
        constexpr int r = decltype(rc)::value;
        if constexpr (r + PD < 16) { int64_t idn; if (row_idx(r + PD, idn)) dw_ops_load<BAYES, ADAM, NJT>(p, idn, ops[(r + PD) % (PD + 1)]); }
        int64_t idx0;
        if (row_idx(r, idx0)) {
            float s1[NJT], s2[NJT];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) { s1[jt] = acc1[jt][r] * p.unscale; s2[jt] = acc2[jt][r] * p.unscale; }
            dw_finish_ops<BAYES, ADAM, NJT>(p, idx0, s1, s2, ops[r % (PD + 1)], nx_kl, nx_amax);
        }
    });
    if (STAMP && p.stamps && lane == 0) {   // diagnostics (NTF_DW_STAMP): per wave - entry, main loop begin / end, exit (100 MHz ticks), where it ran, its LDS base
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_t[3] = __builtin_amdgcn_s_memrealtime();
        uint32_t hw_id, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = p.stamps + ((int64_t)blockIdx.x * QW + wave) * 10;
        o[0] = st_t[0]; o[1] = st_t[1]; o[2] = st_t[2]; o[3] = st_t[3]; o[4] = ((unsigned long long)xcc << 32) | hw_id; o[5] = lds_base;
        o[6] = st_c[0]; o[7] = st_c[1]; o[8] = st_c[2]; o[9] = 0;
    }
    if (ADAM && p.produce) dw_produce_finish<BAYES>(p, nx_kl, nx_amax, reinterpret_cast<double*>(smem + 2 * STAGE), QW);   // (scratch behind the stages)
}

void launch_fused_out_dw(hipStream_t st, const FusedDw& f) {
    const Geom g = geom(f.B, f.M);
    const WsLayout w = ws_layout(f.B, f.H, f.M);
    char* ws = static_cast<char*>(f.ws);
    DwArgs a;
    a.B = f.B; a.M = f.M; a.Bpad = g.Bpad; a.dzT = f.dzT; a.h = reinterpret_cast<const float*>(ws + w.hz); a.hs = reinterpret_cast<const float*>(ws + w.hs);
    a.mu = f.mu; a.rho = f.rho; a.wp = f.wp; a.sbits = reinterpret_cast<const uint32_t*>(ws + w.sbits); a.nCB = g.nCB;
    a.so_k0 = f.s_out.k0; a.so_k1 = f.s_out.k1; a.so_inj = f.s_out_inj;
    a.g_mu = f.g_mu; a.g_rho = f.g_rho; a.g_b = f.g_b; a.g_bp = f.g_bp; a.klw = f.klw;
    a.w_mu = f.w_mu; a.w_rho = f.w_rho; a.m_mu = f.m_mu; a.v_mu = f.v_mu; a.m_rho = f.m_rho; a.v_rho = f.v_rho;
    a.lr_over_bc1 = f.lr_over_bc1; a.b1 = f.b1; a.b2 = f.b2; a.eps = f.eps; a.bc2_sqrt = f.bc2_sqrt;
    const int total = (f.M + DW_TC - 1) / DW_TC;
    const int grid = f.wg_count > 0 ? std::min(f.wg_count, total - f.wg_begin) : total;
    a.wg_begin = f.wg_count > 0 ? f.wg_begin : 0;
    if (grid <= 0) return;
    a.hb = reinterpret_cast<const uint16_t*>(ws + w.hb);
    a.rflag = f.rflag; a.rmode = 0;
    a.produce = (f.produce && f.adam && f.H == 128) ? 1 : 0;      // (Fnn: the planes of the updated mu and the range flag only)
    a.cur_eps = f.cur_eps; a.lean = (a.produce && f.lean) ? 1 : 0;
    a.nx_eps = f.nx_eps; a.nx_wp = f.nx_wp; a.nx_pl_wp = f.nx_pl_wp; a.nx_pl_mu = f.nx_pl_mu; a.nx_pscale = f.nx_pscale; a.nx_klw = f.nx_klw; a.nx_kl = f.nx_kl; a.nx_rflag = f.nx_rflag;
    a.ntile = 0; a.stagger = 0; a.stamps = nullptr;
    const bool guard = f.bf16x6 && f.np == 2 && f.rflag != nullptr;
    a.sT = reinterpret_cast<const uint32_t*>(ws + w.sbitsT);
    a.ksplit = 1; a.part = nullptr; a.slab = 0; a.part_row0 = 0;
    if (f.fallback_only) {   // the exact-f32 kernel of a step whose split-product launches (several, e.g. the tail split of a whole step) were issued with no_fallback
        if (!guard) return;
        a.rmode = 2; a.a_scale = f.a_scale; a.unscale = 1.f / (f.a_scale * f.h_scale);
        goto exact_f32;
    }
    if (f.bf16x6 && f.np == 2 && f.dz_packed) {   // fp16x3 step, H = 128: the forward kernel left packed plane pairs in dzT
        a.a_scale = f.a_scale; a.unscale = 1.f / (f.a_scale * f.h_scale); a.rmode = guard ? 1 : 0;
        const int ks = (f.ksplit > 1 && f.part) ? std::min(f.ksplit, std::max(1, g.Bpad / 32)) : 1;
        if (ks > 1) {   // few expert tiles (a narrow expert shard under a wide minibatch): every half-tile's K range split over ks workgroups (k_out_dw_q<.., SPLIT>), k_out_dw_finish adds the parts and runs the epilogue
            a.sinT = reinterpret_cast<const uint32_t*>(ws + w.sinT);
            const int total_q = (f.M + QTC - 1) / QTC, qb = f.wg_count > 0 ? 2 * f.wg_begin : 0;
            const int qgrid = f.wg_count > 0 ? std::min(2 * f.wg_count, total_q - qb) : total_q;
            const int wg256 = a.wg_begin;
            a.wg_begin = qb;
            a.ksplit = ks; a.part = f.part; a.slab = (int64_t)qgrid * QTC * 128; a.part_row0 = qb * QTC;
            const size_t ldsq = 2 * ((size_t)QTC * 128 + 2 * 128 * 64 + (f.bayes ? 1024 : 0)) + 64;
            const int64_t nq = (int64_t)qgrid * QTC * 128 / 4;
#define NTF_DWQS(BY, AD) do { auto kf = k_out_dw_q<BY, false, false, true>;                                                    \
            set_max_lds(reinterpret_cast<const void*>(kf), (int)ldsq);     \
            hipLaunchKernelGGL(kf, dim3(qgrid * ks), dim3(64 * QW), ldsq, st, a);                                              \
            hipLaunchKernelGGL((k_out_dw_finish<BY, AD>), dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, a); } while (0)
            if (f.bayes) { if (f.adam) NTF_DWQS(true, true); else NTF_DWQS(true, false); } else { if (f.adam) NTF_DWQS(false, true); else NTF_DWQS(false, false); }
#undef NTF_DWQS
            if (!guard || f.no_fallback) return;
            a.rmode = 2; a.ksplit = 1; a.wg_begin = wg256; a.part_row0 = 0;
            goto exact_f32;
        }
        {   // two half-tile workgroups per CU, the epilogue of one beside the main loop of the other (k_out_dw_q)
            a.sinT = reinterpret_cast<const uint32_t*>(ws + w.sinT);
            const int total_q = (f.M + QTC - 1) / QTC, qb = f.wg_count > 0 ? 2 * f.wg_begin : 0;
            const int qgrid = f.wg_count > 0 ? std::min(2 * f.wg_count, total_q - qb) : total_q;
            const int wg256 = a.wg_begin;
            a.wg_begin = qb;
            static int n_cu = 0;
            if (!n_cu) { int dev = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
            static const int stagger_q = getenv("NTF_DW_STAGGER") ? atoi(getenv("NTF_DW_STAGGER")) : -1;   // 10 ns ticks; default: half of a tile's main loop + epilogue
            const int nib = g.Bpad / 32;
            a.ntile = 2 * n_cu;
            a.stagger = (f.adam && qgrid > n_cu) ? (stagger_q >= 0 ? stagger_q : (nib * 140 + 4500) / 2) : 0;
            const size_t ldsq = 2 * ((size_t)QTC * 128 + 2 * 128 * 64 + (f.bayes ? 1024 : 0)) + 64;
#define NTF_DWQ(BY, AD) do { auto kf = k_out_dw_q<BY, AD>;                                                                     \
            set_max_lds(reinterpret_cast<const void*>(kf), (int)ldsq);     \
            hipLaunchKernelGGL(kf, dim3(qgrid), dim3(64 * QW), ldsq, st, a); } while (0)
#ifdef NTF_DIAG
            static const char* stamp_file = getenv("NTF_DW_STAMP_FILE");   // -DNTF_DIAG builds: the 30th launch's per-wave stamps, raw (10 x u64 per wave), to this file (profiles/dw_stamps.py)
#else
            static const char* stamp_file = nullptr;
#endif
            if (stamp_file && f.bayes && f.adam) {
                static unsigned long long* d_st = nullptr; static int n_launch = 0;
                if (!d_st) hipMalloc(&d_st, (size_t)qgrid * QW * 10 * 8);
                a.stamps = d_st;
                auto kf = k_out_dw_q<true, true, true>;
                set_max_lds(reinterpret_cast<const void*>(kf), (int)ldsq);
                hipLaunchKernelGGL(kf, dim3(qgrid), dim3(64 * QW), ldsq, st, a);
                if (++n_launch == 30) {
                    std::vector<unsigned long long> hst((size_t)qgrid * QW * 10);
                    hipStreamSynchronize(st); hipMemcpy(hst.data(), d_st, hst.size() * 8, hipMemcpyDeviceToHost);
                    if (FILE* fp = fopen(stamp_file, "wb")) { fwrite(hst.data(), 8, hst.size(), fp); fclose(fp); }
                }
            }
            else if (f.bayes) { if (f.adam) NTF_DWQ(true, true); else NTF_DWQ(true, false); } else { if (f.adam) NTF_DWQ(false, true); else NTF_DWQ(false, false); }
#undef NTF_DWQ
            if (!guard || f.no_fallback) return;
            a.rmode = 2; a.wg_begin = wg256; a.ntile = 0; a.stagger = 0;   // the exact-f32 kernel behind it runs only when the range flag is raised
            goto exact_f32;
        }
    }
    if (f.bf16x6) {
        constexpr int np = 2;
        a.rmode = guard ? 1 : 0;
        a.a_scale = np == 2 ? f.a_scale : 1.f; a.unscale = np == 2 ? 1.f / (f.a_scale * f.h_scale) : 1.f;
#define NTF_DWB2(HH, BY, AD, NPV) do { auto kf = k_out_dw_b6<HH, BY, AD, NPV>; const size_t lds = 2 * ((size_t)DW_TC * 128 + (size_t)(BY ? 2 : 1) * NPV * HH * 64); \
        set_max_lds(reinterpret_cast<const void*>(kf), (int)lds);                                      \
        hipLaunchKernelGGL(kf, dim3(grid), dim3(64 * DW_WAVES), lds, st, a); } while (0)
#define NTF_DWB1(HH, BY, AD) NTF_DWB2(HH, BY, AD, 2)
#define NTF_DWB(HH) do { if (f.bayes) { if (f.adam) NTF_DWB1(HH, true, true); else NTF_DWB1(HH, true, false); }                                            \
                         else { if (f.adam) NTF_DWB1(HH, false, true); else NTF_DWB1(HH, false, false); } } while (0)
        if (f.H == 128) NTF_DWB(128); else if (f.H == 64) NTF_DWB(64); else NTF_DWB(32);
#undef NTF_DWB
#undef NTF_DWB1
#undef NTF_DWB2
        if (!guard) return;
        a.rmode = 2;   // fall through: the exact-f32 kernel, which runs only when the range flag is raised
    }
exact_f32:
    a.ntile = grid;
#define NTF_DW1(HH, BY) do { const bool fb = a.rmode == 2;                                                                                    \
        auto kf = fb ? (f.adam ? k_out_dw_fallback<HH, BY, true> : k_out_dw_fallback<HH, BY, false>) : (f.adam ? k_out_dw<HH, BY, true> : k_out_dw<HH, BY, false>);   \
        const size_t lds = 2 * (DW_TC * 32 * 4 + (BY ? 2 : 1) * 32 * 4 * HH);                                                                \
        set_max_lds(reinterpret_cast<const void*>(kf), (int)lds);                      \
        hipLaunchKernelGGL(kf, dim3(fb ? std::min(grid, 256) : grid), dim3(64 * DW_WAVES), lds, st, a); } while (0)
#define NTF_DW(HH) do { if (f.bayes) NTF_DW1(HH, true); else NTF_DW1(HH, false); } while (0)
    if (f.H == 128) NTF_DW(128); else if (f.H == 64) NTF_DW(64); else NTF_DW(32);
#undef NTF_DW
}

// bf16 split planes of the hidden activations for the dW kernel (once per step, after launch_fused_out_fwd's phase 1)
// which: 1 = the transposed s_out words (they depend on the sign key only: the engine issues them on its auxiliary stream), 2 = the h planes, 3 = both
void launch_fused_prep_planes(hipStream_t st, int B, int H, int M, int bayes, void* ws_, int np, float h_scale, const SignSpec* s_out, int s_out_inj, int which) {
    const Geom g = geom(B, M);
    const WsLayout w = ws_layout(B, H, M);
    char* ws = static_cast<char*>(ws_);
    if (bayes && s_out && (which & 1)) {   // packed fp16x3 path: the dW kernel's s_out words, transposed once
        const int ncb_all = rup(M, DW_TC) / 32, nib = g.Bpad / 32;
        hipLaunchKernelGGL(k_sign_words_T, dim3((ncb_all + 7) / 8, (nib + SWT_IB - 1) / SWT_IB), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(ws + w.sbits), s_out_inj, s_out->k0, s_out->k1,
                           B, g.nCB, ncb_all, nib, reinterpret_cast<uint32_t*>(ws + w.sbitsT));
    }
    if (!(which & 2)) return;
    const int n = g.Bpad * H;
    hipLaunchKernelGGL(k_prep_planes_T, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<const float*>(ws + w.hz), reinterpret_cast<const float*>(ws + w.hs),
                       bayes, g.Bpad, H, np == 2 ? 2 : 3, h_scale, reinterpret_cast<uint16_t*>(ws + w.hb));
    if (bayes && H == 128 && np == 2)   // k_out_dw_q rebuilds the planes of h * s_in from these words
        hipLaunchKernelGGL(k_sin_words_T, dim3((g.Bpad / 32 * 128 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(ws + w.sinbits), g.Bpad, reinterpret_cast<uint32_t*>(ws + w.sinT));
}

}  // namespace ntf
