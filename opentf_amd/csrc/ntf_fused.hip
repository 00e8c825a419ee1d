// Fused output-layer kernels for gfx950.  The output layer ([B,H] x [H,M], M = number of experts, up to
// millions) carries >95 % of the step's FLOPs; it is computed by exact-f32 MFMA (v_mfma_f32_32x32x2_f32) in
// two kernels that never materialise dense labels or logits:
//
//   k_out_fwd   persistent, one workgroup per CU = (128 batch rows) x (a contiguous group of 64-expert tiles).
//               Per tile: zT = mu_tile . hT (+ Flipout perturbation product), bias, leaky_relu, BCE against the
//               all-negative dense labelling, d(loss)/dz -> dzT (transposed, one write), and immediately
//               dh += dz . mu_tile out of the same LDS tile and the accumulator registers (the 32x32 accumulator
//               layout is exactly the A-operand layout of the next MFMA when it sums over the accumulator's rows).
//               Weight tiles arrive by LDS-DMA (global_load_lds_dwordx4), double buffered, XOR-swizzled on the
//               source address so that both the ds_read_b128 fragment reads and the DMA writes are conflict free.
//   k_out_special  (ntf_special.hip since round 5) one wave per team: sparse fix-up for the positives / sampled negatives (labels stay CSR),
//               reduction of the per-group dh slabs, leaky_relu' mask.
//   k_out_dw    dmu = dzT . h, dWp = (dzT*s_out) . (h*s_in) with K = batch, bias gradients from the A operand,
//               Flipout rho-gradient + KL finalised in the epilogue.
#include "ntf_fused_common.h"
#include "ntf_special.h"

namespace ntf {

int fused_ldb(int B) { return rup(B, BM); }
int fused_dw_tile() { return DW_TC; }
int64_t fused_dw_part_floats(int M, int H, int ksplit) { return (int64_t)ksplit * 2 * ((int64_t)rup(M, DW_TC) * (H + 1)); }
int64_t fused_planes_elems(int M, int H) { return ((int64_t)M + 63) / 64 * 64 * H * 3; }   // rows padded to the 64-expert tile of k_out_fwd_h3x
bool fused_supported(int H) { return H == 32 || H == 64 || H == 128; }
int fused_loss_slots(int) { return 0; }

int64_t fused_dh_slab_floats(int, int H, int) { return (int64_t)4 * NCG_MAX * BM * H; }   // (x 4: up to four range launches of the forward kernel, each with its own column groups)

size_t fused_workspace_bytes(int B, int H, int M) { return ws_layout(B, H, M).total; }
FusedWsPtrs fused_ws_ptrs(void* ws_, int B, int H, int M) {
    const WsLayout w = ws_layout(B, H, M); char* ws = static_cast<char*>(ws_);
    FusedWsPtrs r; r.hz = reinterpret_cast<float*>(ws + w.hz); r.hs = reinterpret_cast<float*>(ws + w.hs); r.sinbits = reinterpret_cast<uint32_t*>(ws + w.sinbits);
    r.hb = reinterpret_cast<uint16_t*>(ws + w.hb); r.sinT = reinterpret_cast<uint32_t*>(ws + w.sinT); r.Bpad = rup(B, BM);
    return r;
}

// ------------------------------------------------------------------------------------------------
// sign bit images: sbits[i][cb] (bit c&31 of word cb = c>>5), its 32x32-block transpose sbitsT[c][i>>5],
// sinbits[i][j>>5] and hs = h * s_in
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_sign_bits(SignSpec so, int B, int M, int nCB, uint32_t* __restrict__ sbits) {
    // packed row image sbits[i][cb] of an INJECTED sign tensor (tests); native runs regenerate the words by hash in the kernels
    const int cb = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y;
    if (cb >= nCB) return;
    uint32_t w = 0;
    if (i < B) {
        if (so.inj) { for (int b = 0; b < 32; ++b) { const int c = cb * 32 + b; if (c < M && so.inj[(int64_t)i * so.ld + c] < 0.f) w |= 1u << b; } }
        else w = sign_word(so.k0, so.k1, (uint32_t)i, (uint32_t)cb);
    }
    sbits[(int64_t)i * nCB + cb] = w;
}

// hz = h zero-padded to Bpad rows (so that DMA / MFMA never touch stale rows); Flipout: sinbits and hs = h * s_in
// h_limit > 0 (fp16x3 arithmetic): an activation beyond the fp16 window of the scaled split raises *rflag (the step then runs on the f32 kernels)
__global__ void k_prep_h(SignSpec si, int bayes, const float* __restrict__ h, int B, int H, int Bpad, uint32_t* __restrict__ sinbits,
                         float* __restrict__ hs, float* __restrict__ hz, float h_limit, int* __restrict__ rflag) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int wpr = H / 32;
    if (t >= Bpad * wpr) return;
    const int i = t / wpr, wj = t % wpr;
    uint32_t w = 0;
    if (bayes && i < B) {
        if (si.inj) { for (int b = 0; b < 32; ++b) if (si.inj[(int64_t)i * si.ld + wj * 32 + b] < 0.f) w |= 1u << b; }
        else w = sign_word(si.k0, si.k1, (uint32_t)i, (uint32_t)wj);
    }
    if (bayes) sinbits[t] = w;
    for (int b = 0; b < 32; ++b) {
        const int j = wj * 32 + b;
        const float v = (i < B) ? h[(int64_t)i * H + j] : 0.f;
        if (rflag && !(fabsf(v) <= h_limit)) *rflag = 1;   // also catches NaN / inf
        hz[(int64_t)i * H + j] = v;
        if (bayes) hs[(int64_t)i * H + j] = ((w >> b) & 1u) ? -v : v;
    }
}

template <int H, bool BAYES, bool TRAIN, bool DH, bool INJ>
__device__ __forceinline__ void out_fwd_f32_body(const OutFwdArgs& p, char* smem) {
    constexpr int ROWB = 4 * H;              // bytes per weight row
    constexpr int TB = BN * ROWB;            // bytes per weight tile
    constexpr int NMAT = BAYES ? 2 : 1;
    constexpr int STAGE = NMAT * TB + 512;   // + two 64-float bias tiles
    constexpr int NE = H / 2;                // operand elements per lane (k split over the two lane halves)
    constexpr int NJT = H / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, half = lane >> 5;
    if (range_guard_skip(p.rflag, p.rmode, true)) return;

    // XCD-aware block -> (column group, row block): blocks that stream the same weight tiles share an XCD's L2
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int cg = bid / p.NRB, rb = bid % p.NRB;
    const int t_beg = (int)((int64_t)cg * p.T / p.NCG), t_end = (int)((int64_t)(cg + 1) * p.T / p.NCG);
    const int i0 = rb * BM + wave * 32;
    const int i = i0 + il;
    const bool row_ok = i < p.B;

    // B operand of zT = mu . hT : this lane's batch row, k = 8t + 4*half + e
    float hf[NE];
#pragma unroll
    for (int t = 0; t < H / 8; ++t) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        v = *reinterpret_cast<const float4*>(p.h + (int64_t)i * H + 8 * t + 4 * half);  // p.h = zero-padded copy
        hf[4 * t] = v.x; hf[4 * t + 1] = v.y; hf[4 * t + 2] = v.z; hf[4 * t + 3] = v.w;
    }
    // s_in sign words of this row, pre-shifted so that bit (8t + e) & 31 of word (8t + e) >> 5 is the sign of k = 8t + 4*half + e
    uint32_t sinw[NJT];
#pragma unroll
    for (int w = 0; w < NJT; ++w)
        sinw[w] = BAYES ? ((INJ ? p.sinbits[(int64_t)i * NJT + w] : (row_ok ? sign_word(p.si_k0, p.si_k1, (uint32_t)i, (uint32_t)w) : 0u)) >> (4 * half)) : 0u;
    const float rmask = row_ok ? 1.f : 0.f;                 // padding rows of the last row block contribute nothing
    const float rscale = row_ok ? p.tnw * p.inv_B : 0.f;

    // LDS byte offsets (relative to the stage base) that depend on the lane; everything else is an immediate.
    //   A fragment of zT: row u*32 + il, logical 16-byte chunk q = 2*tq + half, physical chunk q ^ swz(row)
    //   B operand of dh : row u*32 + rowmap(s, half), float j = jt*32 + il
    constexpr int QM = (H >= 64) ? 8 : 4;   // distinct values of the swizzled low chunk bits over tq
    int aoff[QM];
#pragma unroll
    for (int k = 0; k < QM; ++k) {
        if (H >= 64) aoff[k] = il * ROWB + 16 * (((2 * k + half) & 15) ^ (il & 15));
        else aoff[k] = il * ROWB + 16 * ((2 * k + half) ^ ((il >> 1) & 7));
    }
    // dh B operand: ONE wide read per (row, matrix) feeds all NJT column tiles: lane il owns hidden units j = NJT*il + jt,
    // i.e. NJT consecutive floats of the weight row (b128 for H=128).  Row = u*32 + (s&3) + 8*(s>>2) + 4*half.
    //   H >= 64: swz(row) = (s&3) + 4*half + 8*((s>>2)&1)  -> one offset per ((s&3), (s>>2)&1)
    int boff[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int b8 = 0; b8 < 2; ++b8) {
            const int q = (NJT * il) >> 2;
            if (H >= 64) boff[k][b8] = 4 * half * ROWB + 16 * (q ^ (k + 4 * half + 8 * b8)) + 4 * ((NJT * il) & 3);
            else boff[k][b8] = 0;  // H == 32: computed per step (few registers at stake there)
        }

    f32x16 Y1[NJT], Y2[NJT];
#pragma unroll
    for (int j = 0; j < NJT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { Y1[j][r] = 0.f; Y2[j][r] = 0.f; }
    LossAcc lacc;

    const uint32_t smem_base = lds_addr(smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto stage_tile = [&](int t, int buf) {
        const uint32_t sb = smem_base + buf * STAGE;
        const int c0 = t * BN;
        constexpr int PER_WAVE = TB / 4096;  // 1 KiB wave-instructions per wave per matrix
#pragma unroll
        for (int n = 0; n < PER_WAVE; ++n) {
            const int inst = wave_u * PER_WAVE + n;
            const int off = inst * 1024 + lane * 16;
            const int row = off / ROWB, pch = (off % ROWB) >> 4;
            const int q = pch ^ swz<H>(row);
            const int64_t grow = min(c0 + row, p.M - 1);
            glds16(p.mu + grow * H + 4 * q, sb + inst * 1024);
            if (BAYES) glds16(p.wp + grow * H + 4 * q, sb + TB + inst * 1024);
        }
        if (wave_u == 0) glds4(p.mu_b + min(c0 + lane, p.M - 1), sb + NMAT * TB);
        if (BAYES && wave_u == 1) glds4(p.bp + min(c0 + lane, p.M - 1), sb + NMAT * TB + 256);
    };

    // s_out sign words of (row i, tile t) = words 2t, 2t+1 of the row: regenerated by the hash (no memory traffic inside the
    // loop); only with injected signs (tests) are they fetched from the packed image.
    auto sign_words = [&](int t) -> uint2 {
        if (!BAYES || !row_ok) return make_uint2(0u, 0u);
        if (INJ) return *reinterpret_cast<const uint2*>(p.sbits + (int64_t)i * p.nCB + 2 * t);  // injected signs (tests): packed image
        return make_uint2(sign_word(p.so_k0, p.so_k1, (uint32_t)i, (uint32_t)(2 * t)), sign_word(p.so_k0, p.so_k1, (uint32_t)i, (uint32_t)(2 * t + 1)));
    };
    if (t_beg < t_end) stage_tile(t_beg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        const uint2 w2 = sign_words(t);
        if (t + 1 < t_end) stage_tile(t + 1, buf ^ 1);
        char* sb = smem + buf * STAGE;
        const int c0 = t * BN;
        if (c0 + BN > p.M) {  // ragged last tile (workgroup-uniform): mask the experts past M through their bias
            if (tid < BN && c0 + tid >= p.M) reinterpret_cast<float*>(sb + NMAT * TB)[tid] = -1e30f;
            __syncthreads();
        }

        // ---- zT tile: rows = experts (two 32-row sub-tiles u), cols = this wave's 32 batch rows.
        // Issue order (one basic block, software-interleaved so that the VALU epilogue runs under the MFMA pipe):
        //   X(u=0) | X(u=1) with epilogue(u=0) spread over its k-steps | dh(u=0) with epilogue(u=1) spread | dh(u=1)
        f32x16 X1[2], X2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { X1[u][r] = 0.f; X2[u][r] = 0.f; }
        uint32_t sw[2] = {0u, 0u};
        if (BAYES) { sw[0] = w2.x >> (4 * half); sw[1] = w2.y >> (4 * half); }
        // buffer descriptor over this tile's 64 rows of dzT (the tile base is wave-uniform; rows are Bpad floats)
        // this tile's experts inside their 256-expert dzT tile: base = block-uniform, the wave's 32-row K block and the lane in voffset
        constexpr int dz_row_bytes = 128;
        const __amdgpu_buffer_rsrc_t dz_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.dzT + dzt_tile_base(c0, p.Bpad), 0, ((p.Bpad >> 5) * 8192 - ((c0 & 255) << 5)) * 4, 0x00020000);
        const int dz_voff = ((i >> 5) * 8192 + 4 * half * 32 + (i & 31)) * 4;
        const float* bias_mu = reinterpret_cast<const float*>(sb + NMAT * TB) + 4 * half;
        const float* bias_p = reinterpret_cast<const float*>(sb + NMAT * TB + 256) + 4 * half;

        // operand fetch and MFMA issue are separate so that the fetch for group g+1 can be issued before the MFMAs of group g
        // (one wave per SIMD: nobody else hides the LDS latency)
        auto x_load = [&](int u, int tq, float4& a, float4& aw) {
            const int imm = u * 32 * ROWB + ((H >= 64) ? ((2 * tq) & ~15) * 16 : 0);
            const char* ap = sb + aoff[tq % QM] + imm;
            a = *reinterpret_cast<const float4*>(ap);
            if (BAYES) aw = *reinterpret_cast<const float4*>(ap + TB);
        };
        auto x_mma = [&](int u, int tq, const float4& a, const float4& aw) {   // 4 k-pairs of sub-tile u
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float awv[4] = {aw.x, aw.y, aw.z, aw.w};
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int e = 4 * tq + e4;
                X1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e4], hf[e], X1[u], 0, 0, 0);
                if (BAYES) {
                    const int kb = 8 * tq + e4;  // k without the lane half
                    const uint32_t m = (sinw[kb >> 5] << (31 - (kb & 31))) & 0x80000000u;
                    X2[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(awv[e4], __uint_as_float(__float_as_uint(hf[e]) ^ m), X2[u], 0, 0, 0);
                }
            }
        };
        auto epilogue = [&](int u, int r) {  // lane = batch row i, register r <-> expert c0 + u*32 + rowmap(r, half)
            const int cr = u * 32 + (r & 3) + 8 * (r >> 2);  // + 4*half folded into the bias pointers / shifted sign word
            float z = X1[u][r] + bias_mu[cr];                // experts past M carry a -1e30 bias (patched below): sp = dz = 0 there
            uint32_t sbit = 0u;
            if (BAYES) {
                sbit = (sw[u] << (31 - ((r & 3) + 8 * (r >> 2)))) & 0x80000000u;
                z += __uint_as_float(__float_as_uint(X2[u][r] + bias_p[cr]) ^ sbit);
            }
            // softplus(l) = l + ln(1 + e^-l), sigmoid(l) = 1 / (1 + e^-l) on the raw hardware transcendentals (v_exp_f32 / v_log_f32 are
            // base 2; 1 + e^-l is in [1, 1 + e^80]: no denormal fix-ups needed).  l is clamped at -80 for the exponent only
            // (leaky_relu keeps real logits far above it; the -1e30 mask of padded experts lands there and yields sp = dz = 0).
            // The f32 matrix pipe shares the FMA hardware with the VALU, so every instruction here is paid in MFMA time: keep it short.
            const bool pos = z > 0.f;
            const float l = pos ? z : z * kLeakySlope;
            const float lc = fmaxf(l, -80.f);
            const float tt = 1.f + __builtin_amdgcn_exp2f(lc * -1.4426950408889634f);
            lacc.tile = fmaf(fmaf(__builtin_amdgcn_logf(tt), 0.6931471805599453f, lc), rmask, lacc.tile);
            if (TRAIN) {
                const float dz = rscale * __builtin_amdgcn_rcpf(tt) * (pos ? 1.f : kLeakySlope);
                // dzT tile base in the buffer descriptor, lane part in voffset, register part as a scalar offset: no per-element address math
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz), dz_rsrc, dz_voff, cr * dz_row_bytes, 0);
                X1[u][r] = dz;
                X2[u][r] = __uint_as_float(__float_as_uint(dz) ^ sbit);
            }
        };
        struct BOp { float v[NJT], w[NJT]; };
        auto dh_load = [&](int u, int s, BOp& o) {   // weight rows c = u*32 + rowmap(s, half): NJT consecutive hidden units per lane
            const int rowc = u * 32 + (s & 3) + 8 * (s >> 2);  // row without the lane half
            const char* bp_;
            if (H >= 64) bp_ = sb + boff[s & 3][(s >> 2) & 1] + rowc * ROWB;
            else {
                const int row = rowc + 4 * half;
                bp_ = sb + row * ROWB + 16 * (((NJT * il) >> 2) ^ swz<H>(row)) + 4 * ((NJT * il) & 3);
            }
            if (NJT == 4) {
                const float4 b4 = *reinterpret_cast<const float4*>(bp_); o.v[0] = b4.x; o.v[1] = b4.y; o.v[NJT > 2 ? 2 : 0] = b4.z; o.v[NJT > 3 ? 3 : 0] = b4.w;
                if (BAYES) { const float4 w4 = *reinterpret_cast<const float4*>(bp_ + TB); o.w[0] = w4.x; o.w[1] = w4.y; o.w[NJT > 2 ? 2 : 0] = w4.z; o.w[NJT > 3 ? 3 : 0] = w4.w; }
            } else if (NJT == 2) {
                const float2 b2 = *reinterpret_cast<const float2*>(bp_); o.v[0] = b2.x; o.v[NJT > 1 ? 1 : 0] = b2.y;
                if (BAYES) { const float2 w2_ = *reinterpret_cast<const float2*>(bp_ + TB); o.w[0] = w2_.x; o.w[NJT > 1 ? 1 : 0] = w2_.y; }
            } else {
                o.v[0] = *reinterpret_cast<const float*>(bp_);
                if (BAYES) o.w[0] = *reinterpret_cast<const float*>(bp_ + TB);
            }
        };
        auto dh_mma = [&](int u, int s, const BOp& o) {   // dh += dz[:, c] * W[c, :]: A operand = accumulator register s as it stands
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                Y1[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(X1[u][s], o.v[jt], Y1[jt], 0, 0, 0);
                if (BAYES) Y2[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(X2[u][s], o.w[jt], Y2[jt], 0, 0, 0);
            }
        };

        constexpr int NTQ = H / 8, RPT = 16 / NTQ;  // epilogue registers handled per k-step of the other sub-tile
        {
            float4 xa[2], xw[2];
            xw[0] = xw[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            x_load(0, 0, xa[0], xw[0]);
#pragma unroll
            for (int g = 0; g < 2 * NTQ; ++g) {
                const int u = g / NTQ, tq = g % NTQ;
                if (g + 1 < 2 * NTQ) x_load((g + 1) / NTQ, (g + 1) % NTQ, xa[(g + 1) & 1], xw[(g + 1) & 1]);
                x_mma(u, tq, xa[g & 1], xw[g & 1]);
                if (u == 1) {
#pragma unroll
                    for (int rr = 0; rr < RPT; ++rr) epilogue(0, tq * RPT + rr);
                }
            }
        }
        if (TRAIN && DH) {
            BOp bo[2];
            dh_load(0, 0, bo[0]);
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                const int u = g >> 4, s2 = g & 15;
                if (g + 1 < 32) dh_load((g + 1) >> 4, (g + 1) & 15, bo[(g + 1) & 1]);
                if (u == 1 && s2 == 0) { /* epilogue(1, .) has produced every dz of sub-tile 1 by now */ }
                dh_mma(u, s2, bo[g & 1]);
                if (u == 0) epilogue(1, s2);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) epilogue(1, r);
        }
        // The next tile's DMA was issued before this tile's 32 dzT stores: wait until at most those stores are outstanding
        // (vmcnt counts loads, DMA and stores in issue order), not for the stores themselves, then a bare barrier.
        lacc.end_tile();
        if (TRAIN) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // per-row loss partial of this column group
    float lsum = lacc.sum;
    lsum += __shfl_xor(lsum, 32, 64);
    if (half == 0) p.lossp[(int64_t)i * p.NCG + cg] = p.tnw * lsum;

    if (TRAIN && DH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int irow = i0 + rowmap(r, half);
            float v[NJT];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                const int j = NJT * il + jt;  // this lane's hidden units are consecutive: one 4*NJT-byte store per register
                v[jt] = Y1[jt][r];
                if (BAYES) {
                    const uint32_t w = INJ ? p.sinbits[(int64_t)irow * NJT + (j >> 5)] : sign_word(p.si_k0, p.si_k1, (uint32_t)irow, (uint32_t)(j >> 5));
                    const float y2 = Y2[jt][r];
                    v[jt] += ((w >> (j & 31)) & 1u) ? -y2 : y2;
                }
            }
            float* dst = p.slab + ((int64_t)cg * p.Bpad + irow) * H + NJT * il;
            if (NJT == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[NJT > 2 ? 2 : 0], v[NJT > 3 ? 3 : 0]);
            else if (NJT == 2) *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[NJT > 1 ? 1 : 0]);
            else dst[0] = v[0];
        }
    }
}

template <int H, bool BAYES, bool TRAIN, bool DH, bool INJ>
__global__ __launch_bounds__(256, 1) void k_out_fwd(OutFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    out_fwd_f32_body<H, BAYES, TRAIN, DH, INJ>(p, smem);
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// forward + loss + dz + dh of the output layer in bf16x6 arithmetic (H = 128).
// Weights arrive pre-split: k_split_planes writes, per 32-expert tile, the three bf16 planes [32 rows][H] of mu (and of Wp); a tile
// (2 matrices x 3 planes x 8 KiB + biases) is one LDS stage, filled by LDS-DMA.  In LDS a plane is the dual-use image
//   off(row, ch) = 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3)))          (ch = 16-byte chunk = 8 hidden units)
// read by rows (ds_read_b128: A operand of zT = mu . hT, k = hidden unit) and by columns (ds_read_b64_tr_b16: B operand of
// dh = dz . mu, k = expert), conflict free both ways.  h's planes stay in registers (B operand of zT); the zT accumulator (lane = batch
// row, register = expert) is, converted pairwise, the A operand of the dh products, whose k order the transposed reads follow:
// element e of lane half h  <->  expert 16s + 8(e>>2) + 4h + (e&3).
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int BN6 = 32;   // experts per tile of the bf16x6 forward kernel

// planes[tile][plane][row][H] (bf16) of a row-major f32 matrix W [M, H]; rows past M are zero
__global__ void k_split_planes(const float* __restrict__ W, int M, int H, int np, float scale, uint16_t* __restrict__ out, int* __restrict__ rflag) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;    // one thread per pair of hidden units
    const int hp = H / 2;
    const int64_t row = t / hp; const int j = (int)(t % hp) * 2;
    const int64_t Mp = ((int64_t)M + BN6 - 1) / BN6 * BN6;
    if (row >= Mp) return;
    float x0 = 0.f, x1 = 0.f;
    if (row < M) { const float2 v = *reinterpret_cast<const float2*>(W + row * H + j); x0 = v.x; x1 = v.y; }
    if (np == 2 && rflag && !(fmaxf(fabsf(x0), fabsf(x1)) * scale <= 65504.f)) *rflag = 1;
    if (np == 3) planes_store_pair<3>(out, row, j, H, x0, x1, 1.f); else planes_store_pair<2>(out, row, j, H, x0, x1, scale);
}

struct OutFwd6Args {
    OutFwdArgs a;
    const uint16_t *mu_pl, *wp_pl;   // k_split_planes images of mu and Wp
    float pscale; int pacc;          // PROBS: dzT[c][i] (+)= sigmoid(leaky_relu(z)) * pscale; pacc: accumulate onto the previous MC passes
    int plogit;                      // PROBS: store the logit leaky_relu(z) itself instead (ntf_logits: the quantity the 1e-4 parity bar is stated on)
    unsigned long long* stamps;           // diagnostics (k_out_fwd_h3p<.., STAMP>, -DNTF_DIAG builds): per wave cycle sums
    float h_scale, dz_scale, u_z, u_dh;   // fp16x3 (NP = 2): scales applied to h / dz before their split, and 1/(w scale * h scale), 1/(dz scale * w scale); 1 for bf16x6
};
// fp16x3 training step: dzT holds, per element, the two fp16 planes of dz * dz_scale packed in one dword (hi | lo << 16) - the split the forward
// kernel makes anyway for its dh products - so that the dW kernel reads MFMA operands instead of splitting f32 values again
__device__ __forceinline__ void pack_planes(uint32_t p_hi, uint32_t p_lo, uint32_t& d0, uint32_t& d1) {   // planes of an element pair -> the pair's dwords
    d0 = __builtin_amdgcn_perm(p_lo, p_hi, 0x05040100u);
    d1 = __builtin_amdgcn_perm(p_lo, p_hi, 0x07060302u);
}
// the same two steps in one: f32 -> the element's packed dword (hi | lo << 16), hi = fp16(x), lo = fp16(x - hi), both rounded to nearest even like split_pair_h's
// conversions.  v_fma_mixhi_f16 reads hi as the fp16 source of an f32 fma and writes the rounded result into the upper half: 2 vector instructions an element
// against 5 (convert, convert back, subtract, convert, permute) - bit-identical dwords (tests/test_gpu_round3.py, the packed dz of the two forward kernels)
__device__ __forceinline__ uint32_t split_packed(float x) {
    uint32_t d;
    asm("v_cvt_f16_f32 %0, %1\n\tv_fma_mixhi_f16 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "=&v"(d) : "v"(x));
    return d;
}
__device__ __forceinline__ float unpack_planes(uint32_t d, float inv_scale) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = __builtin_bit_cast(h2_t, d);
    return ((float)v[0] + (float)v[1]) * inv_scale;
}

// PROBS (inference, TRAIN = false): instead of the loss, the probabilities sigmoid(leaky_relu(z)) go (accumulated over the MC passes) to the
// transposed buffer dzT [expert][batch], and lossp gets the row's entropy terms sum_c -p log(p + 1e-15) of this pass (src/mdl/fnn.py:196-208)
template <bool BAYES, bool TRAIN, bool DH, bool INJ, bool PROBS, int NP>
__global__ __launch_bounds__(256, 1) void k_out_fwd_b6(OutFwd6Args pp) {   // NP = 3: bf16x6, NP = 2: fp16x3
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const OutFwdArgs& p = pp.a;
    constexpr int H = 128, NJT = 4, NKS = H / 16;
    constexpr int PLANE = BN6 * H * 2;          // 8 KiB
    constexpr int TM = NP * PLANE;              // one matrix of a tile
    constexpr int NMAT = BAYES ? 2 : 1;
    constexpr int STAGE = NMAT * TM + 512;      // + two 64-float bias tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, il = lane & 31, half = lane >> 5;
    if (NP == 2 && range_guard_skip(p.rflag, p.rmode, false)) return;

    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int cg = bid / p.NRB, rb = bid % p.NRB;
    const int T = (p.M + BN6 - 1) / BN6;
    const int t_beg = (int)((int64_t)cg * T / p.NCG), t_end = (int)((int64_t)(cg + 1) * T / p.NCG);
    const int i0 = rb * BM + wave * 32;
    const int i = i0 + il;
    const bool row_ok = i < p.B;

    // B operand of zT: h[i][16s + 8*half + e] split into planes; sign masks of s_in for the same elements
    u32x4 hp[NKS][3];
    uint32_t sinw[NJT];      // s_in sign words of this row: the signed B operand h*s_in is hp ^ (mask built from these bits), per use
#pragma unroll
    for (int w = 0; w < NJT; ++w)
        sinw[w] = BAYES ? (INJ ? p.sinbits[(int64_t)i * NJT + w] : (row_ok ? sign_word(p.si_k0, p.si_k1, (uint32_t)i, (uint32_t)w) : 0u)) : 0u;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const float4 v0 = *reinterpret_cast<const float4*>(p.h + (int64_t)i * H + 16 * s + 8 * half);   // p.h = zero-padded copy
        const float4 v1 = *reinterpret_cast<const float4*>(p.h + (int64_t)i * H + 16 * s + 8 * half + 4);
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t pq[3];
            split_pair_np<NP>(x[2 * q], x[2 * q + 1], pp.h_scale, pq);
            hp[s][0][q] = pq[0]; hp[s][1][q] = pq[1]; hp[s][2][q] = pq[2];
        }
    }
    const float rmask = row_ok ? 1.f : 0.f;
    const float rscale = row_ok ? p.tnw * p.inv_B : 0.f;

    // lane parts of the LDS addresses
    const int fil = ((il & 3) << 2) | ((il >> 2) & 3);
    int troff[2][NJT];                              // transposed read (rr, jt): rows 8*rr + 4*half + q (+16 s' as an immediate)
    {
        const int gl = lane & 15, q = gl >> 2, pq = gl & 3, bsel = (lane >> 4) & 1;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                const int row = 8 * rr + 4 * half + q;
                const int f = ((row & 3) << 2) | ((row >> 2) & 3);
                troff[rr][jt] = 256 * row + 16 * ((4 * jt + 2 * bsel + (pq >> 1)) ^ f) + 8 * (pq & 1);
            }
    }

    f32x16 Y1[NJT], Y2[NJT];
#pragma unroll
    for (int j = 0; j < NJT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { Y1[j][r] = 0.f; Y2[j][r] = 0.f; }
    LossAcc lacc;

    const uint32_t smem_base = lds_addr(smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // Two LDS stages.  Forward-only launches (evaluation, inference) take two of these workgroups per CU (eval_ncg: twice the column groups; 66.5 KiB of LDS and
    // <= 252 registers each): one's logit epilogue beside the other's MFMAs - 0.64 -> 0.54 ms per evaluation step at config 2, Bnn top-K inference 10.1 -> 8.6 ms
    // (round 5; rounds 2-4 kept three stages in ONE workgroup per CU there)
    constexpr int NST = 2;
    auto stage_tile = [&](int t, int buf) {
        const uint32_t sb = smem_base + buf * STAGE;
        constexpr int PER_WAVE = TM / 1024 / 4;     // 1 KiB wave-instructions per wave per matrix
#pragma unroll
        for (int n = 0; n < PER_WAVE; ++n) {
            const int inst = wave_u * PER_WAVE + n;
            const int pos = inst * 1024 + lane * 16;            // destination inside the matrix image: plane, row, physical chunk
            const int row = (pos >> 8) & 31, chp = (pos >> 4) & 15;
            const int ch = chp ^ (((row & 3) << 2) | ((row >> 2) & 3));
            const size_t src = (size_t)t * TM + (pos & ~255) + 16 * ch;
            glds16(reinterpret_cast<const char*>(pp.mu_pl) + src, sb + inst * 1024);
            if (BAYES) glds16(reinterpret_cast<const char*>(pp.wp_pl) + src, sb + TM + inst * 1024);
        }
        const int c0 = t * BN6;
        if (wave_u == 0) glds4(p.mu_b + min(c0 + lane, p.M - 1), sb + NMAT * TM);
        if (BAYES && wave_u == 1) glds4(p.bp + min(c0 + lane, p.M - 1), sb + NMAT * TM + 256);
    };
    auto sign_word_t = [&](int t) -> uint32_t {     // s_out signs of (row i, experts 32t .. 32t+31)
        if (!BAYES || !row_ok) return 0u;
        if (INJ) return p.sbits[(int64_t)i * p.nCB + t];
        return sign_word(p.so_k0, p.so_k1, (uint32_t)i, (uint32_t)t);
    };
    if (t_beg < t_end) stage_tile(t_beg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) % NST;
        const uint32_t sw = sign_word_t(t) >> (4 * half);
        if (t + 1 < t_end) stage_tile(t + 1, buf ^ 1);
        char* sb = smem + buf * STAGE;
        const int c0 = t * BN6;
        if (c0 + BN6 > p.M) {  // ragged last tile (workgroup-uniform): mask the experts past M through their bias
            if (tid < BN6 && c0 + tid >= p.M) reinterpret_cast<float*>(sb + NMAT * TM)[tid] = -1e30f;
            __syncthreads();
        }
        f32x16 X1, X2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { X1[r] = 0.f; X2[r] = 0.f; }
        const uint32_t sbase = lds_addr(sb);
        constexpr int dz_row_bytes = 128;   // dzT tile layout, see dzt_index
        const __amdgpu_buffer_rsrc_t dz_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.dzT + dzt_tile_base(c0, p.Bpad), 0, ((p.Bpad >> 5) * 8192 - ((c0 & 255) << 5)) * 4, 0x00020000);
        const int dz_voff = ((i >> 5) * 8192 + 4 * half * 32 + (i & 31)) * 4;
        float pold[16];      // PROBS, later MC passes: the running sums of this tile, fetched under the zT products
        if (PROBS) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                pold[r] = pp.pacc ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dz_rsrc, dz_voff, ((r & 3) + 8 * (r >> 2)) * dz_row_bytes, 0)) : 0.f;
        }

        // ---- zT = mu . hT (+ Wp . (h*s_in)T): 8 k-steps of 16 hidden units; half-groups (k-step, matrix) of 3 fragment reads + 6 MFMAs,
        // the reads of the next half-group in flight under the MFMAs of the current one
        {
            constexpr int NHG = NKS * NMAT;
            auto z_load = [&](int hg, u32x4 (&fr)[3]) {
                const int s = hg / NMAT, mat = hg % NMAT;
                const char* ap = sb + 256 * il + 16 * ((2 * s + half) ^ fil) + mat * TM;
#pragma unroll
                for (int q = 0; q < NP; ++q) fr[q] = *reinterpret_cast<const u32x4*>(ap + q * PLANE);
            };
            u32x4 fr[2][3];
            z_load(0, fr[0]);
#pragma unroll
            for (int hg = 0; hg < NHG; ++hg) {
                if (hg + 1 < NHG) z_load(hg + 1, fr[(hg + 1) & 1]);
                asm volatile("" ::: "memory");
                const int s = hg / NMAT, mat = hg % NMAT;
                if (mat == 0) X1 = mfma_np<NP>(fr[hg & 1], hp[s], X1);
                else {
                    u32x4 hs[3];
                    const uint32_t w8 = sinw[s >> 1] >> (16 * (s & 1) + 8 * half);
                    u32x4 hm;
#pragma unroll
                    for (int q = 0; q < 4; ++q) hm[q] = ((w8 << (15 - 2 * q)) & 0x8000u) | ((w8 << (30 - 2 * q)) & 0x80000000u);
#pragma unroll
                    for (int q = 0; q < NP; ++q) hs[q] = hp[s][q] ^ hm;
                    X2 = mfma_np<NP>(fr[hg & 1], hs, X2);
                }
            }
        }

        // ---- epilogue: lane = batch row i, register r <-> expert c0 + rowmap(r, half)
        const float* bias_mu = reinterpret_cast<const float*>(sb + NMAT * TM) + 4 * half;
        const float* bias_p = reinterpret_cast<const float*>(sb + NMAT * TM + 256) + 4 * half;
        auto epilogue = [&](int r) {
            const int cr = (r & 3) + 8 * (r >> 2);
            float z = fmaf(X1[r], pp.u_z, bias_mu[cr]);
            if (BAYES) z += __uint_as_float(__float_as_uint(fmaf(X2[r], pp.u_z, bias_p[cr])) ^ ((sw << (31 - cr)) & 0x80000000u));
            const bool pos = z > 0.f;
            const float l = pos ? z : z * kLeakySlope;
            const float lc = fmaxf(l, -80.f);
            const float tt = 1.f + __builtin_amdgcn_exp2f(lc * -1.4426950408889634f);
            if (PROBS) {
                const float pr = __builtin_amdgcn_rcpf(tt) * rmask;       // experts past M: bias -1e30 -> tt = 1 + e^80 -> 0
                lacc.tile = fmaf(-pr * 0.6931471805599453f, __builtin_amdgcn_logf(pr + 1e-15f), lacc.tile);
                const float o = pp.plogit ? l : fmaf(pr, pp.pscale, pold[r]);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), dz_rsrc, dz_voff, cr * dz_row_bytes, 0);
                return;
            }
            lacc.tile = fmaf(fmaf(__builtin_amdgcn_logf(tt), 0.6931471805599453f, lc), rmask, lacc.tile);
            if (TRAIN) {
                const float dz = rscale * __builtin_amdgcn_rcpf(tt) * (pos ? 1.f : kLeakySlope);
                if (NP != 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz), dz_rsrc, dz_voff, cr * dz_row_bytes, 0);
                X1[r] = dz;
            }
        };
        auto store_packed = [&](int r0, uint32_t p_hi, uint32_t p_lo) {   // NP == 2: registers r0, r0 + 1 as packed plane pairs
            uint32_t d0, d1;
            pack_planes(p_hi, p_lo, d0, d1);
            __builtin_amdgcn_raw_buffer_store_b32(d0, dz_rsrc, dz_voff, ((r0 & 3) + 8 * (r0 >> 2)) * dz_row_bytes, 0);
            __builtin_amdgcn_raw_buffer_store_b32(d1, dz_rsrc, dz_voff, (((r0 + 1) & 3) + 8 * ((r0 + 1) >> 2)) * dz_row_bytes, 0);
        };
        if (!(TRAIN && DH)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) epilogue(r);
            if (TRAIN && NP == 2) {
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 2) { uint32_t pq[3]; split_pair_np<2>(X1[r0], X1[r0 + 1], pp.dz_scale, pq); store_packed(r0, pq[0], pq[1]); }
            }
        } else {
            // ---- dh += dz . mu_tile (+ (dz*s_out) . Wp_tile): the accumulator registers, split, are the A operand.  Only the first half of
            // the epilogue stands alone; the second half is spread over the MFMAs of the first k-step
            auto split_a = [&](int s2, u32x4 (&ad)[3], u32x4 (&as)[3]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t pq[3];
                    const int r0 = 8 * s2 + 2 * q;
                    split_pair_np<NP>(X1[r0], X1[r0 + 1], pp.dz_scale, pq);
                    ad[0][q] = pq[0]; ad[1][q] = pq[1]; ad[2][q] = pq[2];
                    if (NP == 2) store_packed(r0, pq[0], pq[1]);
                    if (BAYES) {
                        const int c0r = (r0 & 3) + 8 * (r0 >> 2);   // registers r0, r0+1 are experts c0r, c0r+1 (+4*half, folded into sw)
                        const uint32_t m = (((sw << (31 - c0r)) & 0x80000000u) >> 16) | ((sw << (30 - c0r)) & 0x80000000u);
                        as[0][q] = pq[0] ^ m; as[1][q] = pq[1] ^ m; as[2][q] = pq[2] ^ m;
                    }
                }
            };
            auto tr_load = [&](int g, u32x4 (&bf)[3]) {   // g = (s2, jt, mat): B fragments (k = expert, n = hidden unit 32 jt + il)
                const int mat = g % NMAT, jt = (g / NMAT) % NJT, s2 = g / (NMAT * NJT);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const uint32_t o = 4096 * s2 + q * PLANE + mat * TM;
                    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)(sbase + troff[0][jt] + o)));
                    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)(sbase + troff[1][jt] + o)));
                    bf[q][0] = lo.x; bf[q][1] = lo.y; bf[q][2] = hi.x; bf[q][3] = hi.y;
                }
            };
            constexpr int NG = 2 * NJT * NMAT;
            u32x4 bf[2][3];
            tr_load(0, bf[0]);
#pragma unroll
            for (int r = 0; r < 8; ++r) epilogue(r);
            u32x4 ad[2][3], as[2][3];
            split_a(0, ad[0], as[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) tr_load(g + 1, bf[(g + 1) & 1]);
                asm volatile("" ::: "memory");
                const int mat = g % NMAT, jt = (g / NMAT) % NJT, s2 = g / (NMAT * NJT);
                if (mat == 0) Y1[jt] = mfma_np<NP>(ad[s2], bf[g & 1], Y1[jt]);
                else Y2[jt] = mfma_np<NP>(as[s2], bf[g & 1], Y2[jt]);
                if (s2 == 0) {   // second half of the epilogue in the shadow of the first k-step's MFMAs
                    constexpr int PER = 8 / (NJT * NMAT) > 0 ? 8 / (NJT * NMAT) : 1;
                    if (g * PER < 8) {
#pragma unroll
                        for (int rr = 0; rr < PER; ++rr) epilogue(8 + g * PER + rr);
                    }
                    if (g == NJT * NMAT - 1) split_a(1, ad[1], as[1]);
                }
            }
        }
        lacc.end_tile();
        if (TRAIN) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the next tile's DMA is older than this tile's 16 dzT stores
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    float lsum = lacc.sum;
    lsum += __shfl_xor(lsum, 32, 64);
    if (half == 0) p.lossp[(int64_t)i * p.NCG + cg] = PROBS ? lsum : p.tnw * lsum;

    if (TRAIN && DH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int irow = i0 + rowmap(r, half);
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                float v = Y1[jt][r] * pp.u_dh;
                if (BAYES) {
                    const uint32_t w = INJ ? p.sinbits[(int64_t)irow * NJT + jt] : sign_word(p.si_k0, p.si_k1, (uint32_t)irow, (uint32_t)jt);
                    const float y2 = Y2[jt][r] * pp.u_dh;
                    v += ((w >> il) & 1u) ? -y2 : y2;
                }
                p.slab[((int64_t)cg * p.Bpad + irow) * H + 32 * jt + il] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_out_fwd_h3p (round 4): the fp16x3 training forward as PRODUCER / CONSUMER WAVE PAIRS, two waves per SIMD.
// Round 3's one-wave form (k_out_fwd_h3x: 64-expert tiles, phases rotated across tiles; retired in round 6 - bit-identical dz / dh while both existed) issued everything in order - 192 MFMAs (8 issue cycles each), ~860 vector instructions, ~210 LDS reads, 17 DMA
// pieces, 32 stores a tile: ~10.5 k cycles of issue against 6.1 k of matrix pipe, and the issue is what its 10.9 k cycles a tile are (without its MFMAs the
// kernel takes 0.59 of its 0.71 ms, without its epilogue 0.57; re-ordering or trimming the vector work moves nothing).  Sixteen-row waves (round 3's k_out_fwd_h3y, retired in round 5) double
// the LDS traffic and the MFMA issue instead.  Here the 32 rows of a wave pair stay together and the WORK is split:
//   wave A ("logit", waves 0-3):    zT(s) = planes(s) . hT on its h / h*s_in planes (128 registers), bias, s_out sign, leaky_relu, softplus into the loss (one
//                                   v_log_f32 per four logits: the log of a product); tt = 1 + e^-l of its 32 rows x 32 experts, the leaky_relu branch in its sign,
//                                   handed to wave B through LDS (4 KiB a sub-tile); the dzT stores of sub-tile s-2 (16 bytes a lane, from the same LDS slot);
//                                   the LDS-DMA of the first matrix and the biases of sub-tile s+1
//   wave B ("gradient", waves 4-7): dz = row constant / tt, its fp16 split (packed back into the slot as [expert][row] for wave A's stores), and
//                                   dh += dz(s-1) . planes(s-1) on its 128 accumulators; the LDS-DMA of the second matrix of sub-tile s+1
// of the same 32 batch rows (lane = row in both: the hand-over is lane to lane), one step = one 32-expert sub-tile, one barrier a step.  A issues its MFMAs
// first and its vector work after them, B its vector work first and its MFMAs after it: matrix work of one beside vector work of the other (the other pairings -
// half of B's vector work riding on its first MFMAs, the roles on the other wave age, priorities per segment - cost 3 to 15 %: vector issue is arbitrated by
// age, and two vector streams side by side starve the younger one's MFMAs).  Measured per step and wave (`-DNTF_DIAG`, NTF_FWD_ABL=9, profiles/r4_fwd_pair_stamps.txt):
// A 535 (top, first fragments) + 1 820 (48 MFMAs + DMA) + 345 (4 stores) + 1 625 (logits); B 830 (top, hand-over and first fragment reads) + 1 280 (dz) + 2 200
// (48 MFMAs beside A's vector work) = 4.7 k cycles a sub-tile against the one-wave kernel's 5.5 k - 1.07 M cycles a wave at 1.65 GHz against 1.27 M at 1.76 (the package
// power limit gives a third of the saving back): 0.72 -> 0.65-0.67 ms on the same box.
// LDS: a ring of three 32-expert stages (2 matrices x 2 planes x [32 rows][256 B] + biases = 32.5 KiB each: zT reads stage s, dh stage s-1, the DMA fills s+1)
// + two hand-over slots of 16 KiB = 129.5 KiB.  Sub-tile order, MFMA order per accumulator, epilogue arithmetic and the packed dz are the one-wave kernel's (dzT and
// the dh slabs were bit-identical to it, the loss differs in the order of its sums).  An operand outside the fp16 window: the kernel returns (exact-f32 launch behind it).
// ------------------------------------------------------------------------------------------------
// static wave priorities for the whole kernel (round 6 experiment, MI355X_MICROARCH.md "static priority for the younger half"; profiles/r6_fwd_prio_ab.md): 0 = none
#ifndef H3P_APRIO
#define H3P_APRIO 0
#endif
#ifndef H3P_BPRIO
#define H3P_BPRIO 0
#endif
template <bool BAYES, bool INJ, bool STAMP = false>      // STAMP (-DNTF_DIAG builds, NTF_FWD_ABL=9): cycle sums per wave and step segment into pp.stamps
__global__ __launch_bounds__(512) void k_out_fwd_h3p(OutFwd6Args pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const OutFwdArgs& p = pp.a;
    constexpr int H = 128, NJT = 4, NKS = H / 16, SUB = 32;
    constexpr int PLANE = SUB * H * 2;          // 8 KiB
    constexpr int TM = 2 * PLANE;               // one matrix of a sub-tile: hi and lo plane
    constexpr int NMAT = BAYES ? 2 : 1;
    constexpr int SLOT = NMAT * TM + 512;       // + two 32-float bias tiles (each fetched by all 64 lanes: 256 B apart)
    constexpr int HB0 = 3 * SLOT, HBSLOT = 4 * 4096;
    const int tid = threadIdx.x, lane = tid & 63, il = lane & 31, half = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), pair = wave_u & 3, role = wave_u >> 2;
    if (p.rmode == 1 && __builtin_nontemporal_load(p.rflag) != 0) return;
    const long long k_c0 = STAMP ? clock64() : 0, k_w0 = STAMP ? wall_clock64() : 0;
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int slot) {
        if (!STAMP) return;
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long tnow;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (slot >= 0) st_sum[slot] += tnow - st_prev;
        st_prev = tnow;
    };
    auto stamp_out = [&](int nstep) {
        if (STAMP && pp.stamps && lane == 0) {
            unsigned long long* o = pp.stamps + ((int64_t)blockIdx.x * 8 + wave_u) * 10;
#pragma unroll
            for (int q = 0; q < 6; ++q) o[q] = st_sum[q];
            o[6] = (unsigned long long)nstep; o[7] = (unsigned long long)(clock64() - k_c0); o[8] = (unsigned long long)(wall_clock64() - k_w0); o[9] = 0;
        }
    };

    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int cg = bid / p.NRB, rb = bid % p.NRB;
    const int tspan = p.t_hi - p.t_lo;      // (a launch over a range of the experts: OutFwdArgs.t_lo)
    const int s_beg = 2 * (p.t_lo + (int)((int64_t)cg * tspan / p.NCG)), s_end = 2 * (p.t_lo + (int)((int64_t)(cg + 1) * tspan / p.NCG));   // 64-expert tiles -> 32-expert sub-tiles
    const int i0 = rb * BM + pair * 32;
    const int i = i0 + il;
    const bool row_ok = i < p.B;
    const uint32_t smem_base = lds_addr(smem);
    typedef const __attribute__((address_space(3))) char* ldsp_t;
    const int fil = ((il & 3) << 2) | ((il >> 2) & 3);
    auto sign_w = [&](int s) -> uint32_t {          // s_out signs of (row i, experts 32 s .. 32 s + 31), shifted to this half's registers
        if (!BAYES || !row_ok) return 0u;
        const uint32_t w = INJ ? p.sbits[(int64_t)i * p.nCB + s] : sign_word(p.so_k0, p.so_k1, (uint32_t)i, (uint32_t)s);
        return w >> (4 * half);
    };
    // a ragged or empty last sub-tile (workgroup-uniform): the experts past M are masked through their bias (l = -20: softplus = 2e-9, dz = 0 to rounding)
    auto mask_past_m = [&](int s) {
        if (32 * s + SUB > p.M) {
            if (tid < SUB && 32 * s + tid >= p.M) reinterpret_cast<float*>(smem + (s % 3) * SLOT + NMAT * TM)[tid] = -2000.f;
            __syncthreads();
        }
    };
    if (role == 0) {
        // ================================================================ wave A: zT, logits, DMA
        if (H3P_APRIO) __builtin_amdgcn_s_setprio(H3P_APRIO);
        u32x4 hp[NKS][2], hs[NKS][2];               // B operand of zT: fp16 planes of h[i][16 s + 8 half ..] and of h * s_in
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(p.h + (int64_t)i * H + 16 * s + 8 * half);   // p.h = zero-padded copy
            const float4 v1 = *reinterpret_cast<const float4*>(p.h + (int64_t)i * H + 16 * s + 8 * half + 4);
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const uint32_t sw_in = BAYES ? (INJ ? p.sinbits[(int64_t)i * NJT + (s >> 1)] : (row_ok ? sign_word(p.si_k0, p.si_k1, (uint32_t)i, (uint32_t)(s >> 1)) : 0u)) : 0u;
            const uint32_t w8 = sw_in >> (16 * (s & 1) + 8 * half);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t pq[3];
                split_pair_np<2>(x[2 * q], x[2 * q + 1], pp.h_scale, pq);
                const uint32_t hm = ((w8 << (15 - 2 * q)) & 0x8000u) | ((w8 << (30 - 2 * q)) & 0x80000000u);
                hp[s][0][q] = pq[0]; hp[s][1][q] = pq[1];
                hs[s][0][q] = pq[0] ^ hm; hs[s][1][q] = pq[1] ^ hm;
            }
        }
        // DMA: the 16 one-KiB pieces of a matrix image over the four A waves (4 each and matrix) + the bias piece
        constexpr int PER_WAVE = TM / 1024 / 4;
        constexpr int NPIECE = PER_WAVE + 1;        // wave B issues the pieces of the second matrix
        uint32_t dsrc[PER_WAVE];
#pragma unroll
        for (int n = 0; n < PER_WAVE; ++n) {
            const int pos = (pair * PER_WAVE + n) * 1024 + lane * 16;       // destination inside the matrix image: plane, row (0..31), physical chunk
            const int plane = pos / PLANE, row = (pos >> 8) & 31, chp = (pos >> 4) & 15;
            const int ch = chp ^ (((row & 3) << 2) | ((row >> 2) & 3));
            dsrc[n] = (uint32_t)(((plane * 32 + row) * 256) + 16 * ch);   // the planes are stored per 32-expert tile: [tile32][plane][32 rows][256 B]
        }
        auto stage_piece = [&](int s, int slot, int n) {    // piece n of sub-tile s into ring slot `slot`
            const uint32_t sb = smem_base + slot * SLOT;
            if (n < NPIECE - 1) {
                const int mat = n / PER_WAVE, nn = n % PER_WAVE;
                const char* base = reinterpret_cast<const char*>(mat ? pp.wp_pl : pp.mu_pl) + (size_t)s * TM;    // wave-uniform
                glds16s(base, dsrc[nn], sb + mat * TM + (pair * PER_WAVE + nn) * 1024);
            } else {   // the two bias tiles: even waves fetch mu_b's, odd waves bp's (lanes 32-63 repeat lanes 0-31 into the 128 B behind them)
                const int which = BAYES ? (pair & 1) : 0;
                glds4((which ? p.bp : p.mu_b) + min(32 * s + il, p.M - 1), sb + NMAT * TM + which * 256);
            }
        };
        if (s_beg < s_end) {
#pragma unroll
            for (int n = 0; n < NPIECE; ++n) stage_piece(s_beg, s_beg % 3, n);
#pragma unroll
            for (int mat = 1; mat < NMAT; ++mat)
#pragma unroll
                for (int nn = 0; nn < PER_WAVE; ++nn) glds16s(reinterpret_cast<const char*>(mat ? pp.wp_pl : pp.mu_pl) + (size_t)s_beg * TM, dsrc[nn], smem_base + (s_beg % 3) * SLOT + mat * TM + (pair * PER_WAVE + nn) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        int zrow[NKS];
#pragma unroll
        for (int s = 0; s < NKS; ++s) zrow[s] = 256 * il + 16 * ((2 * s + half) ^ fil);
        LossAcc lacc;
        const float rmask = row_ok ? 1.f : 0.f;
        constexpr int NHG = NKS * NMAT, BG = 2, NB = NHG / BG;
        uint32_t swn = s_beg < s_end ? sign_w(s_beg) : 0u;
        stamp(-1);
        const int dz_voff = ((i0 >> 5) * 8192 + (lane >> 3) * 32 + 4 * (lane & 7)) * 4;      // rows i0 + 4 (lane & 7) .. + 3 of expert lane >> 3 (+ 8 per store)
        // the packed dz of sub-tile sd, left as [32 experts][32 rows] by wave B in the slot of that sub-tile: read at the top of the step, stored (1 KiB a store)
        // behind its MFMAs - a store waiting for its LDS read cost 120 cycles
        u32x4 dzv[4];
        auto load_dz = [&](int sd) {
            const char* hbr = smem + HB0 + (sd & 1) * HBSLOT + pair * 4096 + lane * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) dzv[q] = *reinterpret_cast<const u32x4*>(hbr + q * 1024);
        };
        auto store_dz = [&](int sd) {
            const int u = sd & 1, c0 = (sd >> 1) * 64;
            const __amdgpu_buffer_rsrc_t dz_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.dzT + dzt_tile_base(c0, p.Bpad), 0, ((p.Bpad >> 5) * 8192 - ((c0 & 255) << 5)) * 4, 0x00020000);
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128(dzv[q], dz_rsrc, dz_voff, (32 * u + 8 * q) * 128, 0);
        };
        for (int s = s_beg; s <= s_end + 1; ++s) {
            if (s - 2 >= s_beg) load_dz(s - 2);      // (round 5: issued behind the first weight fragments instead, these reads cost the kernel +0.006 ms - profiles/r5_gap_experiments.md)
            if (s - 2 >= s_beg && s >= s_end) store_dz(s - 2);      // (behind the last sub-tiles; otherwise after this step's MFMAs, below)
            if (s < s_end) {
                mask_past_m(s);
                const uint32_t swu = swn;
                swn = sign_w(min(s + 1, s_end - 1));
                const uint32_t sbase = smem_base + (s % 3) * SLOT;
                const char* sb = smem + (s % 3) * SLOT;
                ldsp_t zb[NKS];
#pragma unroll
                for (int k = 0; k < NKS; ++k) zb[k] = (ldsp_t)(size_t)(sbase + zrow[k]);
                f32x16 X1, X2;
                u32x4 fb[2][BG][2];
                auto z_load = [&](int hg, u32x4 (&fr)[2]) {
                    const int k = hg / NMAT, mat = hg % NMAT;
#pragma unroll
                    for (int q = 0; q < 2; ++q) fr[q] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(zb[k] + (mat * TM + q * PLANE));
                };
                auto z_mma = [&](int hg, const u32x4 (&fr)[2]) {
                    const int k = hg / NMAT, mat = hg % NMAT;
                    f32x16 zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                    f32x16 acc = mat == 0 ? (k == 0 ? zero : X1) : (k == 0 ? zero : X2);
                    const u32x4 (&b)[2] = mat == 0 ? hp[k] : hs[k];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(fr[1]), as_frag_h(b[0]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(fr[0]), as_frag_h(b[1]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(fr[0]), as_frag_h(b[0]), acc, 0, 0, 0);
                    if (mat == 0) X1 = acc; else X2 = acc;
                };
#pragma unroll
                for (int k = 0; k < BG; ++k) z_load(k, fb[0][k]);
                stamp(4);
                const int sn = min(s + 1, s_end - 1);       // behind the last sub-tile the free stage takes that sub-tile once more
#pragma unroll
                for (int lb = 0; lb < NB; ++lb) {
                    if (lb + 1 < NB) {
#pragma unroll
                        for (int k = 0; k < BG; ++k) z_load((lb + 1) * BG + k, fb[(lb + 1) & 1][k]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < BG; ++k) {
                        const int hg = lb * BG + k;
                        z_mma(hg, fb[lb & 1][k]);
                        if (hg < NPIECE) stage_piece(sn, (s + 1) % 3, hg);
                    }
                }
                stamp(0);
                if (s - 2 >= s_beg) store_dz(s - 2);    // before this step's 1 + e^-l go into the same slot
                stamp(5);
                // logits of the 16 registers: lane = batch row i, register r <-> expert 32 s + rowmap(r, half)
                float bm[16], bq[16];
                {
                    const float* bias_mu = reinterpret_cast<const float*>(sb + NMAT * TM) + 4 * half;
                    const float* bias_p = reinterpret_cast<const float*>(sb + NMAT * TM + 256) + 4 * half;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 m = *reinterpret_cast<const float4*>(bias_mu + 8 * j);
                        bm[4 * j] = m.x; bm[4 * j + 1] = m.y; bm[4 * j + 2] = m.z; bm[4 * j + 3] = m.w;
                        if (BAYES) { const float4 q = *reinterpret_cast<const float4*>(bias_p + 8 * j); bq[4 * j] = q.x; bq[4 * j + 1] = q.y; bq[4 * j + 2] = q.z; bq[4 * j + 3] = q.w; }
                    }
                }
                // what wave B gets is tt = 1 + e^-l with the branch of leaky_relu in its sign (-tt: z > 0); softplus(l) = log(tt) + l goes into the row's loss here
                char* hb = smem + HB0 + (s & 1) * HBSLOT + pair * 4096 + lane * 16;
                float lt = 0.f;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    float l[4], tt[4], v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * j4 + j, cr = (r & 3) + 8 * (r >> 2);
                        float z = fmaf(X1[r], pp.u_z, bm[r]);
                        if (BAYES) z += __uint_as_float(__float_as_uint(fmaf(X2[r], pp.u_z, bq[r])) ^ ((swu << (31 - cr)) & 0x80000000u));
                        const bool pos = z > 0.f;
                        // l >= -21: the product of four 1 + e^-l below must stay finite (four logits of -22.2 reach 3.4e38 - a row with an outlier activation did, at
                        // step 1 134 of the benchmark's run: softplus inf, then NaN through the compensated sum; the gradients never went through it).  Below -21
                        // softplus and its slope are under 7.6e-10: the clamp moves the loss and dz by less than that per expert
                        l[j] = fmaxf(pos ? z : z * kLeakySlope, -21.f);
                        tt[j] = 1.f + __builtin_amdgcn_exp2f(l[j] * -1.4426950408889634f);
                        v[j] = pos ? -tt[j] : tt[j];
                    }
                    // sum of four softplus(l) = log(tt0 tt1 tt2 tt3) + (l0 + l1 + l2 + l3): one v_log_f32 for four.  The product would overflow past sum(-l) = 88, i.e. a mean
                    // logit below -22 (a pre-activation below -2 200 under leaky_relu): l is clamped at -21 above; the experts past M are masked at l = -20 (tt = 4.9e8)
                    lt += fmaf(__builtin_amdgcn_logf((tt[0] * tt[1]) * (tt[2] * tt[3])), 0.6931471805599453f, (l[0] + l[1]) + (l[2] + l[3]));
                    *reinterpret_cast<float4*>(hb + j4 * 1024) = make_float4(v[0], v[1], v[2], v[3]);
                }
                lacc.tile = lt * rmask; lacc.end_tile();
            }
            stamp(1);
            // the DMA of sub-tile s+1 has landed (it is older than the four dz stores of this step, where there are any), the logits of s are in the slot
            if (s - 2 >= s_beg && s < s_end) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            stamp(2);
            __builtin_amdgcn_s_barrier();
            stamp(3);
        }
        stamp_out(s_end - s_beg);
        float lsum = lacc.sum;
        lsum += __shfl_xor(lsum, 32, 64);
        if (half == 0) p.lossp[(int64_t)i * p.ncg_tot + p.cg_off + cg] = p.tnw * lsum;
        return;
    }

    // ==================================================================== wave B: loss, dz, dh
    if (H3P_BPRIO) __builtin_amdgcn_s_setprio(H3P_BPRIO);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // (the barrier behind wave A's first DMA)
    const float rscale_pos = row_ok ? p.tnw * p.inv_B * pp.dz_scale : 0.f;    // dz * dz_scale = this * sigmoid(l)   (z > 0), ...
    const float rscale_neg = rscale_pos * kLeakySlope;                         // ... * leaky slope                       (z <= 0)
    int troff[2][NJT];                              // transposed read (rr, jt): rows 8*rr + 4*half + q (+ 16 per k-step as an immediate)
    {
        const int gl = lane & 15, q = gl >> 2, pq = gl & 3, bsel = (lane >> 4) & 1;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                const int row = 8 * rr + 4 * half + q;
                const int f = ((row & 3) << 2) | ((row >> 2) & 3);
                troff[rr][jt] = 256 * row + 16 * ((4 * jt + 2 * bsel + (pq >> 1)) ^ f) + 8 * (pq & 1);
            }
    }
    f32x16 Y1[NJT], Y2[NJT];
#pragma unroll
    for (int j = 0; j < NJT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { Y1[j][r] = 0.f; Y2[j][r] = 0.f; }
    constexpr int NGD = 2 * NJT * NMAT, BGD = 2, NBD = NGD / BGD;
    uint32_t swn = s_beg < s_end ? sign_w(s_beg) : 0u;
    uint32_t dsrcb[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int pos = (pair * 4 + n) * 1024 + lane * 16;
        const int plane = pos / PLANE, row = (pos >> 8) & 31, chp = (pos >> 4) & 15;
        dsrcb[n] = (uint32_t)(((plane * 32 + row) * 256) + 16 * (chp ^ (((row & 3) << 2) | ((row >> 2) & 3))));
    }
    stamp(-1);
    for (int s = s_beg; s <= s_end + 1; ++s) {
        if (s < s_end) mask_past_m(s);
        if (s < s_end) {       // this wave's share of sub-tile s+1 (behind the last one: that one once more) into the free stage
            const int sn = min(s + 1, s_end - 1);
#pragma unroll
            for (int mat = 1; mat < NMAT; ++mat)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) glds16s(reinterpret_cast<const char*>(mat ? pp.wp_pl : pp.mu_pl) + (size_t)sn * TM, dsrcb[nn], smem_base + ((s + 1) % 3) * SLOT + mat * TM + (pair * 4 + nn) * 1024);
        }
        if (s > s_beg && s <= s_end) {
            const int sd = s - 1;
            const uint32_t swu = swn;
            swn = sign_w(min(sd + 1, s_end - 1));
            const uint32_t sbase = smem_base + (sd % 3) * SLOT;
            ldsp_t tb[2][NJT];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) tb[rr][jt] = (ldsp_t)(size_t)(sbase + troff[rr][jt]);
            u32x4 fb[2][BGD][2];
            auto tr_load = [&](int g, u32x4 (&bf)[2]) {   // g = (s2, jt, mat)
                const int mat = g % NMAT, jt = (g / NMAT) % NJT, s2 = g / (NMAT * NJT);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int o = 4096 * s2 + q * PLANE + mat * TM;
                    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tb[0][jt] + o)));
                    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tb[1][jt] + o)));
                    bf[q][0] = lo.x; bf[q][1] = lo.y; bf[q][2] = hi.x; bf[q][3] = hi.y;
                }
            };
            // 1 + e^-l of sub-tile sd (negative: z > 0), as wave A's same lane left them
            float l[16];
            {
                const char* hb = smem + HB0 + (sd & 1) * HBSLOT + pair * 4096 + lane * 16;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const float4 v = *reinterpret_cast<const float4*>(hb + j4 * 1024);
                    l[4 * j4] = v.x; l[4 * j4 + 1] = v.y; l[4 * j4 + 2] = v.z; l[4 * j4 + 3] = v.w;
                }
            }
#pragma unroll
            for (int k = 0; k < BGD; ++k) tr_load(k, fb[0][k]);
            stamp(4);
            char* hbw = smem + HB0 + (sd & 1) * HBSLOT + pair * 4096 + (4 * half * 32 + il) * 4;      // element (expert 4 half + .., row il) of [32 experts][32 rows]
            u32x4 ad[2][2];         // [k-step of 16 experts][plane]: fp16 planes of dz, the A operand of the dh products
            auto dz_pair = [&](int r0) {
                float rc[2], dz[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) rc[j] = __builtin_amdgcn_rcpf(__builtin_fabsf(l[r0 + j]));
#pragma unroll
                for (int j = 0; j < 2; ++j) dz[j] = rc[j] * (l[r0 + j] < 0.f ? rscale_pos : rscale_neg);
                const uint32_t d0 = split_packed(dz[0]), d1 = split_packed(dz[1]);    // (no clamp: |dz| * dz_scale < 2^14)
                ad[r0 >> 3][0][(r0 & 7) >> 1] = __builtin_amdgcn_perm(d1, d0, 0x05040100u);
                ad[r0 >> 3][1][(r0 & 7) >> 1] = __builtin_amdgcn_perm(d1, d0, 0x07060302u);
                // the packed dz goes back into the slot its 1 + e^-l came from, as [expert][row]: wave A stores it from there, 16 bytes a lane (a dword store
                // per register from here cost this wave ~35 cycles each, 1 100 of its 4 800 a step)
                *reinterpret_cast<uint32_t*>(hbw + ((r0 & 3) + 8 * (r0 >> 2)) * 128) = d0;
                *reinterpret_cast<uint32_t*>(hbw + (((r0 + 1) & 3) + 8 * ((r0 + 1) >> 2)) * 128) = d1;
            };
            // (all of it before the MFMAs: riding half of it on the first k-step's MFMAs puts this wave's vector work beside wave A's and costs 1 350 cycles of the MFMA segment)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 2) dz_pair(r0);
            stamp(0);
            auto d_mma = [&](int g, const u32x4 (&bf)[2]) {
                const int mat = g % NMAT, jt = (g / NMAT) % NJT, s2 = g / (NMAT * NJT);
                u32x4 a0 = ad[s2][0], a1 = ad[s2][1];
                if (mat) {          // planes of dz * s_out
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r0 = 8 * s2 + 2 * q, c0r = (r0 & 3) + 8 * (r0 >> 2);
                        const uint32_t m = (((swu << (31 - c0r)) & 0x80000000u) >> 16) | ((swu << (30 - c0r)) & 0x80000000u);
                        a0[q] ^= m; a1[q] ^= m;
                    }
                }
                f32x16 acc = mat == 0 ? Y1[jt] : Y2[jt];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(a1), as_frag_h(bf[0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(a0), as_frag_h(bf[1]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(a0), as_frag_h(bf[0]), acc, 0, 0, 0);
                if (mat == 0) Y1[jt] = acc; else Y2[jt] = acc;
            };
#pragma unroll
            for (int lb = 0; lb < NBD; ++lb) {
                if (lb + 1 < NBD) {
#pragma unroll
                    for (int k = 0; k < BGD; ++k) tr_load((lb + 1) * BGD + k, fb[(lb + 1) & 1][k]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < BGD; ++k) d_mma(lb * BGD + k, fb[lb & 1][k]);
            }
        }
        stamp(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of the stage and of the hand-over slot have returned
        stamp(2);
        __builtin_amdgcn_s_barrier();
        stamp(3);
    }
    stamp_out(s_end - s_beg);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int irow = i0 + rowmap(r, half);
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            float v = Y1[jt][r] * pp.u_dh;
            if (BAYES) {
                const uint32_t w = INJ ? p.sinbits[(int64_t)irow * NJT + jt] : sign_word(p.si_k0, p.si_k1, (uint32_t)irow, (uint32_t)jt);
                const float y2 = Y2[jt][r] * pp.u_dh;
                v += ((w >> il) & 1u) ? -y2 : y2;
            }
            p.slab[((int64_t)(cg + p.cg_off) * p.Bpad + irow) * H + 32 * jt + il] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_out_fwd_h3e (round 6): the fp16x3 forward-ONLY pass for the loss of an evaluation step (the validation phase of src/mdl/fnn.py:143-151: 566 of a fold-epoch's 1 697
// batches at config 2) as EIGHT logit waves of k_out_fwd_h3p - no gradient wave, no hand-over, no dz.  A workgroup owns 256 batch rows (32 per wave) and walks a column
// group of 32-expert sub-tiles through the same three-stage LDS ring (2 matrices x 2 planes x [32 rows][256 B] + biases, filled by LDS-DMA one step ahead: waves 0-3
// the first matrix and the biases, waves 4-7 the second).  The two waves of a SIMD are STAGGERED by half a step: waves 0-3 run [zT MFMAs of s, logits of s], waves 4-7
// [logits of s - 1, zT MFMAs of s] - one barrier a step, matrix work of one beside vector work of the other (the pairing the training kernel has by its roles; two
// workgroups of k_out_fwd_b6 per CU get it only by accident of their phases: 0.40 ms a launch against this kernel's figure in profiles/r6_eval_ab.md).  The late waves'
// logits read their accumulators before the MFMAs of the same step overwrite them and the biases of stage s - 1, which the DMA of step s (stage s + 1) does not touch.
// Sub-tile order, products per accumulator and the logit arithmetic are wave A's: per row the loss is the same sum of the same terms in another order.
// ------------------------------------------------------------------------------------------------
template <bool BAYES, bool INJ>
__global__ __launch_bounds__(512) void k_out_fwd_h3e(OutFwd6Args pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const OutFwdArgs& p = pp.a;
    constexpr int H = 128, NJT = 4, NKS = H / 16, SUB = 32;
    constexpr int PLANE = SUB * H * 2, TM = 2 * PLANE, NMAT = BAYES ? 2 : 1, SLOT = NMAT * TM + 512;
    const int tid = threadIdx.x, lane = tid & 63, il = lane & 31, half = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), pair = wave_u & 3, late = wave_u >> 2;
    if (p.rmode == 1 && __builtin_nontemporal_load(p.rflag) != 0) return;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int cg = bid / p.NRB, rb = bid % p.NRB;      // (NRB: 256-row blocks here)
    const int tspan = p.t_hi - p.t_lo;
    const int s_beg = 2 * (p.t_lo + (int)((int64_t)cg * tspan / p.NCG)), s_end = 2 * (p.t_lo + (int)((int64_t)(cg + 1) * tspan / p.NCG));
    const int i0 = rb * 256 + wave_u * 32, i = i0 + il;
    const bool row_ok = i < p.B, row_in = i < p.Bpad;      // (the zero-padded h and the loss partials have Bpad rows, a multiple of 128)
    const uint32_t smem_base = lds_addr(smem);
    typedef const __attribute__((address_space(3))) char* ldsp_t;
    const int fil = ((il & 3) << 2) | ((il >> 2) & 3);
    auto sign_w = [&](int s) -> uint32_t {
        if (!BAYES || !row_ok) return 0u;
        const uint32_t w = INJ ? p.sbits[(int64_t)i * p.nCB + s] : sign_word(p.so_k0, p.so_k1, (uint32_t)i, (uint32_t)s);
        return w >> (4 * half);
    };
    auto mask_past_m = [&](int s) {      // (workgroup-uniform) experts past M: l = -20 through the bias
        if (32 * s + SUB > p.M) {
            if (tid < SUB && 32 * s + tid >= p.M) reinterpret_cast<float*>(smem + (s % 3) * SLOT + NMAT * TM)[tid] = -2000.f;
            __syncthreads();
        }
    };
    u32x4 hp[NKS][2], hs[NKS][2];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (row_in) { v0 = *reinterpret_cast<const float4*>(p.h + (int64_t)i * H + 16 * s + 8 * half); v1 = *reinterpret_cast<const float4*>(p.h + (int64_t)i * H + 16 * s + 8 * half + 4); }
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const uint32_t sw_in = BAYES ? (INJ ? (row_in ? p.sinbits[(int64_t)i * NJT + (s >> 1)] : 0u) : (row_ok ? sign_word(p.si_k0, p.si_k1, (uint32_t)i, (uint32_t)(s >> 1)) : 0u)) : 0u;
        const uint32_t w8 = sw_in >> (16 * (s & 1) + 8 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t pq[3];
            split_pair_np<2>(x[2 * q], x[2 * q + 1], pp.h_scale, pq);
            const uint32_t hm = ((w8 << (15 - 2 * q)) & 0x8000u) | ((w8 << (30 - 2 * q)) & 0x80000000u);
            hp[s][0][q] = pq[0]; hp[s][1][q] = pq[1];
            hs[s][0][q] = pq[0] ^ hm; hs[s][1][q] = pq[1] ^ hm;
        }
    }
    // DMA: 16 one-KiB pieces per matrix image: waves 0-3 take the first matrix (4 each) and the bias piece, waves 4-7 the second matrix
    constexpr int PER_WAVE = TM / 1024 / 4;
    uint32_t dsrc[PER_WAVE];
#pragma unroll
    for (int n = 0; n < PER_WAVE; ++n) {
        const int pos = (pair * PER_WAVE + n) * 1024 + lane * 16;
        const int plane = pos / PLANE, row = (pos >> 8) & 31, chp = (pos >> 4) & 15;
        dsrc[n] = (uint32_t)(((plane * 32 + row) * 256) + 16 * (chp ^ (((row & 3) << 2) | ((row >> 2) & 3))));
    }
    constexpr int NPIECE = PER_WAVE + 1;
    auto stage_piece = [&](int s, int slot, int n) {      // piece n of this wave's share of sub-tile s into ring slot `slot`
        const uint32_t sb = smem_base + slot * SLOT;
        if (n < PER_WAVE) {
            if (late && !BAYES) return;
            const char* base = reinterpret_cast<const char*>(late ? pp.wp_pl : pp.mu_pl) + (size_t)s * TM;
            glds16s(base, dsrc[n], sb + late * TM + (pair * PER_WAVE + n) * 1024);
        } else if (!late) {
            const int which = BAYES ? (pair & 1) : 0;
            glds4((which ? p.bp : p.mu_b) + min(32 * s + il, p.M - 1), sb + NMAT * TM + which * 256);
        }
    };
    if (s_beg < s_end) {
#pragma unroll
        for (int n = 0; n < NPIECE; ++n) stage_piece(s_beg, s_beg % 3, n);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int zrow[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) zrow[s] = 256 * il + 16 * ((2 * s + half) ^ fil);
    LossAcc lacc;
    const float rmask = row_ok ? 1.f : 0.f;
    constexpr int NHG = NKS * NMAT, BG = 2, NB = NHG / BG;
    f32x16 X1, X2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { X1[r] = 0.f; X2[r] = 0.f; }
    uint32_t sw_cur = 0u;       // s_out signs of the sub-tile whose products X1 / X2 hold
    auto products = [&](int s) {
        const uint32_t sbase = smem_base + (s % 3) * SLOT;
        ldsp_t zb[NKS];
#pragma unroll
        for (int k = 0; k < NKS; ++k) zb[k] = (ldsp_t)(size_t)(sbase + zrow[k]);
        u32x4 fb[2][BG][2];
        auto z_load = [&](int hg, u32x4 (&fr)[2]) {
            const int k = hg / NMAT, mat = hg % NMAT;
#pragma unroll
            for (int q = 0; q < 2; ++q) fr[q] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(zb[k] + (mat * TM + q * PLANE));
        };
        auto z_mma = [&](int hg, const u32x4 (&fr)[2]) {
            const int k = hg / NMAT, mat = hg % NMAT;
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            f32x16 acc = mat == 0 ? (k == 0 ? zero : X1) : (k == 0 ? zero : X2);
            const u32x4 (&b)[2] = mat == 0 ? hp[k] : hs[k];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(fr[1]), as_frag_h(b[0]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(fr[0]), as_frag_h(b[1]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag_h(fr[0]), as_frag_h(b[0]), acc, 0, 0, 0);
            if (mat == 0) X1 = acc; else X2 = acc;
        };
#pragma unroll
        for (int k = 0; k < BG; ++k) z_load(k, fb[0][k]);
        const int sn = min(s + 1, s_end - 1);
#pragma unroll
        for (int lb = 0; lb < NB; ++lb) {
            if (lb + 1 < NB) {
#pragma unroll
                for (int k = 0; k < BG; ++k) z_load((lb + 1) * BG + k, fb[(lb + 1) & 1][k]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < BG; ++k) {
                const int hg = lb * BG + k;
                z_mma(hg, fb[lb & 1][k]);
                if (hg < NPIECE) stage_piece(sn, (s + 1) % 3, hg);
            }
        }
        sw_cur = sign_w(s);
    };
    auto logits = [&](int s) {      // softplus(leaky_relu(z)) of the 32 x 32 logits in X1 / X2 (sub-tile s) into the row's loss
        const char* sb = smem + (s % 3) * SLOT;
        const uint32_t swu = sw_cur;
        float bm[16], bq[16];
        const float* bias_mu = reinterpret_cast<const float*>(sb + NMAT * TM) + 4 * half;
        const float* bias_p = reinterpret_cast<const float*>(sb + NMAT * TM + 256) + 4 * half;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 m = *reinterpret_cast<const float4*>(bias_mu + 8 * j);
            bm[4 * j] = m.x; bm[4 * j + 1] = m.y; bm[4 * j + 2] = m.z; bm[4 * j + 3] = m.w;
            if (BAYES) { const float4 q = *reinterpret_cast<const float4*>(bias_p + 8 * j); bq[4 * j] = q.x; bq[4 * j + 1] = q.y; bq[4 * j + 2] = q.z; bq[4 * j + 3] = q.w; }
        }
        float lt = 0.f;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            float l[4], tt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * j4 + j, cr = (r & 3) + 8 * (r >> 2);
                float z = fmaf(X1[r], pp.u_z, bm[r]);
                if (BAYES) z += __uint_as_float(__float_as_uint(fmaf(X2[r], pp.u_z, bq[r])) ^ ((swu << (31 - cr)) & 0x80000000u));
                l[j] = fmaxf(z > 0.f ? z : z * kLeakySlope, -21.f);      // (the clamp: see k_out_fwd_h3p - the product of four 1 + e^-l must stay finite)
                tt[j] = 1.f + __builtin_amdgcn_exp2f(l[j] * -1.4426950408889634f);
            }
            lt += fmaf(__builtin_amdgcn_logf((tt[0] * tt[1]) * (tt[2] * tt[3])), 0.6931471805599453f, (l[0] + l[1]) + (l[2] + l[3]));
        }
        lacc.tile = lt * rmask; lacc.end_tile();
    };
    for (int s = s_beg; s <= s_end; ++s) {
        if (s < s_end) mask_past_m(s);
        if (!late) {
            if (s < s_end) { products(s); logits(s); }
        } else {
            if (s > s_beg) logits(s - 1);
            if (s < s_end) products(s);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float lsum = lacc.sum;
    lsum += __shfl_xor(lsum, 32, 64);
    if (half == 0 && row_in) p.lossp[(int64_t)i * p.ncg_tot + p.cg_off + cg] = p.tnw * lsum;
}

// ------------------------------------------------------------------------------------------------
template <int H, bool BAYES>
static void fwd_dispatch(hipStream_t st, const FusedOut& f, const OutFwdArgs& a, const SpecialArgs& s, int grid, int phases) {
    constexpr int STAGE = (BAYES ? 2 : 1) * BN * 4 * H + 512;
    const size_t lds = 2 * STAGE;
    const bool dh = f.dh != nullptr;
    const bool inj = BAYES && (f.s_out.inj != nullptr || f.s_in.inj != nullptr);  // packed images instead of the hash
#define NTF_LAUNCH_FWD(TR, DHF)                                                                                           \
    do {                                                                                                                  \
        auto kf = inj ? k_out_fwd<H, BAYES, TR, DHF, BAYES> : k_out_fwd<H, BAYES, TR, DHF, false>;                        \
        if (phases & 2) {                                                                                                 \
            set_max_lds(reinterpret_cast<const void*>(kf), (int)lds); \
            hipLaunchKernelGGL(kf, dim3(grid), dim3(256), lds, st, a);                                                    \
        }                                                                                                                 \
        if (phases & 4) launch_out_special(st, H, BAYES, TR, DHF, s);                                                     \
    } while (0)
    if (!f.train) NTF_LAUNCH_FWD(false, false);
    else if (dh) NTF_LAUNCH_FWD(true, true);
    else NTF_LAUNCH_FWD(true, false);
#undef NTF_LAUNCH_FWD
}

void launch_fused_out_fwd(hipStream_t st, const FusedOut& f, int phases) {
    Geom g = geom(f.B, f.M);
    if (!f.train && f.H == 128) g.NCG = eval_ncg(g);      // forward-only launches: two workgroups per CU (k_out_fwd_b6)
    // the loss of an evaluation step in fp16x3 (round 6): k_out_fwd_h3e - 256-row workgroups of eight logit waves, one per CU.  NTF_EVAL_KERNEL=0: k_out_fwd_b6 (A/B runs, tests)
    // (Flipout only: with one matrix a sub-tile is 24 MFMAs against the same logit work, and the two workgroups per CU of k_out_fwd_b6 measured 0.309 against 0.317 ms a step)
    const bool evalp = f.eval_kernel && (f.bayes || f.eval_kernel == 2) && !f.train && !f.probs && f.bf16x6 && f.H == 128 && f.np == 2 && f.chunk_ncg_tot == 0;
    const int nrbe = (g.Bpad + 255) / 256;
    if (evalp) g.NCG = std::max(1, std::min({NCG_MAX / nrbe, g.T, NCG_MAX}));
    if (f.ncg_limit > 0) g.NCG = std::max(1, std::min(g.NCG, f.ncg_limit));
    const WsLayout w = ws_layout(f.B, f.H, f.M);
    char* ws = static_cast<char*>(f.ws);
    uint32_t* sbits = reinterpret_cast<uint32_t*>(ws + w.sbits);
    uint32_t* sinbits = reinterpret_cast<uint32_t*>(ws + w.sinbits);
    float* hs = reinterpret_cast<float*>(ws + w.hs);
    float* hz = reinterpret_cast<float*>(ws + w.hz);
    float* lossp = reinterpret_cast<float*>(ws + w.lossp);
    const bool inj = f.bayes && (f.s_out.inj != nullptr || f.s_in.inj != nullptr);
    const bool guard = f.np == 2 && f.rflag != nullptr;   // fp16x3 arithmetic somewhere in this step (forward and / or dW): range-checked operands
    if (inj && (phases & 1)) hipLaunchKernelGGL(k_sign_bits, dim3((g.nCB + 63) / 64, g.Bpad), dim3(64), 0, st, f.s_out, f.B, f.M, g.nCB, sbits);
    if ((phases & 1) && !f.h_ready) {
        const int n = g.Bpad * (f.H / 32);
        hipLaunchKernelGGL(k_prep_h, dim3((n + 63) / 64), dim3(64), 0, st, f.s_in, f.bayes, f.h, f.B, f.H, g.Bpad, sinbits, hs, hz,
                           guard ? 65504.f / f.h_scale : 0.f, guard ? f.rflag : nullptr);
    }
    OutFwdArgs a;
    a.B = f.B; a.M = f.M; a.Bpad = g.Bpad; a.NRB = g.NRB; a.NCG = g.NCG; a.T = g.T; a.nCB = g.nCB;
    a.h = hz; a.hs = hs; a.mu = f.mu; a.mu_b = f.mu_b; a.wp = f.wp; a.bp = f.bp; a.sbits = sbits; a.sinbits = sinbits;
    a.tnw = f.tnw; a.inv_B = f.inv_B; a.dzT = f.dzT; a.slab = f.dh_slab; a.lossp = lossp;
    a.so_k0 = f.s_out.k0; a.so_k1 = f.s_out.k1; a.si_k0 = f.s_in.k0; a.si_k1 = f.s_in.k1; a.so_inj = f.s_out.inj != nullptr; a.si_inj = f.s_in.inj != nullptr;
    a.rflag = f.rflag; a.rmode = 0;
    a.t_lo = 0; a.t_hi = g.T; a.cg_off = 0; a.ncg_tot = g.NCG;
    const bool ranged = f.chunk_ncg_tot > 0;      // the split-product forward of this step runs (ran) as launches over ranges of the experts (FusedOut.chunk_*)
    SpecialArgs s;
    s.B = f.B; s.M = f.M; s.Bpad = g.Bpad; s.NCG = ranged ? f.chunk_ncg_tot : g.NCG; s.nCB = g.nCB; s.ns = f.ns;
    s.nslab = s.NCG; s.fb_ncg = ranged ? g.NCG : 0;
    s.h = f.h; s.hs = hs; s.mu = f.mu; s.mu_b = f.mu_b; s.wp = f.wp; s.bp = f.bp; s.slab = f.dh_slab; s.lossp = lossp; s.h_mask = f.h_mask;
    s.sbits = sbits; s.sinbits = sinbits; s.rows = f.rows; s.m_indptr = f.m_indptr; s.neg = f.neg; s.m_indices = f.m_indices;
    s.tpw = f.tpw; s.tnw = f.tnw; s.inv_B = f.inv_B; s.dzT = f.dzT; s.dh = f.dh; s.row_fix = f.row_fix;
    s.so_k0 = f.s_out.k0; s.so_k1 = f.s_out.k1; s.so_inj = inj;
    s.dz_pack_scale = (f.train && f.bf16x6 && f.H == 128 && f.np == 2) ? f.dz_scale : 0.f; s.rflag = f.rflag; s.c_lo = f.c_lo;
    s.wp_pl = (f.bayes && f.bf16x6 && f.H == 128 && f.np == 2 && f.wp_pl) ? f.wp_pl : nullptr; s.wp_inv_scale = 1.f / f.w_scale;
    int grid = g.NRB * g.NCG;
    const bool range_launch = ranged && f.chunk_ncg > 0 && (phases & 2);      // THIS call launches one range
    if (range_launch) { a.t_lo = f.chunk_t_lo; a.t_hi = f.chunk_t_hi; a.cg_off = f.chunk_cg_off; a.ncg_tot = f.chunk_ncg_tot; a.NCG = f.chunk_ncg; grid = g.NRB * f.chunk_ncg; }
    if (f.bf16x6 && f.H == 128) {
        constexpr int np = 2;
        if ((phases & 1) && !f.planes_ready) {
            const int64_t Mp = ((int64_t)f.M + BN6 - 1) / BN6 * BN6, n = Mp * (f.H / 2);
            hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f.mu, f.M, f.H, np, f.w_scale, f.mu_pl, guard ? f.rflag : nullptr);
            if (f.bayes) hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f.wp, f.M, f.H, np, f.w_scale, f.wp_pl, guard ? f.rflag : nullptr);
        }
        if (phases & 2) {
            OutFwd6Args a6; a6.stamps = nullptr; a6.a = a; a6.mu_pl = f.mu_pl; a6.wp_pl = f.wp_pl; a6.pscale = f.pscale; a6.pacc = f.pacc; a6.plogit = f.plogit;
            a6.a.rmode = (guard && np == 2) ? 1 : 0;
            a6.h_scale = np == 2 ? f.h_scale : 1.f; a6.dz_scale = np == 2 ? f.dz_scale : 1.f;
            a6.u_z = np == 2 ? 1.f / (f.w_scale * f.h_scale) : 1.f; a6.u_dh = np == 2 ? 1.f / (f.dz_scale * f.w_scale) : 1.f;
            const bool dh = f.dh != nullptr;
            const size_t lds = (size_t)2 * ((size_t)(f.bayes ? 2 : 1) * np * BN6 * 128 * 2 + 512);
#define NTF_L6N(BY, TR, DHF, IJ, PR, NPV) do { auto kf = k_out_fwd_b6<BY, TR, DHF, IJ, PR, NPV>;                               \
            set_max_lds(reinterpret_cast<const void*>(kf), (int)lds);       \
            hipLaunchKernelGGL(kf, dim3(grid), dim3(256), lds, st, a6); } while (0)
#define NTF_L6(BY, TR, DHF, IJ, PR) NTF_L6N(BY, TR, DHF, IJ, PR, 2)      /* (NP = 3, the bf16x6 arithmetic, is no longer instantiated: retired in round 6) */
#define NTF_L6B(BY, IJ) do { if (f.probs) NTF_L6(BY, false, false, IJ, true); else if (!f.train) NTF_L6(BY, false, false, IJ, false);  \
                             else if (dh) NTF_L6(BY, true, true, IJ, false); else NTF_L6(BY, true, false, IJ, false); } while (0)
            if (evalp) {
                const size_t ldse = 3 * ((size_t)(f.bayes ? 2 : 1) * 2 * 32 * 128 * 2 + 512);
                OutFwd6Args ae = a6; ae.a.NRB = nrbe;
#define NTF_LE(BY, IJ) do { auto kf = k_out_fwd_h3e<BY, IJ>;                                                                    \
                set_max_lds(reinterpret_cast<const void*>(kf), (int)ldse);  \
                hipLaunchKernelGGL(kf, dim3(nrbe * g.NCG), dim3(512), ldse, st, ae); } while (0)
                if (f.bayes) { if (inj) NTF_LE(true, true); else NTF_LE(true, false); } else NTF_LE(false, false);
#undef NTF_LE
            } else
            if (np == 2 && f.train && dh && f.wide == 5) {    // the fp16x3 training step: producer / consumer wave pairs, two waves per SIMD (a.T counts 64-expert tiles)
                {
                    const size_t ldsp = 3 * ((size_t)(f.bayes ? 2 : 1) * 2 * 32 * 128 * 2 + 512) + 2 * 16384;
#define NTF_LP(BY, IJ) do { auto kf = k_out_fwd_h3p<BY, IJ>;                                                                    \
                set_max_lds(reinterpret_cast<const void*>(kf), (int)ldsp);  \
                hipLaunchKernelGGL(kf, dim3(grid), dim3(512), ldsp, st, a6); } while (0)
#ifdef NTF_DIAG
                    static const int fwd_abl5 = getenv("NTF_FWD_ABL") ? atoi(getenv("NTF_FWD_ABL")) : 0;
                    if (fwd_abl5 == 9 && f.bayes && !inj) {
                        static unsigned long long* d_st = nullptr; static int n_launch = 0;
                        if (!d_st) hipMalloc(&d_st, (size_t)grid * 8 * 10 * 8);
                        a6.stamps = d_st;
                        auto kf = k_out_fwd_h3p<true, false, true>;
                        set_max_lds(reinterpret_cast<const void*>(kf), (int)ldsp);
                        hipLaunchKernelGGL(kf, dim3(grid), dim3(512), ldsp, st, a6);
                        if (++n_launch == 30) {
                            std::vector<unsigned long long> hst((size_t)grid * 80);
                            hipStreamSynchronize(st); hipMemcpy(hst.data(), d_st, hst.size() * 8, hipMemcpyDeviceToHost);
                            double cyc = 0, wall = 0;
                            for (size_t w = 0; w < (size_t)grid * 8; ++w) { cyc += (double)hst[w * 10 + 7]; wall += (double)hst[w * 10 + 8]; }
                            fprintf(stderr, "[pair stamps] per wave, kernel entry to the end of its loop: %.0f shader-clock cycles in %.0f ticks of 100 MHz = %.3f GHz, %.3f ms\n",
                                    cyc / (grid * 8.0), wall / (grid * 8.0), cyc / (wall * 10.0), wall / (grid * 8.0) / 1e5);
                            for (int role = 0; role < 2; ++role) {
                                double sum[7] = {0};
                                for (int wg = 0; wg < grid; ++wg) for (int w = 4 * role; w < 4 * role + 4; ++w) for (int q = 0; q < 7; ++q) sum[q] += (double)hst[((size_t)wg * 8 + w) * 10 + q];
                                if (role == 0) fprintf(stderr, "[pair stamps] waves 0-3, cycles per step: top+first reads %.0f | MFMAs+DMA %.0f | dz stores %.0f | logits %.0f | wait %.0f | barrier %.0f  (steps/wave %.1f)\n",
                                        sum[4] / sum[6], sum[0] / sum[6], sum[5] / sum[6], sum[1] / sum[6], sum[2] / sum[6], sum[3] / sum[6], sum[6] / (grid * 4.0));
                                else fprintf(stderr, "[pair stamps] waves 4-7, cycles per step: top+reads %.0f | dz %.0f | dh MFMAs %.0f | wait %.0f | barrier %.0f\n",
                                        sum[4] / sum[6], sum[0] / sum[6], sum[1] / sum[6], sum[2] / sum[6], sum[3] / sum[6]);
                            }
                        }
                    } else
#endif
                    if (f.bayes) { if (inj) NTF_LP(true, true); else NTF_LP(true, false); } else NTF_LP(false, false);
#undef NTF_LP
                }
            } else if (f.bayes) { if (inj) NTF_L6B(true, true); else NTF_L6B(true, false); } else NTF_L6B(false, false);
#undef NTF_L6B
#undef NTF_L6
#undef NTF_L6N
#ifdef NTF_DIAG
            static const bool skip_fb = getenv("NTF_SKIP_FALLBACK") != nullptr;     // timing only: what the conditional exact-f32 launch behind k_out_fwd_h3p costs
#else
            constexpr bool skip_fb = false;
#endif
            if (guard && np == 2 && !f.probs && !skip_fb && !f.split_fallback && !range_launch) {   // the same pass on the exact-f32 kernel, run only when an operand left the fp16 window
                OutFwdArgs af = a; af.rmode = 2;
                if (f.bayes) fwd_dispatch<128, true>(st, f, af, s, grid, 2); else fwd_dispatch<128, false>(st, f, af, s, grid, 2);
            }
        }
        if ((phases & 8) && f.split_fallback && guard && f.np == 2 && !f.probs) {   // ... as a launch of its own: behind the last range of a ranged (data-parallel) step, once over the whole layer
            OutFwdArgs af = a; af.rmode = 2;
            if (f.bayes) fwd_dispatch<128, true>(st, f, af, s, grid, 2); else fwd_dispatch<128, false>(st, f, af, s, grid, 2);
        }
        if ((phases & 4) && !f.probs) {
            launch_out_special(st, 128, f.bayes != 0, f.train != 0, f.dh != nullptr, s);
        }
        return;
    }
#define NTF_H(HH) do { if (f.bayes) fwd_dispatch<HH, true>(st, f, a, s, grid, phases); else fwd_dispatch<HH, false>(st, f, a, s, grid, phases); } while (0)
    if (f.H == 128) NTF_H(128); else if (f.H == 64) NTF_H(64); else NTF_H(32);
#undef NTF_H
}

// inference helpers: ent[i] += sum over the column groups of the per-row entropy partials; P[i][c] = PT[c][i] (tiled transpose through LDS)
__global__ void k_ent_slots(const float* __restrict__ lossp, int B, int NCG, float scale, float* __restrict__ ent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float s = 0.f;
    for (int cg = 0; cg < NCG; ++cg) s += lossp[(int64_t)i * NCG + cg];
    ent[i] += s * scale;
}
__global__ __launch_bounds__(256) void k_transpose_pt(const float* __restrict__ PT, int M, int Bpad, int B, float* __restrict__ P, float unpack_inv_scale) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, i0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        float v = 0.f;
        if (c < M) { v = PT[dzt_index(c, i0 + tx, Bpad)]; if (unpack_inv_scale > 0.f) v = unpack_planes(__float_as_uint(v), unpack_inv_scale); }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int i = i0 + r, c = c0 + tx; if (i < B && c < M) P[(int64_t)i * M + c] = tile[tx][r]; }
}
void launch_fused_probs_finish(hipStream_t st, int B, int H, int M, void* ws_, const float* PT, float* P, float* ent_rows /*nullable: += this pass * scale*/, float scale, bool transpose,
                               float unpack_inv_scale) {
    Geom g = geom(B, M);
    if (H == 128) g.NCG = eval_ncg(g);      // (the inference launches' column groups, launch_fused_out_fwd)
    const WsLayout w = ws_layout(B, H, M);
    if (ent_rows) hipLaunchKernelGGL(k_ent_slots, dim3((B + 255) / 256), dim3(256), 0, st, reinterpret_cast<const float*>(static_cast<char*>(ws_) + w.lossp), B, g.NCG, scale, ent_rows);
    if (transpose) hipLaunchKernelGGL(k_transpose_pt, dim3((M + 31) / 32, g.Bpad / 32), dim3(256), 0, st, PT, M, g.Bpad, B, P, unpack_inv_scale);
}

}  // namespace ntf
