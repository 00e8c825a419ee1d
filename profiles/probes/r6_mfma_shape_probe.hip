// Round 6 probe for DESIGN.md section 8 #2 ("another tile rather than another schedule"): would the forward kernel's matrix work run faster as 16x16x32 MFMAs?
// MI355X_MICROARCH.md (DVFS give-back, item 7) reports 1.12-1.15 x the FLOP/s for bare bf16 loops of v_mfma_f32_16x16x32 against 32x32x16 at equal cycles per FLOP - the chip holds
// a higher clock.  Here: the same comparison for fp16, two waves per SIMD (512-thread workgroups, one per CU), every A operand re-read from LDS (random data) and F independent
// vector instructions per 32x32x16-equivalent of matrix work - k_out_fwd_h3p's regime (~5 per MFMA, 1 KB of fragments per MFMA and wave).
//   SHAPE 0: four v_mfma_f32_32x32x16_f16 per iteration (4 x 16 Ki MACs, 32 cycles each);  SHAPE 1: eight v_mfma_f32_16x16x32_f16 (8 x 8 Ki MACs, 16 cycles each) on the same
// four 1-KiB fragments - both 4 "units" of 16 Ki MACs per iteration, 512 MACs per cycle and SIMD.
//   hipcc -O3 --offload-arch=gfx950 profiles/probes/r6_mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int F> __global__ __launch_bounds__(512) void k(long long* out, int iters, float* sink, const _Float16* src) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[65536];      // 128 KiB: the forward kernel's footprint
    for (int i = threadIdx.x; i < 65536; i += 512) lds[i] = src[i];
    __syncthreads();
    f16x8 b[4];
    for (int j = 0; j < 4; ++j) for (int q = 0; q < 8; ++q) b[j][q] = src[(threadIdx.x * 32 + j * 8 + q + 32768) & 65535];
    const f16x8* lp = reinterpret_cast<const f16x8*>(lds) + (threadIdx.x & 63);
    f32x16 c0 = {}, c1 = {};
    f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {}, d4 = {}, d5 = {}, d6 = {}, d7 = {};
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = threadIdx.x + q;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const int o = ((it * 4) & 63) * 64;      // wave-rows of 1 KiB: 4 fresh fragments per iteration
        const f16x8 a0 = lp[o], a1 = lp[o + 64], a2 = lp[o + 128], a3 = lp[o + 192];
        if (SHAPE == 0) {      // 4 units of 16 Ki MACs
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b[2], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, b[3], c1, 0, 0, 0);
        } else {               // 8 MFMAs of 8 Ki MACs = 4 units; the same 4 KiB of fragments (a 16x16x32 A operand is also 16 bytes a lane)
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[0], d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[1], d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[2], d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[3], d3, 0, 0, 0);
            d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b[0], d4, 0, 0, 0); d5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b[1], d5, 0, 0, 0);
            d6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b[2], d6, 0, 0, 0); d7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b[3], d7, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4 * F; ++q) v[q & 7] = v[q & 7] * 1.0001f + 0.5f;      // F independent-ish vector instructions per unit (8 chains)
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += v[q];
    for (int q = 0; q < 16; ++q) s += c0[q] + c1[q];
    for (int q = 0; q < 4; ++q) s += d0[q] + d1[q] + d2[q] + d3[q] + d4[q] + d5[q] + d6[q] + d7[q];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * 8 + (threadIdx.x >> 6))] = t1 - t0; out[2 * (blockIdx.x * 8 + (threadIdx.x >> 6)) + 1] = w1 - w0; }
}
template <int SHAPE, int F> void run(const char* name, int iters, const _Float16* src) {
    const int grid = 256;
    long long* d; float* sink; hipMalloc(&d, grid * 8 * 16); hipMalloc(&sink, 4);
    k<SHAPE, F><<<grid, 512>>>(d, iters / 10, sink, src);
    k<SHAPE, F><<<grid, 512>>>(d, iters, sink, src);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * 16); hipMemcpy(h.data(), d, grid * 8 * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0; for (int b = 0; b < grid * 8; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
    const double n = grid * 8.0, units = (double)iters * 4;
    // 2 waves per SIMD share the pipe: cycles per unit and SIMD = wave cycles / units / 2 ... reported per wave; TFLOP/s over the chip: 1024 SIMDs x 2 waves
    const double secs = wall / n / 1e8;
    printf("%-34s F = %d: %7.1f ms  %.3f GHz  %6.1f cycles per 16-Ki-MAC unit and wave  %7.1f TFLOP/s (chip)\n", name, F, secs * 1e3, cyc / (wall * 10.0), cyc / n / units,
           units * 2.0 * 16384.0 * (grid * 8.0) / secs / 1e12);
    hipFree(d); hipFree(sink);
}
int main() {
    std::vector<_Float16> h(65536);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 8) % 20001 - 10000) * 1e-3f); }
    _Float16* src; hipMalloc(&src, 65536 * 2); hipMemcpy(src, h.data(), 65536 * 2, hipMemcpyHostToDevice);
    const int it = 2000000;
    run<0, 0>("v_mfma_f32_32x32x16_f16", it, src); run<1, 0>("v_mfma_f32_16x16x32_f16", it, src);
    run<0, 3>("v_mfma_f32_32x32x16_f16", it, src); run<1, 3>("v_mfma_f32_16x16x32_f16", it, src);
    run<0, 5>("v_mfma_f32_32x32x16_f16", it, src); run<1, 5>("v_mfma_f32_16x16x32_f16", it, src);
    run<0, 8>("v_mfma_f32_32x32x16_f16", it, src); run<1, 8>("v_mfma_f32_16x16x32_f16", it, src);
    return 0;
}
