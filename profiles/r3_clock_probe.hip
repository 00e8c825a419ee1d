// clock / MFMA-rate probe: shader clock (clock64) against the 100 MHz wall clock under different loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(512) void k(long long* out, int iters, float* sink) {
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 0.001f + q); b[q] = (_Float16)(q * 0.5f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    float v = threadIdx.x;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { for (int q = 0; q < 16; ++q) v = v * 1.0001f + 0.5f; }
        else if (MODE == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else if (MODE == 2) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0);
        } else if (MODE == 3) {   // dependent chain 32x32
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        } else if (MODE == 4) {   // dependent chain 16x16
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0);
        } else if (MODE == 5) {   // 32x32 MFMAs + 5 VALU per MFMA
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); for (int q = 0; q < 5; ++q) v = v * 1.0001f + 0.5f;
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0); for (int q = 0; q < 5; ++q) v = v * 1.0001f + 0.5f;
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); for (int q = 0; q < 5; ++q) v = v * 1.0001f + 0.5f;
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0); for (int q = 0; q < 5; ++q) v = v * 1.0001f + 0.5f;
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = v;
    for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q] + c3[q];
    for (int q = 0; q < 4; ++q) s += d0[q] + d1[q] + d2[q] + d3[q];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int MODE> void run(const char* name, int threads, int iters, int per_iter) {
    const int grid = 256;
    long long* d; float* sink; hipMalloc(&d, grid * 16); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, threads>>>(d, iters / 10, sink);
    hipEventRecord(e0);
    k<MODE><<<grid, threads>>>(d, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * 2); hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0; for (int b = 0; b < grid; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
    cyc /= grid; wall /= grid;
    printf("%-34s threads %3d: %.3f ms, clock64 %.0f cycles, wall %.0f ticks (100 MHz -> %.3f ms) => %.3f GHz; cycles per MFMA per wave %.2f\n", name, threads, ms, cyc, wall, wall / 1e5,
           cyc / (wall * 10.0), per_iter ? cyc / ((double)iters * per_iter) : 0.0);
    hipFree(d); hipFree(sink);
}
int main() {
    run<0>("VALU only", 256, 2000000, 0);
    run<1>("32x32x16 f16, 4 independent", 256, 1000000, 4);
    run<1>("32x32x16 f16 x20 (1.1 s)", 256, 20000000, 4);
    run<1>("32x32x16 f16 x20 again", 256, 20000000, 4);
    run<5>("32x32x16 + 5 VALU each x10", 256, 10000000, 4);
    run<2>("16x16x32 f16 x10", 256, 20000000, 4);
    return 0;
}
