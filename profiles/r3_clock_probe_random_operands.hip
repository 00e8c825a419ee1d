// does a dense MFMA stream on CHANGING random operands reach the power limit by itself?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE> __global__ __launch_bounds__(256) void k(long long* out, int iters, float* sink, const _Float16* src) {
    f16x8 a[8], b[8];
    for (int j = 0; j < 8; ++j) for (int q = 0; q < 8; ++q) {
        a[j][q] = MODE == 0 ? (_Float16)1.0f : src[(threadIdx.x * 64 + j * 8 + q) & 65535];
        b[j][q] = MODE == 0 ? (_Float16)0.5f : src[(threadIdx.x * 64 + j * 8 + q + 32768) & 65535];
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[j], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 1], b[j + 1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 2], b[j + 2], c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 3], b[j + 3], c3, 0, 0, 0);
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q] + c3[q];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int MODE> void run(const char* name, int iters, const _Float16* src) {
    const int grid = 256;
    long long* d; float* sink; hipMalloc(&d, grid * 16); hipMalloc(&sink, 4);
    k<MODE><<<grid, 256>>>(d, iters / 10, sink, src);
    k<MODE><<<grid, 256>>>(d, iters, sink, src);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * 2); hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0; for (int b = 0; b < grid; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
    printf("%-44s %.0f ms: %.3f GHz, %.2f cycles per MFMA per wave\n", name, wall / grid / 1e5, cyc / (wall * 10.0), cyc / grid / ((double)iters * 8));
    hipFree(d); hipFree(sink);
}
int main() {
    std::vector<_Float16> h(65536);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 8) % 20001 - 10000) * 1e-3f); }
    _Float16* src; hipMalloc(&src, 65536 * 2); hipMemcpy(src, h.data(), 65536 * 2, hipMemcpyHostToDevice);
    run<0>("32x32x16 f16, constant operands (1.0, 0.5)", 10000000, src);
    run<1>("32x32x16 f16, 8 random operand sets cycled", 10000000, src);
    run<1>("the same again", 10000000, src);
    return 0;
}
