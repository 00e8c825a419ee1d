// how much of the power budget do the LDS fragment reads take?  MFMA stream whose A operands are re-read from LDS (random data) at the forward kernel's ratio
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: operands in registers (8 random sets).  MODE 1: every MFMA's A operand freshly read from LDS (1 KB per wave and MFMA).  MODE 2: two of three.  MODE 3: MODE 1 + 5 VALU per MFMA
template <int MODE> __global__ __launch_bounds__(256) void k(long long* out, int iters, float* sink, const _Float16* src) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[32768];
    for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = src[i];
    __syncthreads();
    f16x8 a[8], b[8];
    for (int j = 0; j < 8; ++j) for (int q = 0; q < 8; ++q) { a[j][q] = src[(threadIdx.x * 64 + j * 8 + q) & 65535]; b[j][q] = src[(threadIdx.x * 64 + j * 8 + q + 32768) & 65535]; }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const f16x8* lp = reinterpret_cast<const f16x8*>(lds) + (threadIdx.x & 63);
    float v = threadIdx.x;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const int o = ((it * 8 + j) & 31) * 64;     // 64 wave-rows of 1 KB
            f16x8 a0 = a[j], a1 = a[j + 1], a2 = a[j + 2], a3 = a[j + 3];
            if (MODE >= 1) { a0 = lp[o]; a1 = lp[o + 64]; if (MODE != 2) { a2 = lp[o + 128]; } a3 = lp[o + 192]; }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[j], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[j + 1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b[j + 2], c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, b[j + 3], c3, 0, 0, 0);
            if (MODE == 3) { for (int q = 0; q < 20; ++q) v = v * 1.0001f + 0.5f; }
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = v;
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
    printf("%-72s %.0f ms: %.3f GHz, %.2f cycles per MFMA per wave -> %.0f MFMA/us per SIMD\n", name, wall / grid / 1e5, cyc / (wall * 10.0), cyc / grid / ((double)iters * 8), (double)iters * 8 / (wall / grid / 100.0));
    hipFree(d); hipFree(sink);
}
int main() {
    std::vector<_Float16> h(65536);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 8) % 20001 - 10000) * 1e-3f); }
    _Float16* src; hipMalloc(&src, 65536 * 2); hipMemcpy(src, h.data(), 65536 * 2, hipMemcpyHostToDevice);
    run<0>("operands in registers (8 random sets)", 5000000, src);
    run<2>("A operand of 3 MFMAs in 4 re-read from LDS (768 B per MFMA and wave)", 5000000, src);
    run<1>("A operand of every MFMA re-read from LDS (1 KB per MFMA and wave)", 5000000, src);
    run<3>("... + 5 dependent VALU per MFMA", 5000000, src);
    return 0;
}
