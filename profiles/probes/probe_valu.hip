// probe: cycles per vector instruction of ONE wave per SIMD, as a function of how many independent dependency chains are interleaved
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int NCH, int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float a, float b) {
    float x[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) x[c] = threadIdx.x * 0.001f + c;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / NCH; ++r) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[c]));
                if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1\n\tv_exp_f32 %0, %0\n\ts_nop 0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0\n\ts_nop 0" : "+v"(x[c]) : "v"(a));
                if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_lt_f32 vcc, 0, %0" : "+v"(x[c]) : "v"(a) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NCH, int KIND> void run(const char* name, int per) {
    float* o; unsigned long long* c; hipMalloc(&o, 256 * 256 * 4); hipMalloc(&c, 8);
    hipLaunchKernelGGL((k<NCH, KIND>), dim3(256), dim3(256), 0, 0, o, c, 0.999f, 0.001f);
    hipLaunchKernelGGL((k<NCH, KIND>), dim3(256), dim3(256), 0, 0, o, c, 0.999f, 0.001f);
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s chains %d: %.2f cycles per instruction\n", name, NCH, (double)h / (256.0 * (64 / NCH) * NCH * per));
    hipFree(o); hipFree(c);
}
int main() {
    run<1, 0>("v_fma_f32", 1); run<2, 0>("v_fma_f32", 1); run<4, 0>("v_fma_f32", 1); run<8, 0>("v_fma_f32", 1);
    run<1, 1>("v_exp_f32", 1); run<2, 1>("v_exp_f32", 1); run<4, 1>("v_exp_f32", 1);
    run<1, 2>("mul exp nop add rcp nop", 6); run<2, 2>("mul exp nop add rcp nop", 6); run<4, 2>("mul exp nop add rcp nop", 6);
    run<1, 3>("cndmask + cmp (vcc)", 2); run<2, 3>("cndmask + cmp (vcc)", 2);
    return 0;
}
