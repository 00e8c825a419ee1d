// probe: split_packed (v_cvt_f16_f32 + v_fma_mixhi_f16) against convert / subtract / convert / permute, bit for bit
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
__device__ __forceinline__ uint32_t split_packed(float x) {
    uint32_t d;
    asm("v_cvt_f16_f32 %0, %1\n\tv_fma_mixhi_f16 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "=&v"(d) : "v"(x));
    return d;
}
__device__ __forceinline__ uint32_t split_ref(float x) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t xx = {x, 0.f};
    const h2_t a1 = __builtin_convertvector(xx, h2_t);
    const f2_t r = xx - __builtin_convertvector(a1, f2_t);
    const h2_t a2 = __builtin_convertvector(r, h2_t);
    const uint32_t p1 = __builtin_bit_cast(uint32_t, a1), p2 = __builtin_bit_cast(uint32_t, a2);
    return (p1 & 0xFFFFu) | (p2 << 16);
}
__global__ void k(const float* x, uint32_t* a, uint32_t* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = split_packed(x[i]); b[i] = split_ref(x[i]); }
}
int main() {
    const int n = 1 << 24;
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        uint32_t u = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        if (i % 3 == 0) { float f = (rand() / (float)RAND_MAX - 0.5f) * 32768.f; memcpy(&u, &f, 4); }          // the range dz * dz_scale lives in
        if (i % 3 == 1) { float f = (rand() / (float)RAND_MAX - 0.5f) * 1e-4f; memcpy(&u, &f, 4); }            // around the fp16 subnormals
        float f; memcpy(&f, &u, 4);
        if (!(f == f) || f > 65000.f || f < -65000.f) f = 0.f;
        h[i] = f;
    }
    float* dx; uint32_t *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    uint32_t* ha = (uint32_t*)malloc(n * 4); uint32_t* hb = (uint32_t*)malloc(n * 4);
    hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) if (ha[i] != hb[i]) { if (bad < 10) printf("x=%g (%08x): mix %08x ref %08x\n", h[i], *(uint32_t*)&h[i], ha[i], hb[i]); ++bad; }
    printf("split_packed probe: %ld of %d differ\n", bad, n);
    return bad != 0;
}
