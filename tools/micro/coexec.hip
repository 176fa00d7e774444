// Do MFMA and VALU streams of two waves on ONE SIMD overlap on gfx950?  512-thread blocks: waves 0-3 and 4-7 share SIMDs 0-3.
// mode bits: wave-half A (wid<4) and B (wid>=4) each run one of: 0 idle, 1 MFMA 32x32x16 chain(s), 2 v_fma, 3 v_exp, 4 mixed exp+cvt+max (softmax-like)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__device__ __forceinline__ float body(int iters, float seed) {
    float acc = 0;
    if (KIND == 1) {
        f32x16 c0, c1;
        for (int e = 0; e < 16; ++e) { c0[e] = seed; c1[e] = seed; }
        half8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)seed; b[e] = (_Float16)0.5f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {      // two independent chains: the pipe is never dependency-stalled
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            }
        }
        for (int e = 0; e < 16; ++e) acc += c0[e] + c1[e];
    } else if (KIND == 5 || KIND == 6 || KIND == 8) {
        f32x16 c0, c1;
        for (int e = 0; e < 16; ++e) { c0[e] = seed; c1[e] = seed; }
        half8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)seed; b[e] = (_Float16)0.5f; }
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                if (KIND == 5) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
                } else if (KIND == 6) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[1]));
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[2]) : "v"(seed));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
                }
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                if (KIND == 5) {
#pragma unroll
                    for (int i = 4; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
                } else if (KIND == 6) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[3]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v[4]));
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[5]) : "v"(seed));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
                }
            }
        }
        for (int e = 0; e < 16; ++e) acc += c0[e] + c1[e];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    } else if (KIND == 7) {
        f32x16 c0;
        for (int e = 0; e < 16; ++e) c0[e] = seed;
        half8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)seed; b[e] = (_Float16)0.5f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        }
        for (int e = 0; e < 16; ++e) acc += c0[e];
    } else if (KIND >= 2) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = seed + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
                if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
    }
    return acc;
}

template <int KA, int KB, int PA = 0, int PB = 0>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float r;
    if (wid < 4) { if (PA) __builtin_amdgcn_s_setprio(PA); r = body<KA>(iters, seed); } else { if (PB) __builtin_amdgcn_s_setprio(PB); r = body<KB>(iters, seed); }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KA, int KB, int PA = 0, int PB = 0>
float run(const char* name, int iters) {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<KA, KB, PA, PB><<<256, 512>>>(d, iters, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<KA, KB, PA, PB><<<256, 512>>>(d, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms\n", name, ms);
    (void)hipFree(d);
    return ms;
}

int main() {
    const int it = 20000;
    run<1, 0>("A=MFMA      B=idle", it);
    run<0, 2>("A=idle      B=fma", it);
    run<0, 3>("A=idle      B=exp", it);
    run<1, 1>("A=MFMA      B=MFMA", it);
    run<2, 2>("A=fma       B=fma", it);
    run<1, 2>("A=MFMA      B=fma", it);
    run<1, 3>("A=MFMA      B=exp", it);
    run<3, 3>("A=exp       B=exp", it);
    run<2, 3>("A=fma       B=exp", it);
    run<1, 2, 0, 1>("A=MFMA      B=fma(prio1)", it);
    run<1, 2, 1, 0>("A=MFMA(p1)  B=fma", it);
    run<1, 3, 0, 1>("A=MFMA      B=exp(prio1)", it);
    run<7, 0>("A=MFMAdep   B=idle", it);
    run<7, 2>("A=MFMAdep   B=fma", it);
    run<7, 7>("A=MFMAdep   B=MFMAdep", it);
    run<5, 0>("A=MFMA+4fma B=idle", it);
    run<8, 0>("A=MFMA+8fma B=idle", it);
    run<6, 0>("A=MFMA+2exp+fma B=idle", it);
    run<5, 5>("A=MFMA+4fma B=same", it);
    run<6, 6>("A=MFMA+2exp+fma B=same", it);
    run<5, 2>("A=MFMA+4fma B=fma", it);
    return 0;
}
