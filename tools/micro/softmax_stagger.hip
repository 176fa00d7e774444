// Does a half-period stagger between the waves of one SIMD let the softmax (VALU) phase of one wave run beside the MFMA phase of its
// partner?  Same register-level instruction mix as softmax_mix.hip (one d=40 attention subtile: 3 QK^T MFMAs -> 16-wide max -> 16 v_exp ->
// 8 cvt_pk -> 4 PV MFMAs), no memory.  Waves w and w+4 (+8, +12) of a block share a SIMD; group g = wave >> 2 starts g * delay later.
// Prints ns and cycles (in-kernel clock) per subtile per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PRIO, int BAR>   // PRIO: 0 none, 1 setprio(1) around the softmax, 2 static: odd groups at priority 1;  BAR: s_barrier every unit (0/1)
__global__ __launch_bounds__(1024) void mix(float* out, unsigned long long* clk, int iters, float seed, int delay) {
    f32x16 o0, o1, negm;
    for (int e = 0; e < 16; ++e) { o0[e] = 0; o1[e] = 0; negm[e] = -seed; }
    half8 k0, k1, k2, q0, q1, q2, v0, v1, v2, v3;
    for (int e = 0; e < 8; ++e) {
        k0[e] = (_Float16)(seed + e); k1[e] = (_Float16)(seed - e); k2[e] = (_Float16)seed;
        q0[e] = (_Float16)0.01f; q1[e] = (_Float16)0.02f; q2[e] = (_Float16)0.03f;
        v0[e] = (_Float16)1.f; v1[e] = (_Float16)2.f; v2[e] = (_Float16)3.f; v3[e] = (_Float16)4.f;
    }
    float thr = 1e30f * seed;
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    if (PRIO == 2 && (grp & 1)) __builtin_amdgcn_s_setprio(1);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < grp * delay; ++i) asm volatile("s_nop 15");
    for (int it = 0; it < iters; ++it) {
        f32x16 s;
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s) : "v"(k0), "v"(q0), "v"(negm));
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q1, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k2, q2, s, 0, 0, 0);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int e = 3; e + 1 < 16; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);
        mx = fmaxf(mx, s[15]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        if (__any(mx > thr)) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] -= mx; o0[e] *= 0.5f; o1[e] *= 0.5f; }
        }
        half8 pf0, pf1;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const float p0 = __builtin_amdgcn_exp2f(s[e]);
            const float p1 = __builtin_amdgcn_exp2f(s[e + 1]);
            const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(p0, p1));
            if (e < 8) { pf0[e] = h2[0]; pf0[e + 1] = h2[1]; } else { pf1[e - 8] = h2[0]; pf1[e - 7] = h2[1]; }
        }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf0, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf0, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v2, pf1, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v3, pf1, o1, 0, 0, 0);
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int e = 0; e < 16; ++e) acc += o0[e] + o1[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int PRIO, int BAR>
void run(const char* name, int wps, int delay) {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4 * 2);
    unsigned long long* c; (void)hipMalloc(&c, 64);
    const int iters = 20000;
    const int threads = 256 * wps;             // one block per CU: waves w, w+4, ... share a SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mix<PRIO, BAR><<<256, threads>>>(d, c, iters, 0.5f, delay);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mix<PRIO, BAR><<<256, threads>>>(d, c, iters, 0.5f, delay);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    const double ns = ms * 1e6 / ((double)iters * wps);
    printf("%-40s w/SIMD=%d delay=%3d: %7.3f ms %6.1f ns/unit/SIMD = %5.0f cycles at %.2f GHz\n", name, wps, delay * 16, ms, ns, ns * ghz, ghz);
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    for (int w : {2, 4}) {
        run<0, 0>("plain", w, 0);
        run<1, 0>("setprio softmax", w, 0);
        for (int dl : {6, 10, 14, 20}) {
            run<0, 0>("stagger", w, dl);
            run<1, 0>("stagger + setprio softmax", w, dl);
            run<2, 0>("stagger + static prio odd groups", w, dl);
        }
        run<0, 1>("barrier per unit", w, 0);
        run<1, 1>("barrier per unit + setprio softmax", w, 0);
        run<1, 1>("barrier + stagger + setprio", w, 10);
    }
    return 0;
}
