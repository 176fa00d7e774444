// Emulates the register-level instruction mix of one attention subtile (32 keys x 32 queries per wave, d=40):
// 3 dependent QK^T MFMAs -> max tree -> 16 v_exp -> 8 cvt_pk -> 4 PV MFMAs, W waves per SIMD, no memory traffic.
// Variants probe how the mix should be ordered / prioritised.  cycles/subtile/SIMD printed from wall time at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(1024) void mix(float* out, int iters, float seed) {
    f32x16 o0, o1, negm;
    for (int e = 0; e < 16; ++e) { o0[e] = 0; o1[e] = 0; negm[e] = -seed; }
    half8 k0, k1, k2, q0, q1, q2, v0, v1, v2, v3;
    for (int e = 0; e < 8; ++e) {
        k0[e] = (_Float16)(seed + e); k1[e] = (_Float16)(seed - e); k2[e] = (_Float16)seed;
        q0[e] = (_Float16)0.01f; q1[e] = (_Float16)0.02f; q2[e] = (_Float16)0.03f;
        v0[e] = (_Float16)1.f; v1[e] = (_Float16)2.f; v2[e] = (_Float16)3.f; v3[e] = (_Float16)4.f;
    }
    float thr = 1e30f * seed;
    half8 pprev0, pprev1;
    for (int e = 0; e < 8; ++e) { pprev0[e] = 0; pprev1[e] = 0; }
    for (int it = 0; it < iters; ++it) {
        f32x16 s;
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s) : "v"(k0), "v"(q0), "v"(negm));
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q1, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k2, q2, s, 0, 0, 0);
        if (VAR == 2) {   // software pipelining: PV of the PREVIOUS subtile issues here, under this subtile's softmax
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pprev0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pprev0, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v2, pprev1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v3, pprev1, o1, 0, 0, 0);
        }
        if (VAR == 1 || VAR == 3) __builtin_amdgcn_s_setprio(1);
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
        if (VAR == 1 || VAR == 3) __builtin_amdgcn_s_setprio(0);
        if (VAR != 2) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf0, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v2, pf1, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v3, pf1, o1, 0, 0, 0);
        } else {
            pprev0 = pf0; pprev1 = pf1;
        }
        if (VAR == 3) { asm volatile("" : "+v"(o0), "+v"(o1)); }
    }
    float acc = 0;
    for (int e = 0; e < 16; ++e) acc += o0[e] + o1[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int VAR>
void run(const char* name, int waves_per_simd) {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4 * 2);
    const int iters = 20000;
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    const int blocks = 256 * ((256 * waves_per_simd + threads - 1) / threads);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mix<VAR><<<blocks, threads>>>(d, iters, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mix<VAR><<<blocks, threads>>>(d, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD=%d: %8.3f ms  -> %7.1f ns per subtile per SIMD (MFMA floor 7*32 cyc)\n", name, waves_per_simd, ms,
           ms * 1e6 / ((double)iters * waves_per_simd));
    (void)hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("plain", w);
        run<1>("setprio(1) around softmax", w);
        run<2>("PV of previous under softmax", w);
    }
    return 0;
}
