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

// In-wave software pipeline: the softmax of unit u shares ONE basic block with the 4 PV MFMAs of unit u-1 and the 3 QK^T MFMAs of unit u+1,
// interleaved 7 x (1 MFMA + NV vector instructions) by sched_group_barrier.  One wave per SIMD has the matrix pipe and the issue port to itself.
template <int NV>
__global__ __launch_bounds__(1024) void mixp(float* out, unsigned long long* clk, int iters, float seed) {
    f32x16 o0, o1, negm;
    for (int e = 0; e < 16; ++e) { o0[e] = 0; o1[e] = 0; negm[e] = -seed; }
    half8 k0, k1, k2, q0, q1, q2, v0, v1, v2, v3;
    for (int e = 0; e < 8; ++e) {
        k0[e] = (_Float16)(seed + e); k1[e] = (_Float16)(seed - e); k2[e] = (_Float16)seed;
        q0[e] = (_Float16)0.01f; q1[e] = (_Float16)0.02f; q2[e] = (_Float16)0.03f;
        v0[e] = (_Float16)1.f; v1[e] = (_Float16)2.f; v2[e] = (_Float16)3.f; v3[e] = (_Float16)4.f;
    }
    half8 pp0, pp1;
    for (int e = 0; e < 8; ++e) { pp0[e] = 0; pp1[e] = 0; }
    f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q0, negm, 0, 0, 0);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        // vector work of unit u (no rescale branch: the lazy reference makes it rare)
        float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int e = 3; e + 1 < 16; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);
        mx = fmaxf(mx, s[15]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        half8 pf0, pf1;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const float p0 = __builtin_amdgcn_exp2f(s[e] - mx * 1e-30f);
            const float p1 = __builtin_amdgcn_exp2f(s[e + 1]);
            const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(p0, p1));
            if (e < 8) { pf0[e] = h2[0]; pf0[e + 1] = h2[1]; } else { pf1[e - 8] = h2[0]; pf1[e - 7] = h2[1]; }
        }
        // matrix work: PV of unit u-1, QK^T of unit u+1
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pp0, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pp0, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v2, pp1, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v3, pp1, o1, 0, 0, 0);
        f32x16 sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q0, negm, 0, 0, 0);
        sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q1, sn, 0, 0, 0);
        sn = __builtin_amdgcn_mfma_f32_32x32x16_f16(k2, q2, sn, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);
        }
        asm volatile("" ::"v"(pf0), "v"(pf1));
        pp0 = pf0; pp1 = pf1; s = sn;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int e = 0; e < 16; ++e) acc += o0[e] + o1[e] + s[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

// Hand-placed version of the same pipeline: source order pinned by sched_barrier(0) between the seven MFMA gaps; two register sets (A / B)
// so that QK^T of unit u+1 never waits for the vector reads of unit u.  Per gap: 1 MFMA + ~30 issue cycles of vector work.
#define SB() __builtin_amdgcn_sched_barrier(0)
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define CVT(d, i, a, b) { const half2v h2_ = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a, b)); d[i] = h2_[0]; d[(i) + 1] = h2_[1]; }
#define UNIT(S, SN, PP0, PP1, PF0, PF1)                                                                               \
    {                                                                                                                 \
        float m0, m1, m2, m3, m4, e0, e1, e2, e3;                                                                      \
        asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));  /* (fresh fragments per unit) */ \
        SN = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q0, negm, 0, 0, 0);                                            \
        m0 = fmaxf(fmaxf(S[0], S[1]), S[2]); m1 = fmaxf(fmaxf(S[3], S[4]), S[5]); m2 = fmaxf(fmaxf(S[6], S[7]), S[8]);  \
        m3 = fmaxf(fmaxf(S[9], S[10]), S[11]); m4 = fmaxf(fmaxf(S[12], S[13]), S[14]);                                 \
        SB();                                                                                                         \
        SN = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q1, SN, 0, 0, 0);                                              \
        m0 = fmaxf(fmaxf(m0, m1), m2); m3 = fmaxf(fmaxf(m3, m4), S[15]); m0 = fmaxf(m0, m3);                           \
        { const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);    \
          mxacc = fmaxf(mxacc, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]))); }                               \
        SB();                                                                                                         \
        SN = __builtin_amdgcn_mfma_f32_32x32x16_f16(k2, q2, SN, 0, 0, 0);                                              \
        e0 = EXP2(S[0]); e1 = EXP2(S[1]); e2 = EXP2(S[2]); CVT(PF0, 0, e0, e1);                                         \
        SB();                                                                                                         \
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, PP0, o0, 0, 0, 0);                                             \
        e3 = EXP2(S[3]); e0 = EXP2(S[4]); e1 = EXP2(S[5]); CVT(PF0, 2, e2, e3); CVT(PF0, 4, e0, e1);                    \
        SB();                                                                                                         \
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, PP0, o1, 0, 0, 0);                                             \
        e0 = EXP2(S[6]); e1 = EXP2(S[7]); e2 = EXP2(S[8]); CVT(PF0, 6, e0, e1);                                         \
        SB();                                                                                                         \
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v2, PP1, o0, 0, 0, 0);                                             \
        e3 = EXP2(S[9]); e0 = EXP2(S[10]); e1 = EXP2(S[11]); CVT(PF1, 0, e2, e3); CVT(PF1, 2, e0, e1);                  \
        SB();                                                                                                         \
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v3, PP1, o1, 0, 0, 0);                                             \
        e0 = EXP2(S[12]); e1 = EXP2(S[13]); e2 = EXP2(S[14]); e3 = EXP2(S[15]); CVT(PF1, 4, e0, e1); CVT(PF1, 6, e2, e3); \
        SB();                                                                                                         \
    }

template <int THREADS>
__global__ __launch_bounds__(THREADS, (THREADS / 256 < 2 ? 2 : THREADS / 256)) void mixh(float* out, unsigned long long* clk, int iters, float seed) {
    f32x16 o0, o1, negm;
    for (int e = 0; e < 16; ++e) { o0[e] = 0; o1[e] = 0; negm[e] = -seed; }
    half8 k0, k1, k2, q0, q1, q2, v0, v1, v2, v3;
    for (int e = 0; e < 8; ++e) {
        k0[e] = (_Float16)(seed + e); k1[e] = (_Float16)(seed - e); k2[e] = (_Float16)seed;
        q0[e] = (_Float16)0.01f; q1[e] = (_Float16)0.02f; q2[e] = (_Float16)0.03f;
        v0[e] = (_Float16)1.f; v1[e] = (_Float16)2.f; v2[e] = (_Float16)3.f; v3[e] = (_Float16)4.f;
    }
    half8 pa0, pa1, pb0, pb1;
    for (int e = 0; e < 8; ++e) { pa0[e] = 0; pa1[e] = 0; pb0[e] = 0; pb1[e] = 0; }
    float mxacc = 0.f;
    f32x16 sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q0, negm, 0, 0, 0), sb;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
        UNIT(sa, sb, pb0, pb1, pa0, pa1)     // unit u: reads sa, PV of u-1 (pb), writes pa, QK^T of u+1 -> sb
        UNIT(sb, sa, pa0, pa1, pb0, pb1)
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = mxacc;
    for (int e = 0; e < 16; ++e) acc += o0[e] + o1[e] + sa[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int WPS>
void runh() {
    const int wps = WPS;
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4 * 2);
    unsigned long long* c; (void)hipMalloc(&c, 64);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mixh<256 * WPS><<<256, 256 * wps>>>(d, c, iters, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mixh<256 * WPS><<<256, 256 * wps>>>(d, c, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    const double ns = ms * 1e6 / ((double)iters * wps);
    printf("hand-placed in-wave pipeline (7 gaps)            w/SIMD=%d: %7.3f ms %6.1f ns/unit/SIMD = %5.0f cycles at %.2f GHz\n", wps, ms, ns, ns * ghz, ghz);
    (void)hipFree(d); (void)hipFree(c);
}

template <int NV>
void runp(int wps) {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4 * 2);
    unsigned long long* c; (void)hipMalloc(&c, 64);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mixp<NV><<<256, 256 * wps>>>(d, c, iters, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mixp<NV><<<256, 256 * wps>>>(d, c, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    const double ns = ms * 1e6 / ((double)iters * wps);
    printf("in-wave pipeline, %d vector instr per MFMA gap  w/SIMD=%d: %7.3f ms %6.1f ns/unit/SIMD = %5.0f cycles at %.2f GHz\n", NV, wps, ms, ns, ns * ghz, ghz);
    (void)hipFree(d); (void)hipFree(c);
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
    runh<1>(); runh<2>(); runh<4>();
    for (int w : {1, 2, 4}) {
        runp<5>(w);
        runp<6>(w);
        runp<8>(w);
    }
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
