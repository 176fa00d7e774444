// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LD_WAVE 64

// status codes shared with include/ld_mi355x.h
#define LD_OK 0
#define LD_ERR_ARG 1
#define LD_ERR_SHAPE 2
#define LD_ERR_HIP 3
#define LD_ERR_STATE 4

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ uint4 zero16() { return make_uint4(0u, 0u, 0u, 0u); }

union H8 {
    uint4 u;
    half8 h;
    half_t e[8];
};

__device__ __forceinline__ half8 as_half8(uint4 v) {
    H8 t;
    t.u = v;
    return t.h;
}

__device__ __forceinline__ void unpack8(uint4 v, float (&f)[8]) {
    H8 t;
    t.u = v;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)t.e[i];
}

// two floats -> packed fp16 pair, round to nearest even: ONE v_cvt_pk_f16_f32 (through the element-wise union hipcc emits two
// v_cvt_f16_f32 and a v_perm / v_pack per pair)
__device__ __forceinline__ unsigned pk2h(float a, float b) {
    const half2v h = {(half_t)a, (half_t)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) { return make_uint4(pk2h(f[0], f[1]), pk2h(f[2], f[3]), pk2h(f[4], f[5]), pk2h(f[6], f[7])); }

// eight fp16 sums, packed (v_pk_add_f16)
__device__ __forceinline__ uint4 add8h(uint4 a, uint4 b) {
    H8 x, y;
    x.u = a;
    y.u = b;
    x.h = x.h + y.h;
    return x.u;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, as torch.nn.functional.gelu default used by the reference's GEGLU (LD.py:4513-4515):
//   gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|),   Phi(-z) = 2^P(z) on z = min(|x|, 6)
// with P a degree-6 minimax fit of log2 Phi(-z) weighted for the absolute error of z Phi(-z) (tools/fit_gelu.py: max |error| 5.1e-7 over
// [-12, 12] evaluated in fp32, relative 5e-5 where |gelu| > 1e-3 — the Abramowitz-Stegun 7.1.26 erf this replaces measured 6.8e-7 / 2.2e-4;
// beyond z = 6 the true tail is < 6e-9).  ONE transcendental (v_exp_f32) and 10 single-issue fp32 operations per value instead of two
// transcendentals + 12 packed-fp32 operations per PAIR: a v_pk_*_f32 beside MFMAs costs ~3x two scalar operations (microarch guide,
// 'price of one filler beside MFMAs'), and the GEGLU epilogues are vector-issue bound (40 GELUs per lane and tile).
// NaN (accepted): min(|x|, 6) and max(x, 0) both return their non-NaN operand, so gelu(NaN) = -6 Phi(-6) ~ -6e-9 instead of
// NaN — torch's erf form propagates it.  In the GEGLU product a NaN VALUE half still propagates (the final multiply), a NaN that reaches only
// the GATE half is masked.  A NaN-transparent form (max(x, 0) as 0.5 (x + |x|)) costs one more vector instruction per value in an epilogue
// that is vector-issue bound; the activations of this path are finite by construction (fp16 storage would have turned an overflow into inf,
// which both forms handle alike: gelu(+inf) = +inf, gelu(-inf) = -0).
#define LD_GELU_ZMAX 6.0f
#define LD_GELU_C0 -0.999993086f
#define LD_GELU_C1 -1.15120173f
#define LD_GELU_C2 -0.458770961f
#define LD_GELU_C3 -0.0534121096f
#define LD_GELU_C4 0.00808071997f
#define LD_GELU_C5 -0.000769220525f
#define LD_GELU_C6 3.30928924e-05f
__device__ __forceinline__ float gelu_f(float x) {
    const float z = fminf(fabsf(x), LD_GELU_ZMAX);
    float q = fmaf(z, LD_GELU_C6, LD_GELU_C5);
    q = fmaf(q, z, LD_GELU_C4);
    q = fmaf(q, z, LD_GELU_C3);
    q = fmaf(q, z, LD_GELU_C2);
    q = fmaf(q, z, LD_GELU_C1);
    q = fmaf(q, z, LD_GELU_C0);
    return fmaf(-z, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));    // (q in [-30, -1]: never a denormal result)
}

// GEGLU products for 8 (value, gate) pairs: out = a * gelu(g) with the formula of gelu_f, written as volatile asm STAGE BY STAGE over the
// 8 values: 7 independent instructions between a result and its use (no dependency or transcendental-result bubbles; the hazard
// recogniser does not see inline asm).  hipcc schedules the 40 independent GELUs of a gate step chain by chain whatever the source order
// (and folds sched_barriers between pure operations): volatile asm keeps its order.  Issue cost per value: 10 x 4 + 8 (v_exp_f32) + 2
// (half a v_cvt_pk) = 50 cycles; round 4's packed form measured ~100 (profiles/README.md).
// g: the 8 gates (fp32); values: aw (4 packed fp16 pairs) or af (fp32 pairs); results: ow (4 packed fp16 pairs) or of (fp32 pairs).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool APACKED, bool OPACKED>
__device__ __forceinline__ void geglu8_staged_t(const unsigned (&aw)[4], const f32x2 (&af)[4], const f32x2 (&g)[4], unsigned (&ow)[4], f32x2 (&of)[4]) {
    const float kZ = LD_GELU_ZMAX, k6 = LD_GELU_C6, k4 = LD_GELU_C4, k3 = LD_GELU_C3, k2 = LD_GELU_C2, k1 = LD_GELU_C1, k0 = LD_GELU_C0;
    float k5 = LD_GELU_C5;
    asm volatile("" : "+v"(k5));                  // lives in a VGPR: a VOP3 instruction reads one scalar operand
    float gg[8], z[8], q[8], r[8], a[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gg[2 * k] = g[k].x;
        gg[2 * k + 1] = g[k].y;
    }
    asm volatile("s_nop 1");                       // (the inputs may come straight from producers the asm reads are invisible to)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_min_f32 %0, |%1|, %2" : "=v"(z[e]) : "v"(gg[e]), "s"(kZ));                 // z = min(|g|, 6)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q[e]) : "v"(z[e]), "s"(k6), "v"(k5));       // Horner in z
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[e]) : "v"(z[e]), "s"(k4));
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[e]) : "v"(z[e]), "s"(k3));
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[e]) : "v"(z[e]), "s"(k2));
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[e]) : "v"(z[e]), "s"(k1));
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[e]) : "v"(z[e]), "s"(k0));
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(q[e]));                                             // Phi(-z)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_max_f32 %0, 0, %1" : "=v"(r[e]) : "v"(gg[e]));                             // max(g, 0)
    if (APACKED) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {              // the value halves, to fp32
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[2 * k]) : "v"(aw[k]));
            asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[2 * k + 1]) : "v"(aw[k]));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[2 * k] = af[k].x;
            a[2 * k + 1] = af[k].y;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(r[e]) : "v"(z[e]), "v"(q[e]));             // gelu = max(g, 0) - z Phi(-z)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[e]) : "v"(a[e]));
    if (OPACKED) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ow[k]) : "v"(r[2 * k]), "v"(r[2 * k + 1]));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) of[k] = (f32x2){r[2 * k], r[2 * k + 1]};
    }
}
__device__ __forceinline__ void geglu8_staged(const unsigned (&aw)[4], const f32x2 (&g)[4], unsigned (&ow)[4]) {
    f32x2 dummy[4];
    geglu8_staged_t<true, true>(aw, dummy, g, ow, dummy);
}
__device__ __forceinline__ void geglu8_staged_f32(const f32x2 (&af)[4], const f32x2 (&g)[4], f32x2 (&of)[4]) {
    unsigned dummy[4];
    geglu8_staged_t<false, false>(dummy, af, g, dummy, of);
}

// quick_gelu of the CLIP MLP: a * sigmoid(1.702 a)  (ACTIVATIONS, LD.py:4296-4299)
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous run of logical tiles; neighbouring tiles then share operand panels in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// One LDS-DMA wave-instruction: lane l copies 16 bytes from its own global address to LDS byte (lds_base + 16*l).
// Issued through inline asm on purpose: hipcc's waitcnt pass does not see it, so it does not force vmcnt(0) before the
// fragment reads of OTHER ring stages (with the builtin it does, draining the ring every K-step).  The counted
// s_waitcnt vmcnt(N) + s_barrier in the loop are the only ordering (cdna guide §5.7: M0 is set and restored inside the
// same statement; lds_base must be wave-uniform).
__device__ __forceinline__ void glds16(const half_t* src, unsigned lds_base) {
    // M0 is written and consumed inside the statement; nothing else in these kernels reads M0 (gfx9+ DS instructions do
    // not), so it is not restored — two scalar instructions less per DMA in an issue-bound loop.
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(src), "s"(lds_base)
        : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const half_t* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#define HIP_CHECK_RET(expr)                      \
    do {                                         \
        hipError_t _e = (expr);                  \
        if (_e != hipSuccess) return LD_ERR_HIP; \
    } while (0)
