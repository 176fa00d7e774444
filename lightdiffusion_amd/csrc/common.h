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
// exact (erf) GELU, as torch.nn.functional.gelu default used by the reference's GEGLU (LD.py:4513-4515)
__device__ __forceinline__ float gelu_f(float x) {
    // erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 output step): one v_rcp + one v_exp + 6 FMA
    // instead of libm erff's branchy ~30 instructions — the GEGLU epilogue runs this on 84M elements per level-0 block.
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// GEGLU products for 8 (value, gate) pairs: ow[k] = fp16x2( a * gelu(g) ) with the erf of gelu_f (Abramowitz-Stegun 7.1.26), written as
// volatile asm STAGE BY STAGE over 4 register pairs: 4..8 independent instructions between a result and its use (no dependency or
// transcendental-result bubbles), packed fp32 where the ISA has it, 12 instructions per product.  hipcc schedules the 40 independent GELUs
// of a gate step chain by chain whatever the source order (and folds sched_barriers between pure operations): volatile asm keeps its order.
// Measured in the row-panel GEMM's gate step: 5400 -> 5000 clocks.  What is left is the instruction mix itself — about 100 issue cycles per
// product (2 transcendentals at quarter rate, 6 packed-fp32 operations at half rate) — not contention with the partner wave's MFMA stream:
// idling that stream with s_nops leaves the 5000 clocks unchanged (profiles/README.md).
// g: the 8 gates (fp32); values: aw (4 packed fp16 pairs) or af (fp32 pairs); results: ow (4 packed fp16 pairs) or of (fp32 pairs).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool APACKED, bool OPACKED>
__device__ __forceinline__ void geglu8_staged_t(const unsigned (&aw)[4], const f32x2 (&af)[4], const f32x2 (&g)[4], unsigned (&ow)[4], f32x2 (&of)[4]) {
    const f32x2 kP = {0.3275911f, 0.3275911f}, kL = {1.44269504088896340736f, 1.44269504088896340736f};
    const f32x2 kA5 = {1.061405429f, 1.061405429f}, kA3 = {1.421413741f, 1.421413741f}, kA2 = {-0.284496736f, -0.284496736f},
                kA1 = {0.254829592f, 0.254829592f};
    f32x2 kA4 = {-1.453152027f, -1.453152027f};
    asm volatile("" : "+v"(kA4));                 // lives in VGPRs: a VOP3P instruction takes one scalar operand
    const float kC = 0.70710678118654752440f;
    const unsigned kM = 0x7fffffffu;
    f32x2 z[4], t[4], e[4], q[4], a[4];
    asm volatile("s_nop 1");                       // (the inputs may come straight from packed-fp32 producers the asm reads are invisible to)
#pragma unroll
    for (int k = 0; k < 4; ++k) {                  // z = |g| / sqrt(2)
        asm volatile("v_mul_f32 %0, |%1|, %2" : "=v"(z[k].x) : "v"(g[k].x), "s"(kC));
        asm volatile("v_mul_f32 %0, |%1|, %2" : "=v"(z[k].y) : "v"(g[k].y), "s"(kC));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel_hi:[1,1,0]" : "=v"(t[k]) : "v"(z[k]), "s"(kP));   // 1 + p z
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_mul_f32 %0, %1, %1 neg_lo:[1,0] neg_hi:[1,0]" : "=v"(e[k]) : "v"(z[k]));         // -z^2
#pragma unroll
    for (int k = 0; k < 4; ++k) {                  // t = 1 / (1 + p z)
        asm volatile("v_rcp_f32 %0, %0" : "+v"(t[k].x));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(t[k].y));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(e[k]) : "s"(kL));
#pragma unroll
    for (int k = 0; k < 4; ++k) {                  // e = exp(-z^2)
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[k].x));
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[k].y));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(q[k]) : "v"(t[k]), "s"(kA5), "v"(kA4));           // Horner in t
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[k]) : "v"(t[k]), "s"(kA3));
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[k]) : "v"(t[k]), "s"(kA2));
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[k]) : "v"(t[k]), "s"(kA1));
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[k]) : "v"(t[k]));
#pragma unroll
    for (int k = 0; k < 4; ++k)                    // erf(|z|) = 1 - poly * e
        asm volatile("v_pk_fma_f32 %0, %0, %1, 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(q[k]) : "v"(e[k]));
    if (APACKED) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {              // the value halves, to fp32
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[k].x) : "v"(aw[k]));
            asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[k].y) : "v"(aw[k]));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = af[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                  // copysign(erf, g)
        asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(q[k].x) : "s"(kM), "v"(g[k].x));
        asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(q[k].y) : "s"(kM), "v"(g[k].y));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_mul_f32 %0, %1, 0.5 op_sel_hi:[1,0]" : "=v"(t[k]) : "v"(g[k]));                   // g / 2
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(q[k]) : "v"(t[k]));                                // gelu = g/2 * erf + g/2
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[k]) : "v"(a[k]));
    if (OPACKED) {
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ow[k]) : "v"(q[k].x), "v"(q[k].y));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) of[k] = q[k];
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
