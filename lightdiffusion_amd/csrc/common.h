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

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    H8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.e[i] = (half_t)f[i];
    return t.u;
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
