// How fast can one CU pull global memory into LDS with LDS-DMA (global_load_lds_dwordx4)?  Working set is L2-resident.
// Variants: row-strided 128-byte rows (8 lanes per row, as the GEMM A/W tiles) vs fully contiguous 1 KB per instruction,
// per-lane 64-bit addresses vs scalar base + 32-bit offsets, 1 or 2 workgroups per CU, and plain global_load_dwordx4 -> VGPR.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ void glds16(const void* src, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void glds16s(unsigned voff, const void* sbase, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

template <int MODE, int WRAP = 8>   // WRAP: slab positions cycled through (8: cache-resident; 176: the whole 23040-byte row, streamed)
// 0: strided rows, per-lane ptr; 1: contiguous, per-lane ptr; 2: strided rows, saddr; 3: plain loads to VGPR (strided); 4: glds dword (4 B/lane) contiguous
__global__ __launch_bounds__(256, 2) void pull(const char* src, size_t row_stride, size_t span, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) char smem[2][36864];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)&smem[0][0];
    // 9 instructions per wave per "slab" (36 KB per workgroup), like the 128x160x64 GEMM tile
    const char* p[9];
    unsigned off[9];
    const char* blk = src + (size_t)(blockIdx.x % 64) * span;   // 64 distinct streams (span bytes each, sized by the host for the stride)
    for (int i = 0; i < 9; ++i) {
        const int q = tid + i * 256;
        const size_t o = (MODE == 1 || MODE == 4) ? (size_t)q * 16 : (size_t)(q >> 3) * row_stride + (q & 7) * 16;
        p[i] = blk + o;
        off[i] = (unsigned)o;
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
    const char* sb = blk;
    for (int it = 0; it < iters; ++it) {
        const unsigned st = base + (it & 1) * 36864 + wid * 1024;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (MODE == 0 || MODE == 1) glds16(p[i] + (size_t)(it % WRAP) * 128, st + i * 4096);
            if (MODE == 2) glds16s(off[i], sb, st + i * 4096);
            if (MODE == 3) {
                const uint4 v = *reinterpret_cast<const uint4*>(p[i] + (size_t)(it % WRAP) * 128);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
        }
        if (MODE == 2) sb = blk + (size_t)((it + 1) % WRAP) * 128;
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // one slab in flight behind the one being issued
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + tid] = (float)smem[0][tid] + (float)(acc.x ^ acc.y ^ acc.z ^ acc.w);
}

template <int MODE, int WRAP = 8>
void run(const char* name, int blocks_per_cu, size_t row_stride) {
    char* src; float* out;
    const size_t span = 288 * (row_stride ? row_stride : 128) + 8192 + (size_t)WRAP * 128;   // rows 0..287 + the 8 x 128 B slab offsets + slack
    const size_t total = 64 * span + (1 << 20);
    if (hipMalloc(&src, total) != hipSuccess) { printf("alloc failed\n"); return; }
    (void)hipMemset(src, 1, total);
    (void)hipMalloc(&out, 512 * 256 * 4);
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    pull<MODE, WRAP><<<blocks, 256>>>(src, row_stride, span, iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    pull<MODE, WRAP><<<blocks, 256>>>(src, row_stride, span, iters, out);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * 36864.0;
    printf("%-44s wg/CU=%d: %7.3f ms  %7.2f TB/s chip  %6.1f GB/s per CU\n", name, blocks_per_cu, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    (void)hipFree(src); (void)hipFree(out);
}

int main() {
    for (int b : {1, 2}) {
        run<0>("LDS-DMA, 128 B rows (stride 2560 B), vaddr", b, 2560);
        run<0>("LDS-DMA, 128 B rows (stride 23040 B), vaddr", b, 23040);
        run<1>("LDS-DMA, contiguous 1 KB / instr, vaddr", b, 0);
        run<2>("LDS-DMA, 128 B rows (stride 2560 B), saddr", b, 2560);
        run<3>("global_load_dwordx4 -> VGPR, 128 B rows", b, 2560);
        run<0, 176>("LDS-DMA, 23040 B rows STREAMED (425 MB set)", b, 23040);
        run<2, 176>("LDS-DMA saddr, 23040 B rows STREAMED", b, 23040);
    }
    return 0;
}
