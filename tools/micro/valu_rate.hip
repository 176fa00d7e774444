// Issue-rate probe for the softmax inner loop on gfx950: cycles per wave-instruction of v_fma_f32 / v_exp_f32 /
// v_cvt_pkrtz / v_max3 / v_pk_fma_f32, alone and with N waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(float* out, int iters, float seed) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-6f + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (OP == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
            if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(seed));
            if (OP == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
            if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
            if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
            if (OP == 8) { if (i & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])); else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed)); }
            if (OP == 9) { if ((i & 3) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])); else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed)); }
            if (OP == 10) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
            if (OP == 11) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));
            if (OP == 12) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));
            if (OP == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
        }
        if (OP == 4) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2 t = {v[i], v[i + 1]};
                f2 s2 = {seed, seed};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(t) : "v"(s2));
                v[i] = t[0]; v[i + 1] = t[1];
            }
        }
    }
    long long t1 = clock64();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)(t1 - t0);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int OP>
void run(const char* name, int waves_per_simd, int n_per_iter) {
    float* d;
    hipMalloc(&d, 1 << 22);
    const int iters = 32768;
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;   // 4 SIMDs per CU
    const int blocks_per_cu = (256 * waves_per_simd + threads - 1) / threads;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<OP><<<256 * blocks_per_cu, threads>>>(d, iters, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP><<<256 * blocks_per_cu, threads>>>(d, iters, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float clk; hipMemcpy(&clk, d, 4, hipMemcpyDeviceToHost);
    const double n_inst = (double)iters * n_per_iter;
    printf("%-16s waves/SIMD=%d: %7.2f clock64-ticks per wave-instr (wave 0), wall %8.3f ms -> %6.2f ns per instr per SIMD-wave-slot\n", name, waves_per_simd,
           clk / n_inst, ms, ms * 1e6 / (n_inst * waves_per_simd));
    hipFree(d);
}

int main() {
    for (int w : {1, 4, 8}) {
        run<0>("v_fma_f32", w, 16);
        run<6>("v_mul_f32", w, 16);
        run<7>("v_add_u32", w, 16);
        run<1>("v_exp_f32", w, 16);
        run<5>("v_exp_f16", w, 16);
        run<2>("v_cvt_pkrtz", w, 16);
        run<3>("v_max3_f32", w, 16);
        run<4>("v_pk_fma_f32", w, 8);
        run<8>("exp:fma 1:1", w, 16);
        run<9>("exp:fma 1:3", w, 16);
        run<10>("v_cvt_pk_f16_f32", w, 16);
        run<11>("v_cvt_f16_f32", w, 16);
        run<13>("v_max_f32", w, 16);
    }
    return 0;
}
