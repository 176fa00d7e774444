// ds_read_b128 throughput for the lane -> address patterns of the kernels (one wave per SIMD, 4 waves per CU busy): which patterns
// the LDS serves at full rate.  Prints ns per wave-instruction; the linear pattern (lane * 16) is the conflict-free reference.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void probe(const int* offs, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int o0 = offs[lane], o1 = offs[64 + lane], o2 = offs[128 + lane], o3 = offs[192 + lane];
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        const uint4 a = *reinterpret_cast<const uint4*>(smem + o0), b = *reinterpret_cast<const uint4*>(smem + o1);
        const uint4 c = *reinterpret_cast<const uint4*>(smem + o2), d = *reinterpret_cast<const uint4*>(smem + o3);
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
        asm volatile("" : "+v"(acc.x));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc.x;
}
static void run(const char* name, int (*f)(int lane, int k)) {
    int h[256];
    for (int k = 0; k < 4; ++k) for (int l = 0; l < 64; ++l) h[k * 64 + l] = f(l, k);
    int* d; float* o;
    (void)hipMalloc(&d, sizeof h); (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipMalloc(&o, 256 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<<<256, 256>>>(d, o, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); probe<<<256, 256>>>(d, o, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %6.2f ns per ds_read_b128 per wave (4 waves per CU)\n", name, ms * 1e6 / (iters * 4.0));
    (void)hipFree(d); (void)hipFree(o);
}
int main() {
    run("linear lane*16", [](int l, int k) { return l * 16 + k * 1024; });
    run("GEMM frag: row=l&15 stride 128, chunk (k*... + l>>4) ^ (row&7)", [](int l, int k) { int r = l & 15, c = ((k & 1) * 4 + (l >> 4)) ^ (r & 7); return (r + 16 * (k >> 1)) * 128 + c * 16; });
    run("attn V: row=l&31 stride 128, chunk (2k + l>>5) ^ (row&7)", [](int l, int k) { int r = l & 31, c = (2 * k + (l >> 5)) ^ (r & 7); return r * 128 + c * 16; });
    run("attn V alt: key (row ^ row>>3) & 7", [](int l, int k) { int r = l & 31, c = (2 * k + (l >> 5)) ^ ((r ^ (r >> 3)) & 7); return r * 128 + c * 16; });
    run("attn V alt: key (row>>2) & 7", [](int l, int k) { int r = l & 31, c = (2 * k + (l >> 5)) ^ ((r >> 2) & 7); return r * 128 + c * 16; });
    run("attn V alt: key ((row&3)<<1 | (row>>2&1))", [](int l, int k) { int r = l & 31, key = ((r & 3) << 1) | ((r >> 2) & 1), c = (2 * k + (l >> 5)) ^ key; return r * 128 + c * 16; });
    run("attn V alt: half-wave in chunk low bit: (k<<1|hh)^key, key=row&7, rows 2 per 256B", [](int l, int k) { int r = l & 31, c = (2 * k + (l >> 5)) ^ (r & 7); return (r >> 1) * 256 + (r & 1) * 128 + c * 16; });
    run("attn K d=40: row=l&31 stride 80 + hh*16 + k*32", [](int l, int k) { return (l & 31) * 80 + (l >> 5) * 16 + (k % 3) * 32; });
    run("attn K d=80: stride 160, chunk (2k+hh)^((row>>2)&1)", [](int l, int k) { int r = l & 31, c = (2 * k + (l >> 5)) ^ ((r >> 2) & 1); return r * 160 + c * 16; });
    run("attn K d=80 alt: key (row>>1)&... 2 bits: (row>>2)&3 on chunk pairs", [](int l, int k) { int r = l & 31, c = (2 * k + (l >> 5)) ^ (((r >> 2) & 1)) ^ (((r >> 3) & 1) << 1); return r * 160 + c * 16; });
    run("attn K d=40 OLD: row=perm(l&31) (bits 2,3 swapped), stride 80", [](int l, int k) { int r = l & 31; r = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); return r * 80 + (l >> 5) * 16 + (k % 3) * 32; });
    run("2-way conflict reference: row=l&31 stride 256, same chunk", [](int l, int k) { return (l & 31) * 256 + (l >> 5) * 16 + k * 32; });
    return 0;
}
