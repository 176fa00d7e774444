// conv8: row-resident 3x3 convolution for the small-M levels of a batch-1 step (UNet batch 2: the 16x16 and 8x8 levels and the
// middle block — ResBlock1 convolutions, LD.py:5189-5287, and Upsample1's convolution, LD.py:5141-5152).
//
// Why a separate kernel: at M = 128 / 512 rows the 64 x 160 tiles of the general kernel re-read every weight byte once per M tile and
// every activation byte nine times, ~10x the unique bytes of the launch, through a per-CU LDS-DMA path that takes in ~30 GB/s from beyond
// L2 (profiles/README.md round 4) — 35..40 us for 29.5 MB of weights.  Here
//   * one workgroup owns ALL M rows x 80 output channels x a slab of the input channels (all nine taps): every weight byte enters exactly
//     one CU, once (LDS-DMA ring, three 25 KB stages in flight), and the grid is (N / 80) x S = 256 workgroups, one per CU;
//   * the activations of a 16-channel sub-slab sit in LDS as a zero-bordered halo image (IMGS x (W+2)^2 pixels x 32 bytes), staged
//     through registers by waves 4-7 — which is where the GroupNorm + SiLU of the input is applied (scale / shift finished in the
//     prologue from the producer's partial statistics): no gn_apply launch, no normalised tensor in HBM; waves 0-3 issue the weight DMA;
//   * a k-step of the 16x16x32 MFMA = two taps x 16 channels (the tenth "tap" is a zero weight chunk), so a lane's tap offset is one of
//     five precomputed values and every fragment read is base + immediate;
//   * the S channel-slab partial sums of an N tile meet in HBM (fp32, written through with sc1 stores) and are reduced INSIDE the launch:
//     arrive counter -> (bounded wait) -> the row parts of the tile are claimed one by one, summed in slab order (bitwise reproducible),
//     finished with bias / time-embedding row / residual, stored as fp16 — and the GroupNorm partial statistics of the OUTPUT are
//     emitted per (image, row part, group), so the next GroupNorm needs no statistics launch either.  A workgroup that cannot wait
//     (time-out: its peers are not resident) leaves; the last arriver always finds every slab complete and takes whatever is left, so
//     the protocol terminates and is correct under any dispatch order or placement (cdna guide, Guideline 16: sc1 payload stores, every
//     storing wave drains vmcnt, one lane signals; the readers acquire once and load with sc1).
// One launch replaces gn_apply + conv + split-K reduce (+ gn_stats of the next norm).
#include "gemm.h"

namespace {

constexpr int C8_BN = 80;                 // output channels per workgroup
constexpr int C8_THREADS = 512;
constexpr int C8_RING = 4;                // weight ring stages (one 16-channel sub-slab each)
constexpr int C8_WSTAGE = 5 * C8_BN * 64; // 5 k-steps x 80 rows x 64 B = 25 wave-instructions of 1 KB
constexpr int C8_MAXCW = 160;             // channels of one workgroup's slab (scale / shift table)

__device__ uint4 g_c8_zero[8];            // 128 zero bytes: the tenth "tap" of a k-step pair

__device__ __forceinline__ int c8_g(int x) { return (0x78 >> (2 * (x & 3))) & 3; }   // {0, 2, 3, 1}: 64-byte-row swizzle of the weight stage (gemm5's)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st16_sc1(float* p, f32x4 v) {   // write-through 16-byte store (agent scope)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 ld16_sc1(const float* p) {      // L1-bypassing 16-byte load, result usable after the caller's vmcnt wait
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int WD, int IMGS, bool GN, bool UP>
__global__ __launch_bounds__(C8_THREADS, 2) void conv8_kernel(const GemmParams p) {
    constexpr int HP = WD + 2, HPP = HP * HP, HPIX = IMGS * HPP;
    constexpr int MT = IMGS * WD * WD, RT = MT / 16, TM = RT / 8;
    constexpr int HALO_B = HPIX * 32;
    constexpr int NCH = (HPIX * 2 + 255) / 256;            // halo chunks per thread of the four staging waves
    constexpr int NP = MT >= 512 ? 16 : 8;                  // row parts of a tile (claimed one by one in the reduction)
    constexpr int RTPP = RT / NP;                           // 16-row tiles per part
    constexpr int ITEMS = RTPP * 5 * 64;                    // float4 items per part
    static_assert(RT % 8 == 0 && RT % NP == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem8[];
    char* const wring = smem8;
    char* const halo = smem8 + C8_RING * C8_WSTAGE;
    float* const gsc = reinterpret_cast<float*>(halo + 2 * HALO_B);   // [IMGS][160] scale, then [IMGS][160] shift
    float* const gsh = gsc + IMGS * C8_MAXCW;
    float* const gmr = gsh + IMGS * C8_MAXCW;                           // [IMGS][16][2] mean / rstd of the slab's groups
    int* const flags = reinterpret_cast<int*>(gmr + IMGS * 32);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int NTN = p.N / C8_BN, S = p.c8_S;
    const int j = blockIdx.x % NTN, s = blockIdx.x / NTN;
    const int Cin = p.C1 + p.C2;
    const int nsub = Cin >> 4;
    const int sb = (int)((long long)s * nsub / S), se = (int)((long long)(s + 1) * nsub / S);
    const int nloc = se - sb;

    // ------------------------------------------------------------------------------------------ loader state (waves 0-3)
    const half_t* wsrc[7];
    unsigned wvalid = 0;
    const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)wring);
    if (wid < 4) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int q = wid + 4 * u;
            const int c = 64 * (q < 25 ? q : 24) + lane;
            const int t = c / 320, rem = c - 320 * t;
            const int n = rem >> 2, pc = rem & 3;
            const int fql = pc ^ c8_g(n >> 2);
            const int tap = 2 * t + (fql >> 1), cc = fql & 1;
            const bool ok = tap < 9;
            wsrc[u] = ok ? p.W + (long long)(j * C8_BN + n) * p.ldw + (long long)tap * Cin + sb * 16 + cc * 8 : reinterpret_cast<const half_t*>(g_c8_zero);
            wvalid |= ok ? (1u << u) : 0u;
        }
    }
    auto w_issue = [&](int slot) {       // one 16-channel sub-slab of the weights: 25 one-KB pieces, wave w takes pieces w, w+4, ...
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            if (u == 6 && wid != 0) break;
            glds16(wsrc[u], ring_base + (unsigned)(slot * C8_WSTAGE + (wid + 4 * u) * 1024));
            wsrc[u] += (wvalid >> u) & 1 ? 16 : 0;
        }
    };
    auto w_wait = [&](int ahead) {       // this wave's pieces of a stage have landed: all but the pieces of the `ahead` stages issued after it
        if (wid == 0) {
            if (ahead >= 2) wait_vmcnt<14>();
            else if (ahead == 1) wait_vmcnt<7>();
            else wait_vmcnt<0>();
        } else {
            if (ahead >= 2) wait_vmcnt<12>();
            else if (ahead == 1) wait_vmcnt<6>();
            else wait_vmcnt<0>();
        }
    };

    // ------------------------------------------------------------------------------------------ halo staging state (waves 4-7)
    const int t4 = tid - 256;
    int hpl[NCH];          // source pixel (linear index into the NHWC source, in pixels), -1 = border / beyond the image
    int himg[NCH];
    uint4 hreg[NCH];
    if (wid >= 4) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int c = t4 + 256 * u;
            const int pix = c >> 1;
            const int img = pix / HPP, r2 = pix - img * HPP;
            const int hy = r2 / HP, hx = r2 - hy * HP;
            const bool in = c < HPIX * 2 && hy >= 1 && hy <= WD && hx >= 1 && hx <= WD;
            const int sy = UP ? (hy - 1) >> 1 : hy - 1, sx = UP ? (hx - 1) >> 1 : hx - 1;
            himg[u] = img;
            hpl[u] = in ? (img * p.Hs + sy) * p.Ws + sx : -1;
        }
    }
    auto halo_load = [&](int ss) {
        const int c0 = ss * 16;
        const bool second = c0 >= p.C1;
        const half_t* src = second ? p.A2 : p.A;
        const int Cs = second ? p.C2 : p.C1, cl = second ? c0 - p.C1 : c0;
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int cc = (t4 + 256 * u) & 1;
            hreg[u] = hpl[u] >= 0 ? ld16(src + (long long)hpl[u] * Cs + cl + cc * 8) : zero16();
        }
    };
    auto halo_store = [&](int k, int buf) {   // k: sub-slab index inside this workgroup's slab (scale / shift table offset)
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int c = t4 + 256 * u;
            if (c >= HPIX * 2) continue;
            uint4 v = hreg[u];
            if (GN && hpl[u] >= 0) {
                const int cc = c & 1;
                const float* sc = gsc + himg[u] * C8_MAXCW + k * 16 + cc * 8;
                const float* sh = gsh + himg[u] * C8_MAXCW + k * 16 + cc * 8;
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[e] = f[e] * sc[e] + sh[e];
                    if (p.gn_silu) f[e] = silu_f(f[e]);
                }
                v = pack8(f);
            }
            *reinterpret_cast<uint4*>(halo + buf * HALO_B + c * 16) = v;
        }
    };

    // ------------------------------------------------------------------------------------------ prologue
    if (wid < 4) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < nloc) w_issue(k);
    } else {
        halo_load(sb);
        if (GN) {   // mean / rstd of the groups this slab touches, from the producer's partial statistics (fixed summation order)
            const int cpg = Cin / 32;
            const int g0 = (sb * 16) / cpg, g1 = (se * 16 - 1) / cpg, ng = g1 - g0 + 1;
            if (t4 < IMGS * ng) {
                const int img = t4 / ng, g = g0 + t4 - img * ng;
                const float* pp = p.gn_in_part + ((long long)img * p.gn_in_P * 32 + g) * 2;
                float a = 0.f, b = 0.f;
                for (int i = 0; i < p.gn_in_P; ++i) {
                    a += pp[(long long)i * 64];
                    b += pp[(long long)i * 64 + 1];
                }
                const float cnt = (float)cpg * (float)(p.Hs * p.Ws);
                const float mu = a / cnt;
                const float var = fmaxf(b / cnt - mu * mu, 0.f);
                gmr[(img * 16 + (g - g0)) * 2] = mu;
                gmr[(img * 16 + (g - g0)) * 2 + 1] = rsqrtf(var + p.gn_eps);
            }
        }
    }
    if (GN) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS writes above
        __builtin_amdgcn_s_barrier();
        if (wid >= 4) {
            const int cpg = Cin / 32;
            const int g0 = (sb * 16) / cpg;
            const int cw = nloc * 16;
            for (int e = t4; e < IMGS * cw; e += 256) {
                const int img = e / cw, c = e - img * cw;
                const int cg = sb * 16 + c, g = cg / cpg - g0;
                const float mu = gmr[(img * 16 + g) * 2], rs = gmr[(img * 16 + g) * 2 + 1];
                const float sc = rs * (float)p.gn_gamma[cg];
                gsc[img * C8_MAXCW + c] = sc;
                gsh[img * C8_MAXCW + c] = (float)p.gn_beta[cg] - mu * sc;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (wid >= 4) halo_store(0, 0);

    // ------------------------------------------------------------------------------------------ fragment addressing
    const int rt0 = wid * TM;
    int pix0;
    if (WD == 16) {
        const int img = rt0 >> 4, y = rt0 & 15;                 // a 16-row tile = one image row
        pix0 = img * HPP + y * HP + fr;
    } else {
        const int img = rt0 >> 2, y0 = (rt0 & 3) * 2;           // W = 8: a 16-row tile = two image rows
        pix0 = img * HPP + (y0 + (fr >> 3)) * HP + (fr & 7);
    }
    const int a_lane = pix0 * 32 + (fq & 1) * 16;
    int toff[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        int tap = 2 * t + (fq >> 1);
        tap = tap > 8 ? 8 : tap;                                // (the tenth tap's weights are zero: any finite activation will do)
        toff[t] = ((tap / 3) * HP + (tap % 3)) * 32;
    }
    const int b_lane = fr * 64 + ((fq ^ c8_g(fr >> 2)) << 4);

    f32x4 acc[TM][5];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ------------------------------------------------------------------------------------------ main loop: one 16-channel sub-slab per barrier
    for (int k = 0; k < nloc; ++k) {
        if (wid < 4) w_wait(nloc - 1 - k);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's halo writes of sub-slab k
        __builtin_amdgcn_s_barrier();
        if (wid < 4) {
            if (k + 3 < nloc) w_issue((k + 3) & 3);
        } else if (k + 1 < nloc) {
            halo_load(sb + k + 1);
        }
        const char* wst = wring + (k & 3) * C8_WSTAGE + b_lane;
        const char* hb = halo + (k & 1) * HALO_B + a_lane;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            half8 fa[TM], fb[5];
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) fb[jj] = as_half8(*reinterpret_cast<const uint4*>(wst + t * 5120 + jj * 1024));
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = as_half8(*reinterpret_cast<const uint4*>(hb + toff[t] + i * (HP * 32)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jj], fa[i], acc[i][jj], 0, 0, 0);
        }
        if (wid >= 4 && k + 1 < nloc) halo_store(k + 1, (k + 1) & 1);
    }

    // ------------------------------------------------------------------------------------------ partial sums -> HBM (write-through)
    float* const slab0 = p.partial + (long long)j * (RT * 5 * 256);
    const long long slab_stride = (long long)NTN * (RT * 5 * 256);
    {
        float* mine = slab0 + (long long)s * slab_stride;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) st16_sc1(mine + (((rt0 + i) * 5 + jj) * 64 + lane) * 4, acc[i][jj]);
    }
    wait_vmcnt<0>();                                            // every storing wave drains its stores ...
    __syncthreads();                                            // ... before ONE lane signals for the workgroup
    int* const cnt = p.sync + j * 4;                            // [0] arrivals, [1] claims, [2] leavers
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        if (t != S - 1) {                                       // not the last arriver: wait (bounded) until every slab of the tile is there
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S) {
                __builtin_amdgcn_s_sleep(8);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) {   // 2 ms at 100 MHz: peers not resident — leave, the last arriver finishes
                    ok = 0;
                    break;
                }
            }
        }
        if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flags[0] = ok;
    }
    __syncthreads();
    const bool take = flags[0] != 0;

    const int hw = WD * WD;
    const int cpg_o = p.N / 32;
    float2* const scratch = reinterpret_cast<float2*>(wring);   // the ring is quiet now
    while (take) {
        __syncthreads();                                        // (flags / scratch of the previous part are consumed)
        if (tid == 0) flags[1] = __hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int r = flags[1];
        if (r >= NP) break;
#pragma unroll
        for (int u = 0; u < (ITEMS + C8_THREADS - 1) / C8_THREADS; ++u) {
            const int idx = tid + C8_THREADS * u;
            if (idx >= ITEMS) continue;
            const int li = idx & 63, tj = idx >> 6;
            const int jj = tj % 5, rtl = tj / 5;
            const int rt = r * RTPP + rtl;
            const float* base = slab0 + ((rt * 5 + jj) * 64 + li) * 4;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            int sl = 0;
            for (; sl + 4 <= S; sl += 4) {                      // four slabs per batch of loads, summed in slab order
                f32x4 x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = ld16_sc1(base + (long long)(sl + q) * slab_stride);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])::"memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) v += x[q];
            }
            for (; sl < S; ++sl) {
                f32x4 x = ld16_sc1(base + (long long)sl * slab_stride);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)::"memory");
                v += x;
            }
            const int m = rt * 16 + (li & 15), n = j * C8_BN + jj * 16 + (li >> 4) * 4;
            const half4 hb4 = p.bias_n != nullptr ? *reinterpret_cast<const half4*>(p.bias_n + n) : (half4){0, 0, 0, 0};
            const half4 he4 = p.rowvec != nullptr ? *reinterpret_cast<const half4*>(p.rowvec + (long long)(m / p.rows_per_vec) * p.ldrv + n) : (half4){0, 0, 0, 0};
            const half4 hr4 = p.R != nullptr ? *reinterpret_cast<const half4*>(p.R + (long long)m * p.ldr + n) : (half4){0, 0, 0, 0};
            half4 o;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[e] + (float)hb4[e] + (float)he4[e] + (float)hr4[e];
                o[e] = (half_t)t;
                const float f = (float)o[e];
                s1 += f;
                s2 += f * f;
            }
            *reinterpret_cast<half4*>(p.C + (long long)m * p.ldc + n) = o;
            scratch[idx] = make_float2(s1, s2);
        }
        if (p.gn_part != nullptr) {   // GroupNorm partial statistics of the output: (image, row part, group), fixed order
            __syncthreads();
            const int ngrp = C8_BN / cpg_o;
            if (wid < ngrp) {
                float a = 0.f, b = 0.f;
                for (int e = lane; e < ITEMS; e += 64) {
                    const int tj = e >> 6;
                    const int col = (tj % 5) * 16 + ((e & 63) >> 4) * 4;
                    if (col / cpg_o == wid) {
                        const float2 t = scratch[e];
                        a += t.x;
                        b += t.y;
                    }
                }
                a = wave_sum(a);
                b = wave_sum(b);
                if (lane == 0) {
                    const int row = r * RTPP * 16;
                    const int img = row / hw, chunk = (row - img * hw) / (RTPP * 16);
                    float* o = p.gn_part + (((long long)img * (hw / (RTPP * 16)) + chunk) * 32 + (j * C8_BN) / cpg_o + wid) * 2;
                    o[0] = a;
                    o[1] = b;
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {   // the last workgroup of the tile to leave resets the counters for the next launch
        const int e = __hip_atomic_fetch_add(cnt + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e == S - 1) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int WD, int IMGS>
constexpr int c8_lds_bytes() {
    return C8_RING * C8_WSTAGE + 2 * IMGS * (WD + 2) * (WD + 2) * 32 + 2 * IMGS * C8_MAXCW * 4 + IMGS * 32 * 4 + 64;
}

template <int WD, int IMGS, bool GN, bool UP>
void c8_launch(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = c8_lds_bytes<WD, IMGS>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8_kernel<WD, IMGS, GN, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv8_kernel<WD, IMGS, GN, UP>), dim3((p.N / C8_BN) * p.c8_S), dim3(C8_THREADS), lds, stream, p);
}

}  // namespace

// Does this convolution run on the row-resident kernel?  Fills the slab split.  (The caller provides p.partial and p.sync.)
bool conv8_plan(const GemmParams& p, int* S_out) {
    if (!(p.conv && p.ksize == 3 && p.stride == 1 && (p.pad < 0 || p.pad == 1) && p.batch == 1 && p.act == 0 && p.alpha == 1.0f && p.bias_m == nullptr &&
          p.bm == 0 && p.bn == 0 && p.splitk == 0 && p.partial != nullptr && p.sync != nullptr && p.gn_scale == nullptr && p.stat_out == nullptr &&
          p.ln_stat == nullptr))
        return false;
    const bool same = p.Hv == p.Hs && p.Wv == p.Ws;
    const bool up = p.Hv == 2 * p.Hs && p.Wv == 2 * p.Ws && p.C2 == 0 && p.gn_in_part == nullptr;
    if (!(same || up) || p.Ho != p.Hv || p.Wo != p.Wv || p.Ho != p.Wo) return false;
    if (!(p.Wo == 8 || p.Wo == 16) || p.M != 2 * p.Wo * p.Wo) return false;             // two images (the CFG pair of a batch-1 step)
    if (p.N % C8_BN || p.C1 % 16 || p.C2 % 16 || (p.N & 3) || p.ldc % 4 || (p.R != nullptr && p.ldr % 4) || (p.rowvec != nullptr && p.ldrv % 4)) return false;
    if (p.n_valid > 0 && p.n_valid < p.N) return false;
    const int Cin = p.C1 + p.C2, nsub = Cin / 16, ntn = p.N / C8_BN;
    if (p.K != 9 * Cin || p.ldw != p.K) return false;
    if (p.gn_in_part != nullptr && (Cin % 32 || p.gn_gamma == nullptr || p.gn_beta == nullptr || p.gn_in_P <= 0)) return false;
    if (p.gn_part != nullptr && ((p.N / 32) % 4 || C8_BN % (p.N / 32))) return false;
    if (p.gn_in_part != nullptr && C8_MAXCW / (Cin / 32) + 2 > 16) return false;        // groups one slab can touch (mean / rstd table)
    int S = 256 / ntn;
    if (S > nsub) S = nsub;
    // every workgroup's slab must fit the scale / shift table and the weight stream must be worth splitting
    while (S > 1 && (size_t)S * p.M * p.N * sizeof(float) > p.partial_bytes) --S;
    if (S < 1 || (nsub + S - 1) / S * 16 > C8_MAXCW) return false;
    if (ntn * S < 128 || ntn * 4 > 256) return false;
    if (S_out) *S_out = S;
    return true;
}

// number of pixel chunks per image of the GroupNorm partials the kernel emits (GemmParams::gn_part) for this shape
int conv8_gn_chunks(const GemmParams& p) { return p.Wo == 16 ? 8 : 4; }

int conv8_launch(const GemmParams& pin, hipStream_t stream) {
    GemmParams p = pin;
    int S = 0;
    if (!conv8_plan(p, &S)) return LD_ERR_ARG;
    p.c8_S = S;
    p.pad = 1;
    const bool gn = p.gn_in_part != nullptr;
    const bool up = p.Hv == 2 * p.Hs;
    if (p.Wo == 16) {
        if (up) c8_launch<16, 2, false, true>(p, stream);
        else if (gn) c8_launch<16, 2, true, false>(p, stream);
        else c8_launch<16, 2, false, false>(p, stream);
    } else {
        if (up) c8_launch<8, 2, false, true>(p, stream);
        else if (gn) c8_launch<8, 2, true, false>(p, stream);
        else c8_launch<8, 2, false, false>(p, stream);
    }
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}
