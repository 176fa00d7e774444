// conv8: row-resident 3x3 convolution (stride 1, pad 1, optional nearest-2x upsampling of the input) for launches whose M is too small
// to fill the chip with the big-tile kernels — every ResBlock1 convolution (LD.py:5189-5287) and Upsample1 convolution (LD.py:5141-5152)
// of a batch-1 step (UNet batch 2, the CFG pair) on all four levels (conv8_plan declines more than two images: at UNet batch 16 the 8x8
// level measured 0.1 ms per forward behind the general kernels).
//
// Why a separate kernel: at M = 128 .. 8192 rows the 64 x 160 tiles of the general kernel + a split over K + a reduce launch + the
// GroupNorm launches in front run at 10-20 % of either roof (profiles/README.md round 4): every weight byte is re-read once per M tile and
// every activation byte nine times through a per-CU LDS-DMA path that takes in ~30 GB/s from beyond L2.  Here
//   * a workgroup owns a 128-pixel patch (whole image rows; two whole images at 8x8) x 80 output channels x a slab of the input
//     channels, all nine taps.  Grid = patches x (N / 80) x S = ~256 workgroups, one per CU.  The patches that share a weight slab are
//     placed on ONE XCD (speed only: blockIdx -> XCD is a label), so a weight byte leaves HBM once and reaches the other patches' CUs
//     through that XCD's L2;
//   * the weights are read from a copy in the kernel's OWN layout (conv8_repack_launch: [N / 80][Cin / 16][the 25 600-byte LDS image of one
//     ring stage, swizzle and the zero tenth tap included]): a workgroup's whole stream is ONE contiguous byte range and every LDS-DMA
//     instruction copies 1 KB of consecutive bytes (streaming the general [O][tap][I] layout in 32-byte runs measured 6 GB/s per CU:
//     every run pulls its whole 128-byte line from HBM, and the line is evicted before the next sub-slab asks for it);
//   * the activations sit in LDS as a zero-bordered halo image per 16-channel plane, staged through registers by waves 4-7 in groups of
//     two planes (64 contiguous bytes per pixel), double buffered, loaded one group ahead; waves 8-11 issue the weight DMA (a four-stage
//     ring = two groups of two stages, the next group in flight); waves 0-3 only read fragments and issue MFMAs.  (A variant that also applied the GroupNorm + SiLU of the
//     input while staging the halo — no gn_apply launch — is kept in tools/experiments/conv8_fused_groupnorm_r04.hip.txt: correct, but the
//     normalisation is vector-ALU work on the SIMDs that issue the MFMAs, the two ADD on this chip, and every N tile repeats it:
//     5.32 vs 5.19 ms per batch-1 forward, profiles/README.md round 4.)
//   * a k-step of the 16x16x32 MFMA = two taps x 16 channels (the tenth "tap" is a zero weight chunk), so a lane's tap offset is one of
//     five precomputed values and every fragment read is base + immediate;
//   * the S channel-slab partial sums of a (patch, N tile) meet in HBM (fp32, written through with sc1 stores) and are reduced INSIDE the
//     launch: arrive counter -> (bounded wait) -> the eight 16-row parts of the tile are claimed one by one, summed in slab order (bitwise
//     reproducible), finished with bias / time-embedding row / residual, stored as fp16 — and the GroupNorm partial statistics of the
//     OUTPUT are emitted per (image, 16-pixel chunk, group), so the next GroupNorm needs no statistics launch either.  A workgroup that
//     cannot wait (time-out: its peers are not resident) leaves; the last arriver always finds every slab complete and takes whatever is
//     left, so the protocol terminates and is correct under any dispatch order or placement (cdna guide, Guideline 16: sc1 payload stores,
//     every storing wave drains vmcnt, one lane signals; the readers acquire once and load with sc1).
// One launch replaces conv + split-K reduce (+ gn_stats of the next norm).  Small patches keep S small: the fp32 slabs
// (S x M x N x 4 bytes, written and read once) were the largest cost of a first version with 512-row tiles (S = 16: 84 MB per launch).
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "gemm.h"

namespace {

constexpr int C8_BN = 80;                 // output channels per workgroup
constexpr int C8_THREADS = 768;            // 12 waves: 0-3 MFMA consumers, 4-7 halo staging, 8-11 weight DMA
constexpr int C8_RING = 4;                // weight ring stages (one 16-channel sub-slab each)
constexpr int C8_WSTAGE = 5 * C8_BN * 64; // 5 k-steps x 80 rows x 64 B = 25 wave-instructions of 1 KB

__device__ uint4 g_c8_zero[1024];          // 16 KB of zeros: stands in for an absent bias / row vector / residual (row stride 0)

__device__ __forceinline__ int c8_g(int x) { return (0x78 >> (2 * (x & 3))) & 3; }   // {0, 2, 3, 1}: 64-byte-row swizzle of the weight stage (gemm5's)

__device__ __forceinline__ void st16_sc1(float* p, f32x4 v) {   // write-through 16-byte store (agent scope)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void glds16b(unsigned voff, const char* sbase, unsigned lds_base) {   // LDS-DMA, scalar base + per-lane byte offset
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_base)
        : "memory");
}
__device__ __forceinline__ f32x4 ld16_sc1(const float* p) {      // L1-bypassing 16-byte load, result usable after the caller's vmcnt wait
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// WD: image width (= height).  A patch = 128 output pixels: two whole images (WD = 8) or 128 / WD whole rows of one image.
template <int WD, bool UP>
__global__ __launch_bounds__(C8_THREADS, 3) void conv8_kernel(const GemmParams p) {
    constexpr int TI = WD == 8 ? 2 : 1;                      // images per patch
    constexpr int RH = WD == 8 ? 8 : 128 / WD;               // image rows per patch (per image)
    constexpr int HP = WD + 2, HR = RH + 2, HPI = HR * HP, HPIX = TI * HPI;   // halo: pixels per row, rows, per image, total
    constexpr int PLANE_B = HPIX * 32;                       // one 16-channel plane of the halo image
    constexpr int GROUP_B = 2 * PLANE_B;                     // staged two planes (32 channels = 64 contiguous source bytes per pixel) at a time
    constexpr int NCH = (HPIX * 4 + 255) / 256;              // halo chunks per thread of the four staging waves and group
    constexpr int ITEMS = 5 * 64;                            // float4 items of one 16-row part
    extern __shared__ __attribute__((aligned(16))) char smem8[];
    char* const wring = smem8;
    char* const halo = smem8 + C8_RING * C8_WSTAGE;
    int* const flags = reinterpret_cast<int*>(halo + 2 * GROUP_B);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int NTN = p.N / C8_BN, S = p.c8_S, TMS = p.M / 128;
#ifdef LD_AB_BUILD
    // phase clocks (tools/conv8_phases.py): waves 0 and 4 of every workgroup stamp s_memrealtime (100 MHz) at the phase boundaries
    unsigned long long* const sbase = (p.dbg & 1) ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.partial) + p.partial_bytes - 65536) + blockIdx.x * 32 : nullptr;
    unsigned long long* const stamps = sbase != nullptr ? sbase + (tid >= 256 ? 8 : 0) : nullptr;   // wave 0: [0..7], wave 4: [8..15]; [16..]: wait-time sums
#define C8_STAMP(i) do { if (stamps != nullptr && (tid == 0 || tid == 256)) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    unsigned long long acc_a = 0, acc_b = 0, tmark = 0;   // shader-clock sums of the time a wave spends in its barrier / in its vmcnt wait
#define C8_T0() do { if (stamps != nullptr) tmark = __builtin_amdgcn_s_memtime(); } while (0)
#define C8_TA() do { if (stamps != nullptr) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_a += t_ - tmark; tmark = t_; } } while (0)
#define C8_TB() do { if (stamps != nullptr) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_b += t_ - tmark; tmark = t_; } } while (0)
#else
#define C8_STAMP(i) do { } while (0)
#define C8_T0() do { } while (0)
#define C8_TA() do { } while (0)
#define C8_TB() do { } while (0)
#endif
    C8_STAMP(0);
    // block -> (weight slab q = (s, j), patch tm): logical id q * TMS + tm, and every XCD takes a contiguous run of logical ids (blocks b and
    // b + 8 share an XCD under round-robin placement): the patches of one slab sit on ONE XCD and stream its weights through that L2 together
    const int lid = xcd_remap(blockIdx.x, NTN * S * TMS);
    const int q = lid / TMS, tm = lid - q * TMS;
    const int j = q % NTN, s = q / NTN;
#ifdef LD_AB_BUILD
    if (stamps != nullptr && tid == 0) {   // which XCD runs this (slab, patch): hardware id, read for the placement check of tools/conv8_phases.py
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[6] = xcc & 15u;
        stamps[7] = ((unsigned long long)q << 32) | (unsigned)tm;
    }
#endif
    const int Cin = p.C1 + p.C2;
    const int nsub = Cin >> 4;
    const int sb = (int)((long long)s * nsub / S), se = (int)((long long)(s + 1) * nsub / S);
    const int nloc = se - sb;
    const int HW = WD * WD;
    const int img0 = WD == 8 ? tm * 2 : tm / (WD / RH), y0 = WD == 8 ? 0 : (tm % (WD / RH)) * RH;   // first image / first image row of the patch

    // ------------------------------------------------------------------------------------------ weight stream (waves 8-11)
    // this workgroup's stages are consecutive 25 600-byte blocks of the repacked weights: piece i of a stage = bytes [1024 i, 1024 i + 1024)
    const char* wbase = reinterpret_cast<const char*>(p.W8) + ((long long)j * nsub + sb) * C8_WSTAGE;   // wave-uniform, advances one stage per issue
    const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)wring);
    const int wd = wid - 8;              // DMA wave 0..3 takes pieces wd, wd + 4, ...: 7 (wd = 0) or 6 one-KB pieces of a stage
    const unsigned wlane = (unsigned)(wd * 1024 + lane * 16);
    auto w_issue = [&](int slot) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            if (u == 6 && wd != 0) break;
            glds16b(wlane + (unsigned)(u * 4096), wbase, ring_base + (unsigned)(slot * C8_WSTAGE + (wd + 4 * u) * 1024));
        }
        wbase += C8_WSTAGE;
    };
    auto w_wait = [&](int ahead) {       // this wave's pieces of a stage have landed: all but the pieces of the `ahead` stages issued after it
        if (wd == 0) {
            if (ahead >= 2) wait_vmcnt<14>();
            else if (ahead == 1) wait_vmcnt<7>();
            else wait_vmcnt<0>();
        } else {
            if (ahead >= 2) wait_vmcnt<12>();
            else if (ahead == 1) wait_vmcnt<6>();
            else wait_vmcnt<0>();
        }
    };

    // ------------------------------------------------------------------------------------------ halo staging (waves 4-7)
    const bool halo_wave = wid >= 4 && wid < 8;
    // chunk c of a group: pixel c >> 2, plane (c >> 1) & 1, half-plane c & 1 — four consecutive lanes read one pixel's 64 contiguous bytes
    const int t4 = tid - 256;
    int hsp[NCH];                  // source pixel (linear, NHWC, in pixels; -1 = zero border) of this thread's chunks
    uint4 hregA[NCH], hregB[NCH];  // group h travels in set h & 1: loaded one whole group before it is normalised and stored (the L2 latency of
                                   // a load-then-store staging was the pole of the loop: 2.2 us per group against 0.9 us of MFMA)
    if (halo_wave) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int pix = (t4 + 256 * u) >> 2;
            const int il = pix / HPI, r2 = pix - il * HPI;
            const int hy = r2 / HP, hx = r2 - hy * HP;
            const int iy = y0 - 1 + hy, ix = hx - 1;                                  // position in the (upsampled) image the convolution sees
            const bool in = pix < HPIX && iy >= 0 && iy < WD && ix >= 0 && ix < WD;
            const int sy = UP ? iy >> 1 : iy, sx = UP ? ix >> 1 : ix;
            hsp[u] = in ? ((img0 + il) * p.Hs + sy) * p.Ws + sx : -1;
        }
    }
    // (both take a chunk range [u0, u1): the work of a group is spread over its two sub-slab iterations — all of it in the first one made
    // every other iteration 2800 clocks long against 1600 of the consumers)
    constexpr int NCH0 = (NCH + 1) / 2;
    using U0 = std::integral_constant<int, 0>;
    using UH = std::integral_constant<int, NCH0>;
    using UN = std::integral_constant<int, NCH>;
    auto halo_load = [&](int g, uint4 (&hreg)[NCH], auto u0c, auto u1c) {   // group g = sub-slabs sb + 2g, sb + 2g + 1 (the second one may lie beyond the slab: loaded, never read)
        constexpr int u0 = decltype(u0c)::value, u1 = decltype(u1c)::value;   // (compile-time range: no branch around a load)
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            const int c = t4 + 256 * u;
            int ss = sb + 2 * g + ((c >> 1) & 1);
            ss = ss < nsub ? ss : nsub - 1;
            const int c0 = ss * 16;
            const bool second = c0 >= p.C1;
            const half_t* src = second ? p.A2 : p.A;
            const int Cs = second ? p.C2 : p.C1, cl = second ? c0 - p.C1 : c0;
            // (always a load — a select between a load and zeros makes hipcc branch around every load and wait for each one: the border
            // pixels read source pixel 0 and are zeroed when they are stored)
            hreg[u] = ld16(src + (long long)(hsp[u] >= 0 ? hsp[u] : 0) * Cs + cl + (c & 1) * 8);
        }
    };
    auto halo_store = [&](int g, int buf, const uint4 (&hreg)[NCH], auto u0c, auto u1c) {
        constexpr int u0 = decltype(u0c)::value, u1 = decltype(u1c)::value;
        (void)g;
        char* const dst = halo + buf * GROUP_B + ((t4 >> 1) & 1) * PLANE_B + (t4 & 1) * 16;   // (a thread's chunks all have the same plane and half-plane)
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            const int c = t4 + 256 * u;
            if ((NCH * 256 > HPIX * 4) && u == NCH - 1 && c >= HPIX * 4) break;   // (only the last chunk can be ragged)
            *reinterpret_cast<uint4*>(dst + (c >> 2) * 32) = hsp[u] >= 0 ? hreg[u] : zero16();   // zero border
        }
    };

    // ------------------------------------------------------------------------------------------ prologue
    if (wid >= 8) {
#pragma unroll
        for (int k = 0; k < C8_RING; ++k)
            if (k < nloc) w_issue(k);
    } else if (halo_wave) {
        halo_load(0, hregA, U0{}, UN{});
        if (2 < nloc) halo_load(1, hregB, U0{}, UN{});
    }
    if (halo_wave) halo_store(0, 0, hregA, U0{}, UN{});
    C8_STAMP(1);

    // ------------------------------------------------------------------------------------------ main loop: 16-channel sub-slabs, one barrier per two
    // Three roles with SEPARATE code paths (one s_barrier per GROUP of two sub-slabs joins them — the ring holds two groups of weight stages, the halo
    // buffers two groups of planes; with a barrier per sub-slab the batch-1 forward measured 0.6 % slower, profiles/r04_ab_conv8.txt): waves 0-3 = consumers — fragment reads and MFMA only
    // (wave w owns rows 32 w .. 32 w + 31 of the patch: two 16-row tiles x five 16-column tiles); waves 4-7 = the halo staging with its
    // global loads; waves 8-11 = the weight DMA with its counted waits.  Measured on the way here (profiles/README.md round 4): with shared
    // code hipcc parks a vmcnt(0) for the halo registers in front of every fragment read, which drains the weight ring each sub-slab
    // (3.5x slower); with the DMA issued by the consumers its 6-7 issues per sub-slab (60-180 cycles each) sit in front of the same wave's
    // 50 MFMAs (1.15 us per sub-slab instead of ~0.5).
    f32x4 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wid >= 8) {           // ---- weight DMA: stage k + 3 goes out as soon as everybody is past stage k - 1 (three stages in flight)
        // one barrier per GROUP of two sub-slabs (k even): the consumers run the group's second sub-slab without meeting the other roles again.
        // The four ring slots hold two groups: group g + 1 goes out behind the barrier of group g into the slots group g - 1 was read from
        // (groups 0 and 1 went out in the prologue), and has one group of the loop to land.
        for (int k = 0; k < nloc; k += 2) {
            C8_T0();
            if (k == 0 && nloc > 2) {                           // group 1 (one or two stages) may still be in flight
                const bool two = nloc > 3;
                if (wd == 0) { if (two) wait_vmcnt<14>(); else wait_vmcnt<7>(); }
                else { if (two) wait_vmcnt<12>(); else wait_vmcnt<6>(); }
            } else {
                wait_vmcnt<0>();
            }
            C8_TB();                                            // (time in the vmcnt wait)
            __builtin_amdgcn_s_barrier();
            C8_TA();                                            // (time in the barrier)
#ifdef LD_AB_BUILD
            if (p.dbg & 32) continue;
#endif
            if (k >= 2) {
                if (k + 2 < nloc) w_issue((k + 2) & (C8_RING - 1));
                if (k + 3 < nloc) w_issue((k + 3) & (C8_RING - 1));
            }
        }
#ifdef LD_AB_BUILD
        if (sbase != nullptr && tid == 512) { sbase[18] = acc_a; sbase[19] = acc_b; }
#endif
    } else if (wid >= 4) {    // ---- halo staging
        // (a vector-ALU wave beside an MFMA wave on the same SIMD: at equal priority their issue times ADD on this chip, with the vector wave
        // at priority 1 they overlap — tools/micro/coexec.hip, profiles/README.md)
        __builtin_amdgcn_s_setprio(1);
        for (int k = 0; k < nloc; ++k) {
            if (!(k & 1)) {                                     // (one barrier per group: see the weight DMA loop)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's halo writes of the group that starts here
                C8_T0();
                __builtin_amdgcn_s_barrier();
                C8_TA();
            }
            const int g = k >> 1;
            // during group g: group g + 1 (loaded a group ago) goes into the other buffer — its previous contents were read before the barrier
            // that opened this group — and the loads of group g + 2 go out into the registers that just became free; first half of the
            // chunks in the group's first iteration, second half in its second one
#ifdef LD_AB_BUILD
            if (p.dbg & 16) continue;
#endif
            if (2 * g + 2 < nloc) {
                const bool more = 2 * g + 4 < nloc;
                if (g & 1) {
                    if (k & 1) {
                        halo_store(g + 1, 0, hregA, UH{}, UN{});
                        if (more) halo_load(g + 2, hregB, UH{}, UN{});
                    } else {
                        halo_store(g + 1, 0, hregA, U0{}, UH{});
                        if (more) halo_load(g + 2, hregB, U0{}, UH{});
                    }
                } else {
                    if (k & 1) {
                        halo_store(g + 1, 1, hregB, UH{}, UN{});
                        if (more) halo_load(g + 2, hregA, UH{}, UN{});
                    } else {
                        halo_store(g + 1, 1, hregB, U0{}, UH{});
                        if (more) halo_load(g + 2, hregA, U0{}, UH{});
                    }
                }
            }
        }
    } else {
        int pixa[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rt = wid * 2 + i;
            if (WD == 8) pixa[i] = (rt >> 2) * HPI + ((rt & 3) * 2 + (fr >> 3)) * HP + (fr & 7);
            else if (WD == 16) pixa[i] = rt * HP + fr;
            else if (WD == 32) pixa[i] = (rt >> 1) * HP + (rt & 1) * 16 + fr;
            else pixa[i] = (rt >> 2) * HP + (rt & 3) * 16 + fr;
        }
        const int a_lane0 = pixa[0] * 32 + (fq & 1) * 16, a_lane1 = pixa[1] * 32 + (fq & 1) * 16;
        int toff[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            int tap = 2 * t + (fq >> 1);
            tap = tap > 8 ? 8 : tap;                            // (the tenth tap's weights are zero: any finite activation will do)
            toff[t] = ((tap / 3) * HP + (tap % 3)) * 32;
        }
        const int b_lane = fr * 64 + ((fq ^ c8_g(fr >> 2)) << 4);
        half8 fa[3][2], fb[3][5];                               // fragments of k-step t in set t % 3: read TWO steps ahead (one step = 160 MFMA cycles
                                                                // does not cover an LDS round trip for the one consumer wave of a SIMD)
        auto read_frags = [&](const char* wst, const char* hb, int t, int set) {
            fa[set][0] = as_half8(*reinterpret_cast<const uint4*>(hb + a_lane0 + toff[t]));
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) fb[set][jj] = as_half8(*reinterpret_cast<const uint4*>(wst + t * 5120 + jj * 1024));
            fa[set][1] = as_half8(*reinterpret_cast<const uint4*>(hb + a_lane1 + toff[t]));
        };
#ifdef LD_AB_BUILD
        const unsigned long long loop_t0 = sbase != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        for (int k = 0; k < nloc; ++k) {
            if (!(k & 1)) {                                     // (one barrier per group: see the weight DMA loop)
                C8_T0();
                __builtin_amdgcn_s_barrier();
                C8_TA();
            }
            const char* wst = wring + (k & (C8_RING - 1)) * C8_WSTAGE + b_lane;
            const char* hb = halo + ((k >> 1) & 1) * GROUP_B + (k & 1) * PLANE_B;
#ifdef LD_AB_BUILD
            const bool no_rd = (p.dbg & 4) != 0, no_mm = (p.dbg & 8) != 0;
            if (!no_rd || k == 0)
#endif
            {
                read_frags(wst, hb, 0, 0);
                read_frags(wst, hb, 1, 1);
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                // the fragments of step t are in: everything but the seven reads of step t + 1 (issued behind the MFMAs of step t - 1)
                if (t + 1 < 5) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#ifdef LD_AB_BUILD
                if (!no_rd || k == 0)
#endif
                if (t + 2 < 5) read_frags(wst, hb, t + 2, (t + 2) % 3);
#ifdef LD_AB_BUILD
                if (!no_mm)
#endif
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 5; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[t % 3][jj], fa[t % 3][i], acc[i][jj], 0, 0, 0);
                // the seven fragment reads go out one behind each of the first seven MFMAs (in a block in front their issue would sit in
                // front of this wave's MFMAs; left to itself hipcc reads each fragment right before its first use and waits every time)
                if (t + 2 < 5) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef LD_AB_BUILD
        if (sbase != nullptr && tid == 0) sbase[20] = __builtin_amdgcn_s_memtime() - loop_t0;
#endif
    }

    C8_STAMP(2);
#ifdef LD_AB_BUILD
    if (sbase != nullptr && (tid == 0 || tid == 256)) sbase[tid ? 17 : 16] = acc_a;   // barrier time of the consumer / halo wave
#endif
    // ------------------------------------------------------------------------------------------ epilogue
    const long long tile = (long long)tm * NTN + j;
    const int cpg_o = p.N / 32;
    float4* const scratch = reinterpret_cast<float4*>(wring);   // [8 parts][320 items]: the ring is quiet by the time it is written
    // one float4 item = four consecutive output channels of one pixel: bias / time-embedding row / residual, fp16 store, and the statistics of
    // what was stored per channel PAIR (a group boundary never splits a pair: N / 32 is even)
    auto finish_item = [&](int r, int jj, int li, f32x4 v, half4 hb4, half4 he4, half4 hr4) {
        const long long m = (long long)tm * 128 + r * 16 + (li & 15);
        const int n = j * C8_BN + jj * 16 + (li >> 4) * 4;
        half4 o;
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = v[e] + (float)hb4[e] + (float)he4[e] + (float)hr4[e];
            o[e] = (half_t)t;
            f[e] = (float)o[e];
        }
        *reinterpret_cast<half4*>(p.C + m * p.ldc + n) = o;
        scratch[r * ITEMS + jj * 64 + li] = make_float4(f[0] + f[1], f[0] * f[0] + f[1] * f[1], f[2] + f[3], f[2] * f[2] + f[3] * f[3]);
    };
    auto operands = [&](int r, int jj, int li, half4& hb4, half4& he4, half4& hr4) {
        const long long m = (long long)tm * 128 + r * 16 + (li & 15);
        const int n = j * C8_BN + jj * 16 + (li >> 4) * 4;
        // (always loads: conv8_launch points an absent operand at a page of zeros with a zero row stride)
        hb4 = *reinterpret_cast<const half4*>(p.bias_n + n);
        he4 = *reinterpret_cast<const half4*>(p.rowvec + (m / p.rows_per_vec) * p.ldrv + n);
        hr4 = *reinterpret_cast<const half4*>(p.R + m * p.ldr + n);
    };
    // GroupNorm partial statistics of part r: (image, 16-pixel chunk, group), fixed order; one wave per (part, group)
    auto part_stats = [&](int r, int grp) {
        float a = 0.f, b = 0.f;
        for (int e = lane; e < ITEMS; e += 64) {
            const int col = (e >> 6) * 16 + ((e & 63) >> 4) * 4;
            const float4 t = scratch[r * ITEMS + e];
            if (col / cpg_o == grp) {
                a += t.x;
                b += t.y;
            }
            if ((col + 2) / cpg_o == grp) {
                a += t.z;
                b += t.w;
            }
        }
        a = wave_sum(a);
        b = wave_sum(b);
        if (lane == 0) {
            const long long row = (long long)tm * 128 + r * 16;
            const int img = (int)(row / HW), chunk = (int)(row - (long long)img * HW) >> 4;
            float* o = p.gn_part + (((long long)img * (HW / 16) + chunk) * 32 + (j * C8_BN) / cpg_o + grp) * 2;
            o[0] = a;
            o[1] = b;
        }
    };
    const int ngrp = C8_BN / cpg_o;                             // groups of this N tile: 2 / 4 / 8

    if (S == 1) {   // no split: every consumer wave finishes its own 32 rows straight from the accumulators
        __syncthreads();                                        // (everybody is done with the ring: scratch lives there)
        if (wid < 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                half4 hb4[5], he4[5], hr4[5];
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) operands(wid * 2 + i, jj, lane, hb4[jj], he4[jj], hr4[jj]);
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) finish_item(wid * 2 + i, jj, lane, acc[i][jj], hb4[jj], he4[jj], hr4[jj]);
            }
        }
        if (p.gn_part != nullptr) {
            __syncthreads();
            if (wid < 8)
                for (int grp = 0; grp < ngrp; ++grp) part_stats(wid, grp);   // wave w: part w
        }
        C8_STAMP(5);
        return;
    }

    // ---- partial sums -> HBM (write-through)
    float* const slab0 = p.partial + tile * (8 * 5 * 256);
    const long long slab_stride = (long long)TMS * NTN * (8 * 5 * 256);
    {
        float* mine = slab0 + (long long)s * slab_stride;
        if (wid < 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) st16_sc1(mine + (((wid * 2 + i) * 5 + jj) * 64 + lane) * 4, acc[i][jj]);
        }
    }
    wait_vmcnt<0>();                                            // every storing wave drains its stores ...
    __syncthreads();                                            // ... before ONE lane signals for the workgroup
    C8_STAMP(3);
    int* const cnt = p.sync + tile * 4;                         // [0] arrivals, [1] claims, [2] leavers
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
#ifdef LD_AB_BUILD
        if ((p.dbg & 2) && t != S - 1) ok = 0;                  // LD_C8_NO_WAIT=1 (tools/conv8_timeout_check.py): behave as if no peer were resident — the last arriver reduces the whole tile alone
#endif
        if (ok && t != S - 1) {                                 // not the last arriver: wait (bounded) until every slab of the tile is there
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S) {
                __builtin_amdgcn_s_sleep(2);
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
    C8_STAMP(4);

    // A claim = PC consecutive 16-row parts (PC = 8 / S for a split of 2 or 4: every workgroup then makes ONE round trip to the slabs).
    // Every load of the claim goes out before the first wait: the epilogue operands, then the S slabs of every item; slabs are summed in
    // slab order.  SS = the split as a compile-time constant (0: run-time S <= 16, clamped duplicates beyond it; further slabs one at a time).
    auto do_claim = [&](auto SSc, auto PCc) {
        constexpr int SS = decltype(SSc)::value, PC = decltype(PCc)::value;
        constexpr int NIT = (PC * ITEMS + C8_THREADS - 1) / C8_THREADS, NL = SS ? SS : 16;
        __syncthreads();                                        // (flags / scratch of the previous claim are consumed)
        if (tid == 0) flags[1] = __hip_atomic_fetch_add(cnt + 1, PC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int r0 = flags[1];                                // parts r0 .. r0 + PC - 1: rows 16 r .. 16 r + 15 of the patch
        if ((unsigned)r0 >= 8u) return false;                   // (also a counter that was not zeroed: never an out-of-range part)
        half4 hb4[NIT], he4[NIT], hr4[NIT];
        f32x4 x[NIT][NL];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int idx = tid + C8_THREADS * u;
            const int id = idx < PC * ITEMS ? idx : 0;
            const int pr = id / ITEMS, it = id - pr * ITEMS;
            const int li = it & 63, jj = it >> 6, r = r0 + pr;
            operands(r, jj, li, hb4[u], he4[u], hr4[u]);
            const float* base = slab0 + ((r * 5 + jj) * 64 + li) * 4;
#pragma unroll
            for (int q2 = 0; q2 < NL; ++q2) x[u][q2] = ld16_sc1(base + (long long)(SS ? q2 : (q2 < S ? q2 : S - 1)) * slab_stride);
        }
        // the asm loads above are untracked (DESIGN "hipcc traps" (c)): the wait names every destination register as an in/out operand, so
        // no copy, spill or reuse of x can be scheduled between a load and the wait (one wait statement per item: <= 16 operands each)
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            if constexpr (NL == 16)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[u][0]), "+v"(x[u][1]), "+v"(x[u][2]), "+v"(x[u][3]), "+v"(x[u][4]), "+v"(x[u][5]), "+v"(x[u][6]), "+v"(x[u][7]),
                             "+v"(x[u][8]), "+v"(x[u][9]), "+v"(x[u][10]), "+v"(x[u][11]), "+v"(x[u][12]), "+v"(x[u][13]), "+v"(x[u][14]), "+v"(x[u][15])::"memory");
            else if constexpr (NL == 8)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[u][0]), "+v"(x[u][1]), "+v"(x[u][2]), "+v"(x[u][3]), "+v"(x[u][4]), "+v"(x[u][5]), "+v"(x[u][6]), "+v"(x[u][7])::"memory");
            else if constexpr (NL == 4)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[u][0]), "+v"(x[u][1]), "+v"(x[u][2]), "+v"(x[u][3])::"memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[u][0]), "+v"(x[u][1])::"memory");
        }
        __builtin_amdgcn_sched_barrier(0);                      // (nothing that reads x moves above the wait)
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int idx = tid + C8_THREADS * u;
            if (idx >= PC * ITEMS) continue;
            const int pr = idx / ITEMS, it = idx - pr * ITEMS;
            const int li = it & 63, jj = it >> 6, r = r0 + pr;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q2 = 0; q2 < NL; ++q2)
                if (SS || q2 < S) v += x[u][q2];
            if (!SS) {
                for (int sl = 16; sl < S; ++sl) {               // (run-time splits beyond 16: one slab at a time)
                    f32x4 y = ld16_sc1(slab0 + ((r * 5 + jj) * 64 + li) * 4 + (long long)sl * slab_stride);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(y)::"memory");
                    v += y;
                }
            }
            finish_item(r, jj, li, v, hb4[u], he4[u], hr4[u]);
        }
        if (p.gn_part != nullptr) {
            __syncthreads();
            for (int pi = wid; pi < PC * ngrp; pi += 12) part_stats(r0 + pi / ngrp, pi % ngrp);
        }
        return true;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;
    if (take) {
        if (S == 2) while (do_claim(I2{}, I4{})) {}
        else if (S == 4) while (do_claim(I4{}, I2{})) {}
        else if (S == 8) while (do_claim(I8{}, I1{})) {}
        else if (S == 16) while (do_claim(I16{}, I1{})) {}
        else while (do_claim(I0{}, I1{})) {}
    }
    __syncthreads();
    C8_STAMP(5);
    if (tid == 0) {   // the last workgroup of the tile to leave resets the counters for the next launch
        const int e = __hip_atomic_fetch_add(cnt + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e == S - 1) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int WD>
constexpr int c8_lds_bytes() {
    constexpr int TI = WD == 8 ? 2 : 1, RH = WD == 8 ? 8 : 128 / WD;
    return C8_RING * C8_WSTAGE + 4 * TI * (RH + 2) * (WD + 2) * 32 + 64;
}

template <int WD, bool UP>
int c8_launch(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = c8_lds_bytes<WD>();
    static_assert(lds <= 163840, "LDS budget");
    // > 64 KB of dynamic LDS needs the attribute on EVERY device the kernel is launched on: one once_flag per device (thread-safe), and a
    // failed call is reported (the launch itself would only fail later, as LD_ERR_HIP)
    constexpr int MAXDEV = 64;
    static std::once_flag once[MAXDEV];
    static bool ok[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return LD_ERR_HIP;
    std::call_once(once[dev], [&] {
        ok[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8_kernel<WD, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
    });
    if (!ok[dev]) return LD_ERR_HIP;
    const int groups = (p.N / C8_BN) * p.c8_S, tms = p.M / 128;
    hipLaunchKernelGGL((conv8_kernel<WD, UP>), dim3((unsigned)(groups * tms)), dim3(C8_THREADS), lds, stream, p);
    return LD_OK;
}

template <int WD>
int c8_dispatch(const GemmParams& p, hipStream_t stream) {
    return p.Hv == 2 * p.Hs ? c8_launch<WD, true>(p, stream) : c8_launch<WD, false>(p, stream);
}

// [O][tap][I] (the general layout, ld_op_repack_conv / PK_CONV3) -> conv8's stage images: chunk (j, ss, t, n, pc) of 8 halfs holds
// W[80 j + n][tap = 2 t + (fq >> 1)][16 ss + 8 (fq & 1) .. + 8] with fq = pc ^ g(n >> 2), zeros for the tenth tap
__global__ __launch_bounds__(256) void conv8_repack_kernel(const half_t* __restrict__ w, int N, int Cin, uint4* __restrict__ dst) {
    const int nsub = Cin >> 4;
    const long long total = (long long)(N / C8_BN) * nsub * 1600;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long long)gridDim.x * blockDim.x) {
        const int within = (int)(c % 1600);
        const long long st = c / 1600;
        const int ss = (int)(st % nsub), j = (int)(st / nsub);
        const int t = within / 320, rem = within - 320 * t;
        const int n = rem >> 2, pc = rem & 3;
        const int fql = pc ^ c8_g(n >> 2);
        const int tap = 2 * t + (fql >> 1), cc = fql & 1;
        dst[c] = tap < 9 ? ld16(w + ((long long)(j * C8_BN + n) * 9 + tap) * Cin + ss * 16 + cc * 8) : zero16();
    }
}

}  // namespace

size_t conv8_weight_bytes(int N, int Cin) { return (size_t)(N / C8_BN) * (Cin / 16) * C8_WSTAGE; }
bool conv8_weight_eligible(int N, int Cin) { return N > 0 && Cin > 0 && N % C8_BN == 0 && Cin % 32 == 0; }

int conv8_repack_launch(const half_t* w, int N, int Cin, half_t* dst, hipStream_t stream) {
    if (w == nullptr || dst == nullptr || !conv8_weight_eligible(N, Cin)) return LD_ERR_ARG;
    const long long total = (long long)(N / C8_BN) * (Cin / 16) * 1600;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv8_repack_kernel, dim3(blocks), dim3(256), 0, stream, w, N, Cin, reinterpret_cast<uint4*>(dst));
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

// Does this convolution run on the row-resident kernel?  Fills the slab split.  (The caller provides p.W8, p.partial and p.sync.)
bool conv8_plan(const GemmParams& p, int* S_out) {
    if (!(p.conv && p.ksize == 3 && p.stride == 1 && (p.pad < 0 || p.pad == 1) && p.batch == 1 && p.act == 0 && p.alpha == 1.0f && p.bias_m == nullptr &&
          p.bm == 0 && p.bn == 0 && p.splitk == 0 && p.W8 != nullptr && p.partial != nullptr && p.sync != nullptr && p.stat_out == nullptr && p.ln_stat == nullptr))
        return false;
    const bool same = p.Hv == p.Hs && p.Wv == p.Ws;
    if (p.gn_scale != nullptr || p.SC1 > 0) return false;                               // (this kernel takes the normalised tensor; no second K segment)
    const bool up = p.Hv == 2 * p.Hs && p.Wv == 2 * p.Ws && p.C2 == 0;
    if (!(same || up) || p.Ho != p.Hv || p.Wo != p.Wv || p.Ho != p.Wo) return false;
    if (!(p.Wo == 8 || p.Wo == 16 || p.Wo == 32 || p.Wo == 64) || p.M % 128 || p.M % (p.Wo * p.Wo)) return false;
    // up to two images (the CFG pair of a batch-1 step): at UNet batch 16 the 8 x 8 level (8 patches x 16 N tiles, split 2) measured 0.1 ms per
    // forward behind the general kernels (tools/ab_unet.py 4, profiles/README.md round 4)
    if (p.M > 2 * p.Wo * p.Wo) return false;
    if (p.N % C8_BN || p.C1 % 32 || p.C2 % 32 || (p.N & 3) || p.ldc % 4 || (p.R != nullptr && p.ldr % 4) || (p.rowvec != nullptr && p.ldrv % 4)) return false;
    if (p.n_valid > 0 && p.n_valid < p.N) return false;
    const int Cin = p.C1 + p.C2, nsub = Cin / 16, ntn = p.N / C8_BN, tms = p.M / 128;
    if (p.K != 9 * Cin) return false;
    // GroupNorm partials of the output: part_stats assumes every 80-column N tile holds WHOLE groups (80 % (N / 32) == 0: N = 320, 640, 1280 ..;
    // N = 960 / 1600 / 1920 would split groups over tiles) of an even number of channels (a float4 item carries channel PAIRS)
    if (p.gn_part != nullptr && ((p.N % 32) || ((p.N / 32) & 1) || (C8_BN % (p.N / 32)))) return false;
    // the patches x N tiles must leave room for a split that fills the chip, and the split must keep the weight stream worth it
    const long long tiles = (long long)tms * ntn;
    if (tiles > 256 || tiles * 4 > LD_SYNC_INTS) return false;
    int S = (int)(256 / tiles);
    // round 6 (tools/conv8_slab_sweep.py, profiles/r06_conv8_slab_sweep.txt: the batch-1 forward per image width and slab count): the plan is at
    // the optimum at 64 / 32 / 16 pixels (S = 1 / 2 / 4; one more doubling costs 0.11 - 0.17 ms, one less 0.2 - 0.7 ms), but the 8 x 8 level
    // (16 tiles -> S = 16, one workgroup per CU) is faster with 8 deeper slabs on half the CUs: 5.169 -> 5.122 ms — sixteen fp32 partials per
    // output meet in HBM where eight do
    if (S > 8) S = 8;
    if (S > nsub / 2) S = nsub / 2;                                       // at least two sub-slabs per workgroup
    while (S > 1 && (size_t)S * p.M * p.N * sizeof(float) > p.partial_bytes) --S;
    if (S < 1 || (size_t)p.M * p.N * sizeof(float) > p.partial_bytes) return false;
    if (tiles * S < 96) return false;
    if (S_out) *S_out = S;
    return true;
}

// number of pixel chunks per image of the GroupNorm partials the kernel emits (GemmParams::gn_part) for this shape
int conv8_gn_chunks(const GemmParams& p) { return p.Wo * p.Wo / 16; }

int conv8_launch(const GemmParams& pin, hipStream_t stream) {
    GemmParams p = pin;
    int S = 0;
    if (!conv8_plan(p, &S)) return LD_ERR_ARG;
#ifdef LD_AB_BUILD
    {   // slab-count sweep (tools/conv8_slab_sweep.py): LD_C8_S_W<width> = number of channel slabs per tile for that image width
        static const char* names[4] = {"LD_C8_S_W8", "LD_C8_S_W16", "LD_C8_S_W32", "LD_C8_S_W64"};
        const char* e = getenv(names[p.Wo == 8 ? 0 : p.Wo == 16 ? 1 : p.Wo == 32 ? 2 : 3]);
        if (e != nullptr) {
            int want = atoi(e);
            const int nsub = (p.C1 + p.C2) / 16;
            const long long tiles = (long long)(p.M / 128) * (p.N / C8_BN);
            if (want < 1) want = 1;
            if (want > 16) want = 16;                                      // (the reducer's generic path sums up to 16 slabs)
            if (want > nsub / 2) want = nsub / 2;
            while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > p.partial_bytes) --want;
            (void)tiles;
            S = want;
        }
    }
#endif
    p.c8_S = S;
    p.pad = 1;
#ifdef LD_AB_BUILD
    p.dbg = (getenv("LD_C8_STAMPS") != nullptr && p.partial_bytes > ((size_t)1 << 20) ? 1 : 0) | (getenv("LD_C8_NO_WAIT") != nullptr ? 2 : 0) |
            (getenv("LD_C8_ABL") != nullptr ? atoi(getenv("LD_C8_ABL")) << 2 : 0);   // ablations (tools/conv8_abl.py; wrong results, timing only): 1 no fragment reads, 2 no MFMAs, 4 no halo staging, 8 no weight DMA in the loop
#endif
    if (p.bias_n == nullptr || p.rowvec == nullptr || p.R == nullptr) {
        // the zero page is a __device__ symbol: one address PER DEVICE (a process that drives several GPUs must not hand device 1 the page of
        // device 0), resolved once per device, thread-safe
        constexpr int MAXDEV = 64;
        static std::once_flag zonce[MAXDEV];
        static const half_t* zpage[MAXDEV];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return LD_ERR_HIP;
        std::call_once(zonce[dev], [&] {
            void* z = nullptr;
            zpage[dev] = hipGetSymbolAddress(&z, HIP_SYMBOL(g_c8_zero)) == hipSuccess ? static_cast<const half_t*>(z) : nullptr;
        });
        const half_t* zero_page = zpage[dev];
        if (zero_page == nullptr) return LD_ERR_HIP;
        if (p.bias_n == nullptr) p.bias_n = zero_page;
        if (p.rowvec == nullptr) { p.rowvec = zero_page; p.ldrv = 0; p.rows_per_vec = 1; }
        if (p.R == nullptr) { p.R = zero_page; p.ldr = 0; }
    }
    int st;
    switch (p.Wo) {
        case 8: st = c8_dispatch<8>(p, stream); break;
        case 16: st = c8_dispatch<16>(p, stream); break;
        case 32: st = c8_dispatch<32>(p, stream); break;
        default: st = c8_dispatch<64>(p, stream); break;
    }
    if (st != LD_OK) return st;
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}
