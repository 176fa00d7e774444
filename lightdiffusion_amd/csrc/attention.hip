// Flash-style attention for the UNet's self / cross attention (softmax(QK^T/sqrt(d)) V, no mask — LD.py:3966-3978) and the
// CLIP text model's causal attention (LD.py:4440-4446).
//
// gfx950 design (32x32x16 f16 MFMA, wave = 64) — flash_attn2_kernel:
//  * one workgroup = NW (4 or 8) waves = 32*NW queries of one (batch, head); each wave owns 32 queries.
//  * S^T = K·Q^T ("swapped" product): the query sits on the MFMA lane, so a lane's 16 accumulator registers are
//    16 keys of ONE query and the online softmax needs no cross-lane traffic except one half-wave exchange.
//  * the S^T accumulator tile is fed straight back as the B operand of O^T += V^T·P^T (cdna guide §3 "accumulator
//    tile as the next MFMA's operand").  K rows are read through the bit-2<->bit-3 row permutation so that the
//    permuted k-order of that trick becomes the natural key order and V^T fragments are single 16-byte LDS reads.
//  * V arrives already transposed ([channel][key], produced by a swapped projection GEMM).
//  * K / V^T tiles (64 keys) go global -> LDS by LDS-DMA (no register staging, no commit stores) into a DOUBLE-buffered
//    pair, so a tile costs one s_barrier and its loads have a whole tile of MFMA + softmax time to land.
//  * K tile: dense rows of d halfs (d/8 16-byte chunks); chunk c of row `row` sits at physical chunk c ^ swz(row), with
//    the XOR width chosen from the row stride so that 8 consecutive rows land in 8 distinct 16-byte slots of a 128-byte
//    bank line (d=40: stride 80 B needs none; d=80: 1 bit; d=160: 2 bits).  The DMA writes lane-linear, so the swizzle
//    is applied to the SOURCE address of each lane and to the fragment read alike.
//  * V^T tile: 32*DV rows of 64 keys (8 chunks), chunk ^= row & 7.  Rows >= d are sourced from a zero page (and the
//    spare "ones" row from a page of fp16 1.0) every tile: the tile is always whole instructions, nothing to pre-fill.
//  * a d that is an odd multiple of 8 leaves one pad chunk in the last QK^T k-step; the matching Q fragment is zero and
//    the pad chunk reads the next row's first chunk (or the zero-filled slack after the last row): 0 * finite = 0.
//  * lazy softmax reference: O / l are rescaled only when a tile maximum exceeds the running reference by > 8.
// Head dims: d % 8 == 0, d <= 160 (SD1.5: 40 / 80 / 160); one head of d = 512 with row-major V: flash_attn512_kernel below (the VAE).
// Template DK = ceil(d/16) k-steps of QK^T.
#include <cstdlib>
#include <type_traits>

#include <string>

#include "kernels.h"
// Ablation builds behind profiles/README.md ("no exp", "no tile sync", "1 of 4 PV MFMAs", "2 of 7 fragment reads"): -DLD_ATT_DBG=1..4.
// They compute WRONG results on purpose (timing only) and exist in the A/B build (make ab, -DLD_AB_BUILD) only: the shipped library
// ignores the macro.
#if !defined(LD_AB_BUILD) || !defined(LD_ATT_DBG)
#undef LD_ATT_DBG
#define LD_ATT_DBG 0
#endif

namespace {

constexpr int AT_THREADS = 256;

__device__ uint4 g_att_zero[8];                                                     // 128 zero bytes
__device__ uint4 g_att_ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};   // 8 x fp16 1.0
__device__ uint4 g_att_one_last = {0u, 0u, 0u, 0x3C000000u};                          // 7 x 0, then fp16 1.0 (the "ones" column of a row-major V tile)
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// VROW: V arrives row-major ([key][channel], e.g. the third block of a fused q|k|v projection) and the V^T fragments are read through
// ds_read_b64_tr_b16 (two transposing 8-byte reads per fragment instead of one 16-byte read); otherwise V^T [channel][key].
template <int DK, bool MASKED, int NW, int WPS, bool VROW>   // NW waves = 32*NW queries per workgroup share each K / V tile
__global__ __launch_bounds__(64 * NW, WPS) void flash_attn2_kernel(const AttnParams p) {
    constexpr int DV = (DK + 1) / 2;
    constexpr bool ONES = (DK & 1) != 0;
    constexpr int KT = 64;
    constexpr int NT2 = 64 * NW;
    constexpr int LIT = (DV * 256 + NT2 - 1) / NT2;   // loader wave-instructions per tile
    constexpr int TILE_CH = LIT * NT2;             // 16-byte chunks per K tile and per V^T tile (whole wave-instructions)
    constexpr int BUF_H = 2 * TILE_CH * 8;         // halfs per buffer (K tile then V^T tile)
    __shared__ __attribute__((aligned(16))) half_t smem[2 * BUF_H];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int qblocks = (p.Lq + 32 * NW - 1) / (32 * NW);
    const int nblk = qblocks * p.H * p.B;
    int bid = xcd_remap(blockIdx.x, nblk);
    const int qb = bid % qblocks;
    bid /= qblocks;
    const int head = bid % p.H, b = bid / p.H;
    const int d = p.d, dch = d >> 3;

    const half_t* Qg = p.Q + (long long)b * p.sQ + head * d;
    const half_t* Kg = p.K + (long long)b * p.sK + head * d;
    const half_t* Vg = VROW ? p.V + (long long)b * p.sV + head * d : p.Vt + (long long)b * p.sV + (long long)head * d * p.ldvt;
    const half_t* zp = reinterpret_cast<const half_t*>(g_att_zero);
    const half_t* op = reinterpret_cast<const half_t*>(VROW ? &g_att_one_last : &g_att_ones);

    const int qrow = qb * (32 * NW) + wid * 32 + r;
    // Q is folded with scale*log2(e) once, so QK^T comes out of the matrix core already in the exp2 domain
    // (one fp16 rounding of q*c2: the same size of error the fp16 P tile carries anyway).
    const float c2 = p.scale * 1.44269504088896340736f;
    half8 qf[DK];
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
        const int c = 16 * ks + 8 * hh;
        float qv[8];
        unpack8((qrow < p.Lq && c < d) ? ld16(Qg + (long long)qrow * p.ldq + c) : zero16(), qv);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] *= c2;
        qf[ks] = as_half8(pack8(qv));
    }

    // K-row swizzle: XOR the low tz bits of the chunk index with row bits [3-tz, 3)
    int tz = 0;
    while (tz < 3 && !((dch >> tz) & 1)) ++tz;
    const int sw_mask = (1 << tz) - 1, sw_shift = 3 - tz;

    // ---- loader state (one entry per wave-instruction)
    const half_t* kp[LIT];
    const half_t* vp[LIT];
    bool kval[LIT], vdat[LIT];
    int krow[LIT], vkey[LIT];
#pragma unroll
    for (int i = 0; i < LIT; ++i) {
        const int q = tid + i * NT2;
        // LDS row `row` holds key perm(row) (bits 2 and 3 swapped): the MFMA operand trick wants lane r to read key perm(r), and
        // with the permutation applied HERE lane r simply reads LDS row r (consecutive lanes, consecutive rows; measured 2-5 % faster
        // than permuting at read time, profiles/README.md)
        const int row = q / dch, cph = q - row * dch;
        const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
        krow[i] = key;
        kval[i] = row < KT;
        kp[i] = kval[i] ? Kg + (long long)key * p.ldk + ((cph ^ ((row >> sw_shift) & sw_mask)) << 3) : zp;
        if (VROW) {
            // row-major V tile: 64 key rows of 4*DV 16-byte chunks (32*DV channels: d data channels, zeros, and with ONES a last
            // column of 1.0 whose O^T row is the softmax denominator); the 64-byte groups of a row are XORed with row bits so that the
            // four rows a transposing read gathers (64 bytes each, per 32-lane half) fall into four different quarters of the 256-byte bank line
            const int vr = q / (4 * DV), pc = q - vr * (4 * DV);
            const int vc = pc ^ ((DV == 2 ? ((vr >> 1) & 1) : DV == 4 ? (vr & 3) : 0) << 2);
            vkey[i] = vr;
            vdat[i] = vr < KT && vc < dch;
            vp[i] = vdat[i] ? Vg + (long long)vr * p.ldv + (vc << 3) : ((ONES && vr < KT && vc == 4 * DV - 1) ? op : zp);
        } else {
            const int vr = q >> 3, vc = (q & 7) ^ (vr & 7);
            vkey[i] = vc << 3;
            vdat[i] = vr < d;
            vp[i] = vdat[i] ? Vg + (long long)vr * p.ldvt + (vc << 3) : ((ONES && vr == 32 * DV - 1) ? op : zp);
        }
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    auto issue = [&](int key0, int buf) {
        const unsigned Kb = smem_base + (unsigned)(buf * BUF_H) * 2u + (unsigned)(wid * 64) * 16u;
        const unsigned Vb = Kb + (unsigned)TILE_CH * 16u;
        if (!MASKED || key0 + KT <= p.Lk) {
#pragma unroll
            for (int i = 0; i < LIT; ++i) glds16(kp[i], Kb + (unsigned)(i * NT2) * 16u);
#pragma unroll
            for (int i = 0; i < LIT; ++i) glds16(vp[i], Vb + (unsigned)(i * NT2) * 16u);
        } else {
#pragma unroll
            for (int i = 0; i < LIT; ++i) glds16((kval[i] && key0 + krow[i] >= p.Lk) ? zp : kp[i], Kb + (unsigned)(i * NT2) * 16u);
#pragma unroll
            for (int i = 0; i < LIT; ++i) glds16((vdat[i] && key0 + vkey[i] >= p.Lk) ? zp : vp[i], Vb + (unsigned)(i * NT2) * 16u);
        }
#pragma unroll
        for (int i = 0; i < LIT; ++i) kp[i] += kval[i] ? (long long)KT * p.ldk : 0;
#pragma unroll
        for (int i = 0; i < LIT; ++i) vp[i] += vdat[i] ? (VROW ? (long long)KT * p.ldv : (long long)KT) : 0;
    };

    // ---- fragment read offsets (halfs, relative to the K tile / V^T tile of a buffer)
    const int prow = r;                                             // (the bit-2 <-> bit-3 key permutation is applied by the loader)
    const int ksw = (prow >> sw_shift) & sw_mask;                   // same for row + 32: only row bits < 3 enter
    // (K offsets are kept per 32-key half of the tile while that is cheap, so that with the buffer index a literal (the tile loop
    // is unrolled over the two buffers) every fragment read is one ds_read with an immediate offset and no address arithmetic)
    constexpr int KSUB = DK <= 4 ? 2 : 1;   // (only the kernels that unroll the tile loop profit)
    int koff[KSUB][DK];
#pragma unroll
    for (int h = 0; h < KSUB; ++h)
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) koff[h][ks] = ((h * 32 + prow) * dch + ((2 * ks + hh) ^ ksw)) << 3;
    int voff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) voff[c] = TILE_CH * 8 + ((r * 8 + ((2 * c + hh) ^ (r & 7))) << 3);   // c = 2*sub + k2
    // VROW: lane (r, hh) = 16-lane group (r >> 4) + 2 hh; lane 4 tq + tp of a group addresses row tq, channels 4 tp .. 4 tp + 3 of its 4 x 16
    // block and receives channel (r & 15) of the block's 4 rows: keys 8 hh + 4 rd + (0..3) (+ 32 sub + 16 k2), channels 32 tt + 16 (r >> 4) + ...
    int voffr[DV];
    {
        const int tq = (r >> 2) & 3, tp = r & 3, g1 = (r >> 4) & 1;
        const int sw = DV == 2 ? ((tq >> 1) & 1) : DV == 4 ? tq : 0;
#pragma unroll
        for (int tt = 0; tt < DV; ++tt)
            voffr[tt] = TILE_CH * 8 + (8 * hh + tq) * (32 * DV) + ((tt ^ sw) << 5) + ((2 * g1 + (tp >> 1)) << 3) + 4 * (tp & 1);
    }

    f32x16 o[DV];
#pragma unroll
    for (int t = 0; t < DV; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    // Softmax reference ("lazy max"): scores leave the QK^T MFMA as s - m_ref because -m_ref is its C operand, so the
    // steady state has no per-element subtract/scale at all.  m_ref moves only when a tile's maximum exceeds it by more
    // than TAU (P <= 2^TAU stays well inside fp16, O / l accumulate in fp32) — then the tile is shifted and O, l rescaled;
    // numerator and denominator always share one reference, so the result is the exact softmax, not a thresholded one.
    constexpr float TAU = 8.0f;
    float m_ref = 0.f, l_run = 0.f;
    f32x16 negm;
#pragma unroll
    for (int e = 0; e < 16; ++e) negm[e] = 0.f;

    const int ntiles = (p.Lk + KT - 1) / KT;
    // Retire the Q loads HERE with the builtin (which hipcc's waitcnt pass models): otherwise the pass parks a vmcnt(0)
    // for them at the first MFMA inside the loop, where it also drains the next tile's in-flight LDS-DMA every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0), expcnt/lgkmcnt untouched
    issue(0, 0);
    auto tile = [&](int t, auto bufc) {
        const int BUF = static_cast<int>(bufc);   // a literal when the caller passes an integral_constant (folds after inlining)
        const int key0 = t * KT;
        wait_vmcnt<0>();                   // this wave's share of tile t has landed ...
        __builtin_amdgcn_s_barrier();      // ... everyone's has, and nobody still reads tile t-1's buffer
        if (t + 1 < ntiles) issue(key0 + KT, BUF ^ 1);
        const half_t* T = smem + BUF * BUF_H;
        // S^T of one 32-key subtile, already relative to the softmax reference: -m_ref is the C operand of the first k-step.
        // That first MFMA is written in asm: D (early-clobber) != C pins the three-address form, so -m_ref is read in place;
        // through the builtin hipcc ties D to C and copies the 16 reference registers.  (Hazards: the next reader of s is the
        // following MFMA with C == D — no wait states; the VALU readers come after builtin MFMAs, which the compiler's hazard
        // recogniser spaces as usual.  DK == 1 has no builtin MFMA behind the asm one, so it uses the builtin.)
        auto qk = [&](int sub, f32x16& s) {
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) {
#if LD_ATT_DBG == 5
                const half8 kf = qf[ks];
#elif LD_ATT_DBG == 4
                const half8 kf = as_half8(ld16(T + (KSUB == 2 ? koff[sub & (KSUB - 1)][0] : sub * 32 * d + koff[0][0])));
#else
                const half8 kf = as_half8(ld16(T + (KSUB == 2 ? koff[sub & (KSUB - 1)][ks] : sub * 32 * d + koff[0][ks])));
#endif
                if (ks == 0 && DK == 1) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0], negm, 0, 0, 0);
                else if (ks == 0) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s) : "v"(kf), "v"(qf[0]), "v"(negm));
                else s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
            }
            if (MASKED) {
                if (p.causal) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int i = (e & 3) + 8 * (e >> 2) + 4 * hh;
                        const int key = (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
                        if (key0 + sub * 32 + key > qrow) s[e] = -INFINITY;
                    }
                }
                if (key0 + sub * 32 + 32 > p.Lk) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int i = (e & 3) + 8 * (e >> 2) + 4 * hh;
                        const int key = (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
                        if (key0 + sub * 32 + key >= p.Lk) s[e] = -INFINITY;
                    }
                }
            }
        };
        // softmax of one subtile and its O^T += V^T P^T.  `ahead` = an S^T tile that was already produced against the current
        // reference (the next subtile's): it follows when the reference moves.
        auto finish = [&](int sub, f32x16& s, f32x16* ahead, bool first) {
            // Softmax runs at raised wave priority.  Measured on this chip (tools/micro/coexec.hip): a VALU stream and an MFMA
            // stream of two waves on one SIMD take the SUM of their times at equal priority (the MFMA wave holds the vector
            // issue port while the matrix pipe is busy) and the MAX when the VALU wave has priority 1.
            __builtin_amdgcn_s_setprio(1);
            // 16 -> 1 by v_max3_f32 (this file is built with -fno-honor-nans: otherwise fmaxf() adds a NaN-quieting
            // v_max x,x per MFMA output), then the other half-wave (the other 16 keys of the same query) by one
            // v_permlane32_swap — no LDS round trip
            float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
            for (int e = 3; e + 1 < 16; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);
            mx = fmaxf(mx, s[15]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            if (first || __any(mx > TAU)) {   // `first`: the reference starts at 0, the first subtile moves it to its own maximum
                const float delta = first ? mx : (mx > TAU ? mx : 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_ref += delta;
                l_run *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] -= delta;
                if (ahead != nullptr) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) (*ahead)[e] -= delta;
                }
                {   // in-place (tied asm operand): keeps -m_ref in ONE register tile across both arms of this branch, otherwise
                    // the allocator gives each arm its own tile and copies 16 registers on the common path
                    const float nm = -m_ref;
#pragma unroll
                    for (int e = 0; e < 16; ++e) asm volatile("v_mov_b32 %0, %1" : "+v"(negm[e]) : "v"(nm));
                }
#pragma unroll
                for (int tt = 0; tt < DV; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[tt][e] *= alpha;
            }
            float psum = 0.f;
            half8 pf[2];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float p0 = __builtin_amdgcn_exp2f(s[e]);
                const float p1 = __builtin_amdgcn_exp2f(s[e + 1]);
                if (!ONES) psum += p0 + p1;
                const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(p0, p1));
                pf[e >> 3][e & 7] = h2[0];
                pf[e >> 3][(e & 7) + 1] = h2[1];
            }
            l_run += psum;
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int tt = 0; tt < DV; ++tt)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
#if LD_ATT_DBG == 6
                    const half8 vf = pf[k2];
#elif LD_ATT_DBG == 4
                    const half8 vf = as_half8(ld16(T + voff[2 * sub]));
#else
                    half8 vf;
                    if (VROW) {
                        const half_t* vsrc = T + voffr[tt] + (32 * sub + 16 * k2) * (32 * DV);
                        const half4v lo = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vsrc)));
                        const half4v hi = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vsrc + 4 * (32 * DV))));
                        vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    } else {
                        vf = as_half8(ld16(T + tt * 32 * 64 + voff[2 * sub + k2]));
                    }
#endif
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[k2], o[tt], 0, 0, 0);
                }
        };
        // (issuing both QK^T products of a tile ahead of the two softmaxes — so that they execute underneath this wave's own
        // VALU work — measured no different: 559 vs 558 us at L=4096, d=40; the resident waves already interleave.)
        {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (MASKED && key0 + sub * 32 >= p.Lk) break;   // wave-uniform: nothing valid in this half tile
                f32x16 s;
                qk(sub, s);
                finish(sub, s, nullptr, t == 0 && sub == 0);
            }
        }
    };
    if (DK <= 4) {   // small heads: unroll over the two buffers (d=40: -4 % at L=4096 / 16384); d=80 is flat, d=160 loses registers
        for (int t = 0; t < ntiles; t += 2) {
            tile(t, std::integral_constant<int, 0>{});
            if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (int t = 0; t < ntiles; ++t) tile(t, t & 1);
    }

    float l_tot;
    if (ONES) l_tot = __shfl(o[DV - 1][15], 32 + r, 64);
    else l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < p.Lq) {
        half_t* Og = p.O + (long long)b * p.sO + (long long)qrow * p.ldo + head * d;
#pragma unroll
        for (int tt = 0; tt < DV; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = tt * 32 + 8 * g + 4 * hh;
                if (dd < d) {
                    half4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[tt][4 * g + e] * inv);
                    *reinterpret_cast<half4*>(Og + dd) = h;
                }
            }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// flash_attn512_kernel (round 6): ONE head of 512 channels, row-major V — the VAE's AttnBlock (LD.py:3591-3642), which until now ran as
// three GEMMs around a row softmax over a materialised L x L score matrix (2.1 GB of fp16 at 1024^2, b = 4).
//  * d = 512 turns the attention unit around: per 32 x 32 score block 32 QK^T + 32 PV MFMAs against 16 exponentials per lane (d = 40: 7
//    MFMAs) — matrix-bound; what the kernel has to manage is REGISTERS: O^T of 32 queries is 16 tiles = 256 accumulator registers and the
//    Q^T fragments of 32 queries are another 128.  (One wave per SIMD with the whole 512-entry file was tried first: hipcc keeps the 256
//    accumulators in AGPRs and spills the 128 Q registers to scratch, one reload per MFMA, whatever is pinned.)
//  * so the 512 channels are split over a PAIR of waves: wave (qb, ch) of a 128-query workgroup (8 waves, two per SIMD) owns queries
//    32 qb .. 32 qb + 31 and channels 256 ch .. 256 ch + 255: half of Q^T (64 VGPRs), half of O^T (128 accumulators).  Per 32-key tile it
//    takes its half of the d-reduction of S^T (16 MFMAs), the pair swaps the fp32 partials through LDS (4 KB each way), both add them in the
//    same order (a + b == b + a: the two waves hold bitwise the same scores, reference and probabilities), run the softmax of the 32 x 32
//    block (16 exponentials: cheap against 32 MFMAs) and their half of O^T += V^T P^T (16 MFMAs).
//  * 32-key tiles: K tile 32 KB + V tile 32 KB, double-buffered by LDS-DMA one tile ahead = 128 KB, + 32 KB of exchange = all 160 KB.
//  * K rows permuted (bits 2 <-> 3) and chunk-swizzled (c ^ (row & 7)) as in flash_attn2_kernel; V rows of 64 chunks with their 64-byte
//    groups XORed with (row & 3) for the transposing reads (the VROW path's rule).
//  * S^T starts at 0 (inline constant C operand); the softmax reference is subtracted on the vector side.
//  * D = 256 (a narrower VAE, e.g. the tests' tiny configuration) is the same kernel with half the k-steps and accumulator tiles.
template <int D, bool MASKED>
__global__ __launch_bounds__(512, 2) void flash_attn512_kernel(const AttnParams p) {
    constexpr int DCH = D / 8, KT = 32;
    constexpr int NKS = D / 32, NT = D / 64;             // QK^T k-steps and O^T tiles of ONE wave (half the channels)
    constexpr int LIT = KT * DCH / 512;                  // loader wave-instructions per tile and operand
    constexpr int TILE_H = KT * D;                       // halfs per K tile and per V tile
    __shared__ __attribute__((aligned(16))) half_t smem[4 * TILE_H + 8 * 2048];   // [buf][K tile | V tile], then 8 x 4 KB of exchange
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qbw = wid & 3, ch = wid >> 2;
    const int r = lane & 31, hh = lane >> 5;
    const int qblocks = (p.Lq + 127) / 128;
    int bid = xcd_remap(blockIdx.x, qblocks * p.B);
    const int qb = bid % qblocks, b = bid / qblocks;
    const half_t* Qg = p.Q + (long long)b * p.sQ;
    const half_t* Kg = p.K + (long long)b * p.sK;
    const half_t* Vg = p.V + (long long)b * p.sV;
    const half_t* zp = reinterpret_cast<const half_t*>(g_att_zero);
    const int qrow = qb * 128 + qbw * 32 + r;
    const float c2 = p.scale * 1.44269504088896340736f;
    half8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        float qv[8];
        unpack8(qrow < p.Lq ? ld16(Qg + (long long)qrow * p.ldq + (D / 2) * ch + 16 * ks + 8 * hh) : zero16(), qv);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] *= c2;
        qf[ks] = as_half8(pack8(qv));
    }
    // loader: wave-instruction i of a tile covers the 16-byte slots i * 512 + tid: LDS row (i * 512 + tid) / DCH, physical chunk % DCH.
    // K: that row holds key perm(row) (bits 2 <-> 3) and the slot's source chunk is chunk ^ (row & 7); V: row = key, chunk ^ ((row & 3) << 2).
    int lkey[LIT], lkoff[LIT], lvoff[LIT];
#pragma unroll
    for (int i = 0; i < LIT; ++i) {
        const int q = i * 512 + tid, row = q / DCH, c = q % DCH;
        lkey[i] = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
        lkoff[i] = (c ^ (row & 7)) << 3;
        lvoff[i] = (c ^ ((row & 3) << 2)) << 3;
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    auto issue = [&](int key0, int buf) {
        const unsigned Kb = smem_base + (unsigned)(buf * 2 * TILE_H) * 2u + (unsigned)(wid * 64) * 16u;
        const unsigned Vb = Kb + (unsigned)TILE_H * 2u;
#pragma unroll
        for (int i = 0; i < LIT; ++i) {
            const int key = key0 + lkey[i];
            const half_t* src = Kg + (long long)key * p.ldk + lkoff[i];
            glds16((MASKED && key >= p.Lk) ? zp : src, Kb + (unsigned)(i * 512) * 16u);
        }
#pragma unroll
        for (int i = 0; i < LIT; ++i) {
            const int key = key0 + (i * 512 + tid) / DCH;
            const half_t* src = Vg + (long long)key * p.ldv + lvoff[i];
            glds16((MASKED && key >= p.Lk) ? zp : src, Vb + (unsigned)(i * 512) * 16u);
        }
    };
    // fragment offsets (halfs).  K: row r, chunk ((2 (ks & 3) + hh) ^ (r & 7)) + 8 (ks >> 2) + (DCH / 2) ch.
    int koff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) koff[j] = (r * DCH + ((2 * j + hh) ^ (r & 7)) + (DCH / 2) * ch) << 3;
    // V (see flash_attn2_kernel's VROW comment): lane (r, hh) = 16-lane group (r >> 4) + 2 hh; lane 4 tq + tp addresses row tq, channels 4 tp ..
    int voff[4];
    {
        const int tq = (r >> 2) & 3, tp = r & 3, g1 = (r >> 4) & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            voff[j] = TILE_H + (8 * hh + tq) * D + (D / 2) * ch + ((j ^ tq) << 5) + ((2 * g1 + (tp >> 1)) << 3) + 4 * (tp & 1);
    }
    float* xmine = reinterpret_cast<float*>(smem + 4 * TILE_H) + wid * 1024 + lane * 4;
    const float* xpeer = reinterpret_cast<const float*>(smem + 4 * TILE_H) + (wid ^ 4) * 1024 + lane * 4;
    f32x16 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    constexpr float TAU = 8.0f;
    float m_ref = 0.f, l_run = 0.f;
    const int ntiles = (p.Lk + KT - 1) / KT;
    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): retire the Q loads here (see flash_attn2_kernel)
    issue(0, 0);
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();      // tile t has landed for everyone; tile t-1's buffers and exchange slots are free
        if (t + 1 < ntiles) issue(key0 + KT, (t + 1) & 1);
        const half_t* T = smem + (t & 1) * 2 * TILE_H;
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const half8 kf = as_half8(ld16(T + koff[ks & 3] + 64 * (ks >> 2)));
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(xmine + g * 256) = (f32x4){s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // the pair's partial scores are in LDS
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(xpeer + g * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[4 * g + e] += x[e];
        }
        if (MASKED) {
            if (p.causal || key0 + KT > p.Lk) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * hh;
                    const int key = key0 + ((i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1));
                    if (key >= p.Lk || (p.causal && key > qrow)) s[e] = -INFINITY;
                }
            }
        }
        float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int e = 3; e + 1 < 16; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);
        mx = fmaxf(mx, s[15]);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float over = mx - m_ref;
        if (t == 0 || __any(over > TAU)) {
            const float delta = t == 0 ? (mx == -INFINITY ? 0.f : over) : (over > TAU ? over : 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            m_ref += delta;
            l_run *= alpha;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[tt][e] *= alpha;
        }
        float psum = 0.f;
        half8 pf[2];
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const float p0 = __builtin_amdgcn_exp2f(s[e] - m_ref);
            const float p1 = __builtin_amdgcn_exp2f(s[e + 1] - m_ref);
            psum += p0 + p1;
            const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(p0, p1));
            pf[e >> 3][e & 7] = h2[0];
            pf[e >> 3][(e & 7) + 1] = h2[1];
        }
        l_run += psum;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const half_t* vsrc = T + voff[tt & 3] + 128 * (tt >> 2) + 16 * k2 * D;
                const half4v lo = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vsrc)));
                const half4v hi = __builtin_bit_cast(half4v, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vsrc + 4 * D)));
                const half8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[k2], o[tt], 0, 0, 0);
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < p.Lq) {
        half_t* Og = p.O + (long long)b * p.sO + (long long)qrow * p.ldo + (D / 2) * ch;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[tt][4 * g + e] * inv);
                *reinterpret_cast<half4*>(Og + tt * 32 + 8 * g + 4 * hh) = h;
            }
    }
}


thread_local const char* t_last_attn_kernel = "";

template <int DK, bool VROW>
void launch_attn(const AttnParams& p, hipStream_t s) {
    const int nblk = ((p.Lq + 127) / 128) * p.H * p.B;
    // min waves per SIMD handed to __launch_bounds__: with it hipcc keeps the MFMA accumulators in VGPRs (no v_accvgpr
    // copies around the softmax) and fits 3-4 waves per SIMD for the small heads: +25 % at d=40 (profiles/README.md).
    constexpr int AUTO_WPS = DK <= 5 ? 3 : 2;
    // 8-wave workgroups (256 queries share a K / V^T tile: half the LDS-DMA and barrier work per query) once there
    // are enough 256-query blocks to fill the chip at the same waves/SIMD; 4-wave workgroups otherwise.
    const bool masked = p.causal || (p.Lk % 64) != 0;
    const long long nblk8 = (long long)((p.Lq + 255) / 256) * p.H * p.B;
    // measured (tools/attn_micro.py): d=40 L=4096 -1.6 %, L=16384 -5 %; d=80 L=1024 +1 % (142 VGPRs: one workgroup per CU)
    const bool big = DK <= 4 && p.Lq >= 2048 && nblk8 >= 512;
    static const std::string pre = "flash_attn2_kernel<" + std::to_string(DK) + (VROW ? ",rowV" : "");
    static const std::string names[4] = {pre + ",plain,4>", pre + ",masked,4>", pre + ",plain,8>", pre + ",masked,8>"};
    t_last_attn_kernel = names[(big ? 2 : 0) + (masked ? 1 : 0)].c_str();
    if (big) {
        constexpr int W8 = DK <= 4 ? 4 : 2;   // waves per SIMD the register budget is cut for (two or one workgroup per CU)
        if (masked) hipLaunchKernelGGL((flash_attn2_kernel<DK, true, 8, W8, VROW>), dim3((unsigned)nblk8), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((flash_attn2_kernel<DK, false, 8, W8, VROW>), dim3((unsigned)nblk8), dim3(512), 0, s, p);
    } else {
        if (masked) hipLaunchKernelGGL((flash_attn2_kernel<DK, true, 4, AUTO_WPS, VROW>), dim3(nblk), dim3(AT_THREADS), 0, s, p);
        else hipLaunchKernelGGL((flash_attn2_kernel<DK, false, 4, AUTO_WPS, VROW>), dim3(nblk), dim3(AT_THREADS), 0, s, p);
    }
}

template <int D>
void launch_attn512(const AttnParams& p, hipStream_t s) {
    const int nblk = ((p.Lq + 127) / 128) * p.B;
    const bool masked = p.causal || (p.Lk % 32) != 0;
    static const std::string names[2] = {"flash_attn512_kernel<" + std::to_string(D) + ",plain>", "flash_attn512_kernel<" + std::to_string(D) + ",masked>"};
    t_last_attn_kernel = names[masked ? 1 : 0].c_str();
    if (masked) hipLaunchKernelGGL((flash_attn512_kernel<D, true>), dim3(nblk), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((flash_attn512_kernel<D, false>), dim3(nblk), dim3(512), 0, s, p);
}

}  // namespace

const char* attention_last_kernel_name() { return t_last_attn_kernel; }

int attention_launch(const AttnParams& p, hipStream_t stream) {
    if (p.Q == nullptr || p.K == nullptr || (p.Vt == nullptr) == (p.V == nullptr) || p.O == nullptr) return LD_ERR_ARG;   // exactly one of V^T / V
    if (p.B <= 0 || p.H <= 0 || p.Lq <= 0 || p.Lk <= 0) return LD_ERR_SHAPE;
    // d = 512 / 256: the VAE's single head, row-major V only (flash_attn512_kernel)
    if (p.d % 8 || p.d <= 0 || (p.d > 160 && !((p.d == 512 || p.d == 256) && p.H == 1 && p.V != nullptr))) return LD_ERR_SHAPE;
    // 16-byte row copies: every row start must be 16-byte aligned; V^T rows are read in 8-key chunks, so the
    // V^T buffer must be allocated (and zero-padded) to a multiple of 8 keys per row
    if ((p.ldq & 7) || (p.ldk & 7) || (p.ldo & 3) || (p.sQ & 7) || (p.sK & 7) || (p.sV & 7)) return LD_ERR_SHAPE;
    // a row holds all H heads side by side (head h at columns h d .. h d + d - 1): a shorter row stride would make rows overlap / run off the end
    const long long hd = (long long)p.H * p.d;
    if (p.ldq < hd || p.ldk < hd || p.ldo < hd) return LD_ERR_SHAPE;
    // batch strides of several images: an image's rows must not overlap the next image's
    if (p.B > 1 && (p.sQ < (long long)(p.Lq - 1) * p.ldq + hd || p.sK < (long long)(p.Lk - 1) * p.ldk + hd || p.sO < (long long)(p.Lq - 1) * p.ldo + hd)) return LD_ERR_SHAPE;
    const int dk = (p.d + 15) / 16;
    if (p.V != nullptr) {   // row-major V (the UNet's fused q|k|v projection)
        if ((p.ldv & 7) || p.ldv < hd) return LD_ERR_SHAPE;
        if (p.B > 1 && p.sV < (long long)(p.Lk - 1) * p.ldv + hd) return LD_ERR_SHAPE;
        if (p.d == 512 || p.d == 256) {
            if (p.d == 512) launch_attn512<512>(p, stream);
            else launch_attn512<256>(p, stream);
            return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
        }
        switch (dk) {
            case 1: launch_attn<1, true>(p, stream); break;
            case 2: launch_attn<2, true>(p, stream); break;
            case 3: launch_attn<3, true>(p, stream); break;
            case 4: launch_attn<4, true>(p, stream); break;
            case 5: launch_attn<5, true>(p, stream); break;
            case 6: launch_attn<6, true>(p, stream); break;
            case 8: launch_attn<8, true>(p, stream); break;
            case 10: launch_attn<10, true>(p, stream); break;
            default: return LD_ERR_SHAPE;
        }
        return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
    }
    if ((p.ldvt & 7) || p.ldvt < ((p.Lk + 7) & ~7)) return LD_ERR_SHAPE;
    switch (dk) {
        case 1: launch_attn<1, false>(p, stream); break;
        case 2: launch_attn<2, false>(p, stream); break;
        case 3: launch_attn<3, false>(p, stream); break;
        case 4: launch_attn<4, false>(p, stream); break;
        case 5: launch_attn<5, false>(p, stream); break;
        case 6: launch_attn<6, false>(p, stream); break;
        case 8: launch_attn<8, false>(p, stream); break;
        case 10: launch_attn<10, false>(p, stream); break;
        default: return LD_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}
