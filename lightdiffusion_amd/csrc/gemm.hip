// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//
// Tile: BM x BN x 64, 4 waves (2x2), each wave (BM/2)x(BN/2) as 16x16x32 f16 MFMA tiles.
// Staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of slabs, ONE barrier per 64-wide K slab.
// LDS tiles are [rows][64 halfs] with the 16-byte chunk index XOR-swizzled by (row & 7) on the DMA source address
// and on the fragment read: ds_read_b128 fragment reads are bank-conflict free (cdna guide T2, rule 21).
// MFMA operands are swapped (a := W fragment, b := A fragment) so that each lane ends up with 4 consecutive
// output channels of one output row -> 8-byte LDS writes / 16-byte split-K stores in the epilogue.
// Epilogue: accumulators -> fp16 tile in LDS -> row-wise 16-byte coalesced stores with the fused bias /
// time-embedding broadcast / SiLU / GEGLU / residual.
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>
#include <type_traits>

#include "gemm.h"
// Ablation build behind profiles/README.md: -DLD_DBG=6 makes every workgroup stream the SAME tile (all loads hit cache; wrong results,
// timing only).  Never part of the shipped library.
#if !defined(LD_AB_BUILD) || !defined(LD_DBG)   // (A/B build only: the shipped library ignores the macro)
#undef LD_DBG
#define LD_DBG 0
#endif

namespace {

constexpr int BK = 64;
constexpr int NT = 256;

__device__ __forceinline__ void epilogue_store8(const GemmParams& p, int z, int m, int n_out, int n_bias, float (&v)[8]) {
    // v already holds alpha*acc (and, for GEGLU, the gated product with biases applied).  The operands are requested together, then
    // consumed: one memory latency instead of one per operand.
    const bool hb = p.bias_n != nullptr && p.act != 2, hv = p.rowvec != nullptr, hr = p.R != nullptr;
    const uint4 rb = hb ? ld16(p.bias_n + n_bias) : zero16();
    const uint4 rv = hv ? ld16(p.rowvec + (long long)(m / p.rows_per_vec) * p.ldrv + n_out) : zero16();
    const uint4 rr = hr ? ld16(p.R + (long long)z * p.sR + (long long)m * p.ldr + n_out) : zero16();
    const float bm = p.bias_m != nullptr ? (float)p.bias_m[m] : 0.f;
    float b[8], e[8], r[8];
    unpack8(rb, b);
    unpack8(rv, e);
    unpack8(rr, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] + b[j] + bm + e[j];
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
    } else if (p.act == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = quick_gelu_f(v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += r[j];
    st16(p.C + (long long)z * p.sC + (long long)m * p.ldc + n_out, pack8(v));
}

// Tile epilogue: the fp16 C tile sits in LDS; every thread owns EIT 16-byte chunks of it.  All global operands of the
// fused epilogue (bias, time-embedding row vector, residual) are loaded FIRST for every chunk, then consumed: the
// loads overlap each other instead of paying one full memory latency per chunk (the accumulators are dead here, so
// the registers are free).
// Round 5 (tools/gemm3_abl.py, profiles/r05_gemm3_ablations.txt: at 16384 x 640 x 640 the tile epilogue took 8.7 of 23.5 us, 4.6 of them with
// neither residual loads nor stores): the epilogue no longer starts a memory round trip of its own.  `bias_s` = this tile's BN bias halfs,
// staged in LDS by the kernel's prologue (zeros when there is no bias); `pre` = the residual chunks of the first group of the plain interior
// path, requested by the caller BEFORE the tile is staged in LDS (epi_prefetch_residual); the second group's are requested before the first
// group is consumed.
template <int BM, int BN>
struct EpiPre {
    static constexpr int CPR = BN / 8, EIT = (BM * CPR + NT - 1) / NT, GRP = EIT > 5 ? (EIT + 1) / 2 : EIT;
    uint4 r[GRP];
    bool fast;        // the plain interior path runs (workgroup-uniform)
    bool bias_done;   // the bias is already in the staged tile (accumulator start value / LayerNorm-fold finish): the epilogue adds none
};
template <int BM, int BN>
__device__ __forceinline__ EpiPre<BM, BN> epi_prefetch_residual(const GemmParams& p, int z, int m0, int n0, int tid, bool bias_done) {
    EpiPre<BM, BN> e;
    e.bias_done = bias_done;
    constexpr int CPR = EpiPre<BM, BN>::CPR, GRP = EpiPre<BM, BN>::GRP;
    e.fast = (BM * CPR) % NT == 0 && m0 + BM <= p.M && n0 + BN <= p.N && p.act == 0 && p.bias_m == nullptr && bias_done;
#pragma unroll
    for (int k = 0; k < GRP; ++k) e.r[k] = zero16();
#ifdef LD_AB_BUILD
    if (p.dbg & 32) return e;
#endif
    if (e.fast && p.R != nullptr) {
        const half_t* Rb = p.R + (long long)z * p.sR + (long long)m0 * p.ldr + n0;
#pragma unroll
        for (int k = 0; k < GRP; ++k) {
            const int q = tid + k * NT;
            const int row = q / CPR, cc = q - row * CPR;
            e.r[k] = ld16(Rb + (long long)row * p.ldr + cc * 8);
        }
    }
    return e;
}

template <int BM, int BN>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, const half_t* Cs, int z, int m0, int n0, int tid, float* lds_scratch, const half_t* bias_s,
                                              const EpiPre<BM, BN>& pre) {
    constexpr int CLD = BN + 8;
    if (p.act == 2) {
        constexpr int CPR = BN / 16;
        constexpr int EIT = (BM * CPR + NT - 1) / NT;
        if ((BM * CPR) % NT == 0 && m0 + BM <= p.M && n0 + BN <= p.N && pre.bias_done) {
            // interior tile: branch-free, batched reads, stage-by-stage GELUs (common.h) on the packed value chunk; the value / gate biases
            // are in the staged tile already (accumulator start value / LayerNorm-fold finish), the residual is a packed fp16 add
            half_t* Cb = p.C + (long long)z * p.sC + (long long)m0 * p.ldc + n0 / 2;
            const bool hr = p.R != nullptr;
            const half_t* Rb = hr ? p.R + (long long)z * p.sR + (long long)m0 * p.ldr + n0 / 2 : nullptr;
            uint4 rres[EIT], ca[EIT], cg[EIT];
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int q = tid + it * NT;
                const int row = q / CPR, cc = q - row * CPR;
                rres[it] = hr ? ld16(Rb + (long long)row * p.ldr + cc * 8) : zero16();
                ca[it] = ld16(Cs + row * CLD + cc * 8);
                cg[it] = ld16(Cs + row * CLD + BN / 2 + cc * 8);
            }
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int q = tid + it * NT;
                const int row = q / CPR, cc = q - row * CPR;
                float g[8];
                unpack8(cg[it], g);
                const unsigned aw[4] = {ca[it].x, ca[it].y, ca[it].z, ca[it].w};
                const f32x2 gp[4] = {{g[0], g[1]}, {g[2], g[3]}, {g[4], g[5]}, {g[6], g[7]}};
                unsigned ow[4];
                geglu8_staged(aw, gp, ow);
                uint4 packed = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                if (hr) packed = add8h(packed, rres[it]);
                st16(Cb + (long long)row * p.ldc + cc * 8, packed);
            }
            return;
        }
        uint4 rba[EIT], rbg[EIT], rres[EIT];
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int q = tid + it * NT;
            const int row = q / CPR, cc = q - row * CPR;
            const int m = m0 + row, nv = n0 + cc * 8, ng = nv + BN / 2;
            const bool ok = q < BM * CPR && m < p.M && ng < p.N;
            rba[it] = (ok && !pre.bias_done) ? ld16(p.bias_n + nv) : zero16();
            rbg[it] = (ok && !pre.bias_done) ? ld16(p.bias_n + ng) : zero16();
            rres[it] = (ok && p.R != nullptr) ? ld16(p.R + (long long)z * p.sR + (long long)m * p.ldr + n0 / 2 + cc * 8) : zero16();
        }
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int q = tid + it * NT;
            const int row = q / CPR, cc = q - row * CPR;
            const int m = m0 + row, ng = n0 + cc * 8 + BN / 2;
            if (q < BM * CPR && m < p.M && ng < p.N) {
                float a[8], g[8], ba[8], bg[8], r[8];
                unpack8(ld16(Cs + row * CLD + cc * 8), a);
                unpack8(ld16(Cs + row * CLD + BN / 2 + cc * 8), g);
                unpack8(rba[it], ba);
                unpack8(rbg[it], bg);
                unpack8(rres[it], r);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (a[j] + ba[j]) * gelu_f(g[j] + bg[j]) + r[j];
                st16(p.C + (long long)z * p.sC + (long long)m * p.ldc + n0 / 2 + cc * 8, pack8(a));
            }
        }
    } else {
        constexpr int CPR = BN / 8;
        constexpr int EIT = (BM * CPR + NT - 1) / NT;
        constexpr int GRP = EIT > 5 ? (EIT + 1) / 2 : EIT;   // two passes for the big tiles: bounds the live registers
#ifdef LD_AB_BUILD
        const bool hb = p.bias_n != nullptr && !pre.bias_done, hv = p.rowvec != nullptr, hr = p.R != nullptr && !(p.dbg & 32);   // ablations (tools/gemm3_abl.py): 32 no residual loads, 64 no output stores
        const bool no_st = (p.dbg & 64) != 0;
#else
        const bool hb = p.bias_n != nullptr && !pre.bias_done, hv = p.rowvec != nullptr, hr = p.R != nullptr;
        constexpr bool no_st = false;
#endif
        // Interior tiles without an activation: a branch-free path in PACKED fp16 (round 5; tools/gemm3_abl.py: with neither residual loads
        // nor stores the fp32 form of this path still took 4.6 of 23.5 us at 16384 x 640 x 640 — ~70 vector instructions per 16-byte chunk,
        // two workgroups per CU).  The bias is in the staged tile already (accumulator start value, or the LayerNorm-fold finish), so a chunk
        // is: tile chunk (+ time-embedding row) (+ residual) by v_pk_add_f16 — the sum of two fp16 values is exact in fp32, so ONE packed add
        // rounds exactly like the fp32 form did; a chunk that takes both the row vector and the residual is rounded once more (two packed
        // adds: round(round(tile + row) + residual), on top of the one rounding of acc * alpha + bias) — and the LayerNorm-fold row
        // statistics by v_dot2_f32_f16 on the packed result.
        if (pre.fast) {
            static_assert(GRP == EpiPre<BM, BN>::GRP, "group size");
            half_t* Cb = p.C + (long long)z * p.sC + (long long)m0 * p.ldc + n0;
            const half_t* Rb = hr ? p.R + (long long)z * p.sR + (long long)m0 * p.ldr + n0 : nullptr;
            uint4 rnext[GRP];                                   // the residual chunks of the group after the one being consumed
#pragma unroll
            for (int k = 0; k < GRP; ++k) rnext[k] = pre.r[k];
#pragma unroll
            for (int g0 = 0; g0 < EIT; g0 += GRP) {
                uint4 rv[GRP], rres[GRP], cv[GRP];
#pragma unroll
                for (int k = 0; k < GRP; ++k) rres[k] = rnext[k];
#pragma unroll
                for (int k = 0; k < GRP; ++k) {                 // next group's residual: in flight while this group is consumed
                    if (g0 + GRP + k >= EIT) continue;
                    const int q = tid + (g0 + GRP + k) * NT;
                    const int row = q / CPR, cc = q - row * CPR;
                    rnext[k] = hr ? ld16(Rb + (long long)row * p.ldr + cc * 8) : zero16();
                }
#pragma unroll
                for (int k = 0; k < GRP; ++k) {
                    if (g0 + k >= EIT) continue;
                    const int q = tid + (g0 + k) * NT;
                    const int row = q / CPR, cc = q - row * CPR;
                    rv[k] = hv ? ld16(p.rowvec + (long long)((m0 + row) / p.rows_per_vec) * p.ldrv + n0 + cc * 8) : zero16();
                    cv[k] = ld16(Cs + row * CLD + cc * 8);
                }
#pragma unroll
                for (int k = 0; k < GRP; ++k) {
                    if (g0 + k >= EIT) continue;
                    const int q = tid + (g0 + k) * NT;
                    const int row = q / CPR, cc = q - row * CPR;
                    uint4 packed = cv[k];
                    if (hv) packed = add8h(packed, rv[k]);
                    if (hr) packed = add8h(packed, rres[k]);
                    if (!no_st) st16(Cb + (long long)row * p.ldc + cc * 8, packed);
                    if (lds_scratch != nullptr) {   // LN-fold producer: row statistics of what was actually stored (the fp16 values)
                        const half2v one2 = {(half_t)1.f, (half_t)1.f};
                        const half2v h0 = __builtin_bit_cast(half2v, packed.x), h1 = __builtin_bit_cast(half2v, packed.y);
                        const half2v h2 = __builtin_bit_cast(half2v, packed.z), h3 = __builtin_bit_cast(half2v, packed.w);
                        float s1 = __builtin_amdgcn_fdot2(h1, one2, __builtin_amdgcn_fdot2(h0, one2, 0.f, false), false);
                        float s2 = __builtin_amdgcn_fdot2(h1, h1, __builtin_amdgcn_fdot2(h0, h0, 0.f, false), false);
                        s1 = __builtin_amdgcn_fdot2(h3, one2, __builtin_amdgcn_fdot2(h2, one2, s1, false), false);
                        s2 = __builtin_amdgcn_fdot2(h3, h3, __builtin_amdgcn_fdot2(h2, h2, s2, false), false);
                        *reinterpret_cast<float2*>(lds_scratch + q * 2) = make_float2(s1, s2);
                    }
                }
            }
            if (lds_scratch != nullptr) {   // one owner per row sums its CPR chunk partials in chunk order (bitwise reproducible)
                __syncthreads();
                if (tid < BM) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
                    for (int c = 0; c < CPR; ++c) {
                        const float2 t = *reinterpret_cast<const float2*>(lds_scratch + (tid * CPR + c) * 2);
                        s1 += t.x;
                        s2 += t.y;
                    }
                    *reinterpret_cast<float2*>(p.stat_out + ((long long)(n0 / BN) * p.M + m0 + tid) * 2) = make_float2(s1, s2);
                }
            }
            return;
        }
#pragma unroll
        for (int g0 = 0; g0 < EIT; g0 += GRP) {
            uint4 rb[GRP], rv[GRP], rres[GRP];
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int q = tid + (g0 + k) * NT;
                const int row = q / CPR, cc = q - row * CPR;
                const int m = m0 + row, n = n0 + cc * 8;
                const bool ok = (g0 + k) < EIT && q < BM * CPR && m < p.M && n < p.N;
                rb[k] = (ok && hb) ? ld16(p.bias_n + n) : zero16();
                rv[k] = (ok && hv) ? ld16(p.rowvec + (long long)(m / p.rows_per_vec) * p.ldrv + n) : zero16();
                rres[k] = (ok && hr) ? ld16(p.R + (long long)z * p.sR + (long long)m * p.ldr + n) : zero16();
            }
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int q = tid + (g0 + k) * NT;
                const int row = q / CPR, cc = q - row * CPR;
                const int m = m0 + row, n = n0 + cc * 8;
                if ((g0 + k) < EIT && q < BM * CPR && m < p.M && n < p.N) {
                    float v[8], b[8], e[8], r[8];
                    unpack8(ld16(Cs + row * CLD + cc * 8), v);
                    unpack8(rb[k], b);
                    unpack8(rv[k], e);
                    unpack8(rres[k], r);
                    const float bm = p.bias_m != nullptr ? (float)p.bias_m[m] : 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float t = v[j] + b[j] + bm + e[j];
                        if (p.act == 1) t = silu_f(t);
                        else if (p.act == 3) t = quick_gelu_f(t);
                        v[j] = t + r[j];
                    }
                    const uint4 packed = pack8(v);
                    st16(p.C + (long long)z * p.sC + (long long)m * p.ldc + n, packed);
                    if (lds_scratch != nullptr) {   // LN-fold producer: row statistics of what was actually stored (the fp16 values)
                        float f[8];
                        unpack8(packed, f);
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            s1 += f[j];
                            s2 += f[j] * f[j];
                        }
                        lds_scratch[q * 2] = s1;
                        lds_scratch[q * 2 + 1] = s2;
                    }
                } else if (lds_scratch != nullptr && (g0 + k) < EIT && q < BM * CPR) {
                    lds_scratch[q * 2] = 0.f;
                    lds_scratch[q * 2 + 1] = 0.f;
                }
            }
        }
        if (lds_scratch != nullptr) {   // one owner per row sums its CPR chunk partials in chunk order (bitwise reproducible)
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                float s1 = 0.f, s2 = 0.f;
                for (int c = 0; c < CPR; ++c) {
                    s1 += lds_scratch[(tid * CPR + c) * 2];
                    s2 += lds_scratch[(tid * CPR + c) * 2 + 1];
                }
                float* o = p.stat_out + ((long long)(n0 / BN) * p.M + m0 + tid) * 2;
                o[0] = s1;
                o[1] = s2;
            }
        }
    }
}

// ---- LN fold, consumer side (v3 / v4).  ln_prepare: one thread per LN row of the tile finishes (mu, rstd) from the producer's
// per-N-tile partials, in part order, into LDS; ln_apply: acc <- rstd * (acc - mu * wsum) in fp32, before the tile is rounded to fp16.
// (sum, sum of squares) of one row over the producer's parts, in part order; the loads go out four parts at a time (one part after the
// other is one L2 round trip each — 8 in a row at C = 1280 — in the prologue of every consumer launch)
__device__ __forceinline__ void ln_sum_parts(const float* q, long long stride, int parts, float& s1, float& s2) {
    int t = 0;
    for (; t + 4 <= parts; t += 4) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2*>(q + (t + u) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s1 += v[u].x;
            s2 += v[u].y;
        }
    }
    for (; t < parts; ++t) {
        const float2 v = *reinterpret_cast<const float2*>(q + t * stride);
        s1 += v.x;
        s2 += v.y;
    }
}

template <int BM, int BN>
__device__ __forceinline__ void ln_prepare(const GemmParams& p, float* ln_mu, float* ln_rs, int z, int m0, int n0, int tid) {
    const int cnt = p.ln_swapped ? BN : BM;
    if (tid >= cnt) return;
    const bool ok = p.ln_swapped ? (n0 + tid < p.n_valid) : (m0 + tid < p.M);
    const long long row = p.ln_swapped ? (long long)z * p.ln_zrows + n0 + tid : (long long)m0 + tid;
    float s1 = 0.f, s2 = 0.f;
    if (ok) ln_sum_parts(p.ln_stat + row * 2, (long long)p.ln_rows * 2, p.ln_parts, s1, s2);
    const float mu = s1 * p.ln_inv_c;
    ln_mu[tid] = mu;
    // rows / columns beyond the problem get rstd = 0: their (never stored, or padding) outputs stay finite — a V^T padding column
    // scaled by rsqrt(eps) could overflow fp16 to inf, and the attention kernel multiplies it by P = 0
    ln_rs[tid] = ok ? rsqrtf(fmaxf(s2 * p.ln_inv_c - mu * mu, 0.f) + p.ln_eps) : 0.f;
}

template <int TM, int TN>
__device__ __forceinline__ void ln_apply(const GemmParams& p, f32x4 (&acc)[TM][TN], const float* ln_mu, const float* ln_rs, int m0, int n0, int wm0,
                                         int wn0, int fr, int fq, const float* wsum_s, const half_t* bias_s, bool add_bias) {
    if (!p.ln_swapped) {
        f32x4 ws[TN], bj[TN];  // this tile's row sums and (add_bias: act == 0, the epilogue then adds none) its bias / alpha, staged in LDS by the kernel's prologue
        const float inv_alpha = 1.0f / p.alpha;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            ws[j] = *reinterpret_cast<const f32x4*>(wsum_s + wn0 + j * 16 + fq * 4);
            const half4 bh = *reinterpret_cast<const half4*>(bias_s + wn0 + j * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) bj[j][r] = add_bias ? (float)bh[r] * inv_alpha : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float mu = ln_mu[wm0 + i * 16 + fr], rs = ln_rs[wm0 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (acc[i][j] - mu * ws[j]) * rs + bj[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 16 + fr;
            const float wsm = m < p.M ? p.ln_wsum[m] : 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 mu4 = *reinterpret_cast<const f32x4*>(ln_mu + wn0 + j * 16 + fq * 4);
                const f32x4 rs4 = *reinterpret_cast<const f32x4*>(ln_rs + wn0 + j * 16 + fq * 4);
                acc[i][j] = (acc[i][j] - mu4 * wsm) * rs4;
            }
        }
    }
}

__device__ uint4 g_zero_row[4096];  // 64 KB of zeros: conv taps outside the image read it, stepped through like real data (one
                                    // tap's channel run at a time, so it only has to cover max(C1, C2) <= 32768 halfs)

// Prologue staging of a tile's per-column epilogue operands into LDS: bias_s[BN] halfs (zeros without a bias) and, for a LayerNorm-fold
// consumer, wsum_s[BN] floats (zeros beyond N).  epi_stage_load issues the two loads as untracked asm (always a load, from a page of zeros
// where there is nothing to fetch) BEFORE the first LDS-DMA of the wave, so they are its oldest vector-memory operations;
// EPI_STAGE_WAIT(KEEP, ..) waits with a COUNTED vmcnt that leaves the KEEP LDS-DMA instructions issued since in flight (KEEP = 0 where the
// wave issues none, or fewer than the full prologue) and names the destination registers as operands of that wait (DESIGN "hipcc traps" (c));
// epi_stage_store writes them to LDS; the slab loop's barriers publish them long before the epilogue reads them.
// (the loaded values live in two f32x4 locals of the KERNEL — b: 8 bias halfs as a bit pattern, w: 4 row sums — so that the counted wait
// can name them as read-write operands: nothing that copies or spills them can be scheduled between a load and the wait)
template <int BN>
__device__ __forceinline__ void epi_stage_load(const GemmParams& p, int n0, int t, f32x4& b, f32x4& w) {   // t: thread index inside the loading role (>= BN / 4 threads)
    const char* zp = reinterpret_cast<const char*>(g_zero_row);
    const char* bp = (t < BN / 8 && p.bias_n != nullptr && n0 + t * 8 < p.N) ? reinterpret_cast<const char*>(p.bias_n + n0 + t * 8) : zp;
    const char* wp = (t < BN / 4 && p.ln_wsum != nullptr && !p.ln_swapped && n0 + t * 4 < p.N) ? reinterpret_cast<const char*>(p.ln_wsum + n0 + t * 4) : zp;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(b) : "v"(bp) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(w) : "v"(wp) : "memory");
}
#define EPI_STAGE_WAIT(KEEP, b, w) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b), "+v"(w) : "n"(KEEP) : "memory")
template <int BN>
__device__ __forceinline__ void epi_stage_store(const f32x4& b, const f32x4& w, half_t* bias_s, float* wsum_s, int t) {   // (behind EPI_STAGE_WAIT)
    if (t < BN / 8) *reinterpret_cast<f32x4*>(bias_s + t * 8) = b;
    if (t < BN / 4) *reinterpret_cast<f32x4*>(wsum_s + t * 4) = w;
}


// =====================================================================================================================
// v3: 64-wide K slabs, a TWO-stage LDS-DMA ring (2 x (BM+BN) x 128 B <= 73.7 KB) and two workgroups per CU.
// One barrier per slab = per 2 k-steps (40 MFMAs per wave at 128x160), fragments double-buffered at k-step granularity,
// 128-byte LDS rows with the chunk ^ (row & 7) swizzle (conflict-free 16x16x32 fragment reads).
// The slab loop is written to carry (almost) no vector-ALU work, because on this chip a VALU instruction of one wave and
// an MFMA of the other wave on the same SIMD do not overlap at equal priority (tools/micro/coexec.hip: sum, not max):
//  * B (and A of a plain GEMM) are fetched as  scalar base + per-lane 32-bit offset  — the per-slab advance is two SALU adds
//    instead of one 64-bit VALU add per load; rows beyond M / n_valid are clamped to a valid row (their outputs are never
//    stored, resp. are the don't-care padding columns of V^T), so the steady state has no select either;
//  * conv A keeps per-lane pointers (taps outside the image read a run of zeros that is stepped like real data);
//  * every fragment read is  base VGPR + immediate offset;  the two base VGPRs flip between the stages by one add each.
// A K that is not a multiple of 64 takes a select-per-load slow path on its last slab only (wave-uniform branch).
// =====================================================================================================================
__device__ __forceinline__ void glds16s(unsigned voff, const half_t* sbase, unsigned lds_base) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_base)
        : "memory");
}

template <int BM, int BN, bool CONV, int NST>   // NST = 2: two workgroups per CU;  NST = 4: one workgroup, three slabs in flight
__global__ __launch_bounds__(NT, NST == 2 ? 2 : 1) void gemm3_kernel(const GemmParams p) {
    constexpr int BK3 = 64, PF = NST - 1;
    static_assert(NST == 2 || NST == 4, "ring depth");
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_CH = BM * 8, B_CH = BN * 8;
    constexpr int A_IT = A_CH / NT;                             // 4 (BM=128) or 2 (BM=64)
    constexpr int B_IT = B_CH / NT;                             // 5 (BN=160) or 4 (BN=128): exact, no tail
    static_assert(B_CH % NT == 0 && A_CH % NT == 0, "whole instructions per wave");
    constexpr int STAGE = (BM + BN) * BK3;
    constexpr int CLD = BN + 8;
    static_assert(BM * CLD <= NST * STAGE, "epilogue tile must fit in the ring");
    static_assert((BM * CLD * 2 + 15) / 16 * 16 + BM * (BN / 8) * 8 <= NST * STAGE * 2, "LN-fold row statistics must fit behind the epilogue tile");
    __shared__ __attribute__((aligned(16))) half_t smem[NST * STAGE];
    __shared__ __attribute__((aligned(16))) float ln_mu[BM > BN ? BM : BN], ln_rs[BM > BN ? BM : BN];
    __shared__ __attribute__((aligned(16))) half_t bias_s[BN];
    __shared__ __attribute__((aligned(16))) float wsum_s[BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid >> 1) * WTM, wn0 = (wid & 1) * WTN;
    const int z = blockIdx.z;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    int bid = xcd_remap(blockIdx.x, tiles * splitk);
    const int ks = bid / tiles;
    bid -= ks * tiles;
    const int tn_i = p.m_fastest ? bid / tiles_m : bid % tiles_n;
    const int tm_i = p.m_fastest ? bid % tiles_m : bid / tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int KT = (p.K + BK3 - 1) / BK3;
    const int kt_begin = (int)((long long)ks * KT / splitk), kt_end = (int)((long long)(ks + 1) * KT / splitk);

    const half_t* Ab = p.A + (long long)z * p.sA;
    const half_t* Wb = p.W + (long long)z * p.sW;
    const half_t* zp = reinterpret_cast<const half_t*>(g_zero_row);   // 64 KB of zeros, stepped through like real data
    const int Cin = p.C1 + p.C2;

    // ---- A loader state
    int a_lc[A_IT];
    bool a_ok[A_IT];
    int a_img[A_IT], a_iy0[A_IT], a_ix0[A_IT];
    const half_t* a_ptr[A_IT];     // CONV: per-lane source pointers
    unsigned a_off[A_IT];          // plain GEMM: byte offset from the scalar base a_base
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int q = tid + i * NT;
        const int row = q >> 3;
        a_lc[i] = (q & 7) ^ (row & 7);
#if LD_DBG == 6
        const int m = row;            // every workgroup streams the SAME tile: all loads hit cache (latency probe)
#else
        const int m = m0 + row;
#endif
        a_ok[i] = m < p.M;
        a_img[i] = a_iy0[i] = a_ix0[i] = 0;
        a_ptr[i] = zp;
        a_off[i] = 0;
        if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_img[i] = img;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
        } else {
            a_off[i] = (unsigned)(((long long)(a_ok[i] ? m : p.M - 1) * p.lda + a_lc[i] * 8) * 2);
        }
    }
    const half_t* a_base = Ab + (long long)kt_begin * BK3;   // wave-uniform
    int seg_left = 0;
    auto conv_seek = [&](int k0) {
        // (k0 beyond the taps: the second K segment — the 1x1 skip convolution's raw sources at the output pixel itself, gemm.h S1 / S2)
        const int K9 = p.ksize * p.ksize * Cin;
        const bool skp = k0 >= K9 && p.SC1 > 0;
        const int tap = skp ? 0 : k0 / Cin;
        const int c0 = skp ? k0 - K9 : k0 - tap * Cin;
        const int ky = skp ? p.pad : tap / p.ksize, kx = skp ? p.pad : tap - (tap / p.ksize) * p.ksize;
        const int Ca = skp ? p.SC1 : p.C1, Cb = skp ? p.SC2 : p.C2;
        const bool second = c0 >= Ca;
        const half_t* src = skp ? (second ? p.S2 : p.S1) : (second ? p.A2 : Ab);
        const int Cs = second ? Cb : Ca;
        const int cl = second ? c0 - Ca : c0;
        seg_left = ((second ? Ca + Cb : Ca) - c0) / BK3;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv && tap < p.ksize * p.ksize && c0 < Ca + Cb;
            int sy = iy, sx = ix;
            if (p.Hv == 2 * p.Hs && p.Wv == 2 * p.Ws) {
                sy = iy >> 1;
                sx = ix >> 1;
            } else if (p.Hv != p.Hs || p.Wv != p.Ws) {
                sy = (int)((long long)iy * p.Hs / p.Hv);
                sx = (int)((long long)ix * p.Ws / p.Wv);
            }
            a_ptr[i] = ok ? src + (((long long)a_img[i] * p.Hs + sy) * p.Ws + sx) * Cs + cl + a_lc[i] * 8 : zp + a_lc[i] * 8;
        }
    };
    // ---- B loader state
    unsigned b_off[B_IT];
    int b_lc[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int q = tid + i * NT;
        const int row = q >> 3;
        b_lc[i] = (q & 7) ^ (row & 7);
#if LD_DBG == 6
        const int n = row;
#else
        const int n = n0 + row < p.n_valid ? n0 + row : p.n_valid - 1;
#endif
        b_off[i] = (unsigned)(((long long)n * p.ldw + b_lc[i] * 8) * 2);
    }
    const half_t* b_base = Wb + (long long)kt_begin * BK3;   // wave-uniform

    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    auto issue = [&](int kt, int st) {   // st is a literal at every call site
        if (kt >= kt_end) return;        // the consumer's wait is chosen from the number of slabs really in flight
        const unsigned As = smem_base + (unsigned)(st * STAGE) * 2u + (unsigned)(wid * 64) * 16u;
        const unsigned Bs = As + (unsigned)(BM * BK3) * 2u;
        const int k0 = kt * BK3;
        if (k0 + BK3 <= p.K) {           // steady state: bare DMA issues
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                if (CONV) glds16(a_ptr[i], As + (unsigned)(i * NT) * 16u);
                else glds16s(a_off[i], a_base, As + (unsigned)(i * NT) * 16u);
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) glds16s(b_off[i], b_base, Bs + (unsigned)(i * NT) * 16u);
        } else {                         // ragged last slab of a K that is not a multiple of 64
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const half_t* src = CONV ? a_ptr[i] : reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(a_base) + a_off[i]);
                glds16(k0 + a_lc[i] * 8 < p.K ? src : zp, As + (unsigned)(i * NT) * 16u);
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const half_t* src = reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(b_base) + b_off[i]);
                glds16(k0 + b_lc[i] * 8 < p.K ? src : zp, Bs + (unsigned)(i * NT) * 16u);
            }
        }
        if (CONV) {
            if (--seg_left <= 0) {
                conv_seek(k0 + BK3);
            } else {
#pragma unroll
                for (int i = 0; i < A_IT; ++i) a_ptr[i] += BK3;
            }
        } else {
            a_base += BK3;
        }
        b_base += BK3;
    };
    if (CONV) conv_seek(kt_begin * BK3);

    const int fr = lane & 15, fq = lane >> 4;
    // fragment read bases (halfs, inside the stage being read): row&7 == fr&7 for every fragment row (wm0, wn0, 16*i are
    // multiples of 8), so the swizzled chunk depends on the k-step only and every other term is an immediate offset.
    // rd0 / rd1 = k-step 0 / 1 of the A rows; the B rows sit (BM + wn0 - wm0) rows further.  Both are flipped between the two
    // stages by one add each per slab (the only vector-ALU work of the steady-state loop besides conv A's pointer bumps).
    const half_t* rd0 = smem + (wm0 + fr) * BK3 + ((fq ^ (fr & 7)) << 3);
    const half_t* rd1 = smem + (wm0 + fr) * BK3 + (((4 + fq) ^ (fr & 7)) << 3);
    const int b_rel = (BM + wn0 - wm0) * BK3;   // wave-uniform

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    half8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto read_frags = [&](const half_t* rd, half8 (&fa)[TM], half8 (&fb)[TN]) {
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = as_half8(ld16(rd + b_rel + j * 16 * BK3));
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = as_half8(ld16(rd + i * 16 * BK3));
    };
    auto mma = [&](const half8 (&fa)[TM], const half8 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };

    constexpr int LPT = A_IT + B_IT;   // DMA instructions per wave per slab
    // "slab kt has landed" = at most the loads of the slabs issued after it are still outstanding (they complete in order)
    auto wait_slab = [&](int kt) {
        if (PF == 1) {
            wait_vmcnt<0>();
        } else {
            const int ahead = kt_end - 1 - kt;          // slabs issued after kt (at most PF - 1)
            if (ahead >= PF - 1) wait_vmcnt<LPT*(PF - 1)>();
            else if (ahead == 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
        }
    };
#ifdef LD_AB_BUILD
    if ((p.dbg & 0x700) && NST == 2) {   // experiment (tools/gemm3_abl.py stagger): the second co-resident workgroup of a CU (the second 256 blocks) starts (dbg >> 8) & 7 us late
        if ((int)blockIdx.x >= 256 && (int)blockIdx.x < 512) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            const unsigned long long wait = 100ull * ((p.dbg >> 8) & 7);
            while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
        }
    }
#endif
    f32x4 est_b, est_w;
    epi_stage_load<BN>(p, n0, tid, est_b, est_w);   // (before the first LDS-DMA: see epi_stage_load)
#pragma unroll
    for (int t = 0; t < PF; ++t) issue(kt_begin + t, t);
    // LN fold: finish (mu, rstd) of this tile's LN rows while the first slabs are in flight (the slab loop's barriers publish it)
    if (p.ln_stat != nullptr) ln_prepare<BM, BN>(p, ln_mu, ln_rs, z, m0, n0, tid);
    if (kt_end - kt_begin >= PF) EPI_STAGE_WAIT(PF * (A_IT + B_IT), est_b, est_w);   // (every issue() above went out: A_IT + B_IT instructions each)
    else EPI_STAGE_WAIT(0, est_b, est_w);
    epi_stage_store<BN>(est_b, est_w, bias_s, wsum_s, tid);
    wait_slab(kt_begin);
    __builtin_amdgcn_s_barrier();
    issue(kt_begin + PF, PF);
    // the bias (zeros without one) is the accumulators' START value where the epilogue is a plain or a GEGLU one (no other activation, no
    // split over K, no LayerNorm-fold finish, which adds it itself): the staged tile then holds acc * alpha + bias rounded ONCE, and the
    // epilogue adds none
    const bool bias_acc = splitk == 1 && p.ln_stat == nullptr && (p.act == 0 || p.act == 2);   // (GEGLU: value and gate biases alike, in the tile's column order)
    const bool bias_done = bias_acc || (splitk == 1 && p.ln_stat != nullptr && !p.ln_swapped && (p.act == 0 || p.act == 2));
    if (bias_acc) {
        const float inv_alpha = 1.0f / p.alpha;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const half4 bh = *reinterpret_cast<const half4*>(bias_s + wn0 + j * 16 + fq * 4);
            f32x4 bf;
#pragma unroll
            for (int r = 0; r < 4; ++r) bf[r] = (float)bh[r] * inv_alpha;
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = bf;
        }
    }
    read_frags(rd0, fa0, fb0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int st = 0;
#ifdef LD_AB_BUILD
    // ablations (tools/gemm3_abl.py; wrong results, timing only): 1 no DMA behind the prologue, 2 no fragment reads in the loop, 4 no MFMAs, 8 no epilogue
    const bool no_dma = (p.dbg & 1) != 0, no_rd = (p.dbg & 2) != 0, no_mm = (p.dbg & 4) != 0;
    if (no_rd) read_frags(rd1, fa1, fb1);
#else
    constexpr bool no_dma = false, no_rd = false, no_mm = false;
#endif
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        // k-step 0 of slab kt sits in set 0; fetch k-step 1 under its MFMAs, then (slab kt+1 landed for everyone, stage st
        // free) refill st with slab kt+NST and fetch k-step 0 of slab kt+1 under the k-step-1 MFMAs
        if (!no_rd) read_frags(rd1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mm) mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_slab(kt + 1);
        __builtin_amdgcn_s_barrier();
        if (!no_dma) issue(kt + NST, st);
        const int flip = (st == NST - 1) ? -(NST - 1) * STAGE : STAGE;   // halfs to the next stage of the ring
        rd0 += flip;
        rd1 += flip;
        st = (st + 1) & (NST - 1);
        if (!no_rd) read_frags(rd0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mm) mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#ifdef LD_AB_BUILD
    if (p.dbg & 8) return;
#endif

    if (splitk > 1) {
        float* part = p.partial + (long long)ks * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn0 + j * 16 + fq * 4;
                if (m < p.M && n < p.N) {
                    f32x4 v = acc[i][j];
                    v *= p.alpha;
                    *reinterpret_cast<f32x4*>(part + (long long)m * p.N + n) = v;
                }
            }
        }
        return;
    }
    half_t* Cs = smem;
    const EpiPre<BM, BN> pre = epi_prefetch_residual<BM, BN>(p, z, m0, n0, tid, bias_done);   // in flight while the tile is staged
    if (p.ln_stat != nullptr) ln_apply<TM, TN>(p, acc, ln_mu, ln_rs, m0, n0, wm0, wn0, fr, fq, wsum_s, bias_s, bias_done);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm0 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn0 + j * 16 + fq * 4;
            const f32x4 v = acc[i][j] * p.alpha;
            *reinterpret_cast<uint2*>(Cs + ml * CLD + nl) = make_uint2(pk2h(v[0], v[1]), pk2h(v[2], v[3]));
        }
    }
    __syncthreads();
    float* scratch = p.stat_out != nullptr ? reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + (BM * CLD * 2 + 15) / 16 * 16) : nullptr;
    epilogue_tile<BM, BN>(p, Cs, z, m0, n0, tid, scratch, bias_s, pre);
}

// =====================================================================================================================
// v4: the v3 tile with the roles split over 8 waves ("producer / consumer").  Waves 0-3 only read fragments and issue MFMAs;
// waves 4-7 only issue the LDS-DMA for the slab ring (the loads wave w-4 issues in v3).  Why: an LDS-DMA instruction costs its
// wave 60-180 issue cycles (microarch guide) and a wave issues in order, so in v3 each slab's 9 DMA issues sit in front of the
// same wave's 40 MFMAs — a lone workgroup on a CU spends ~0.75 us per slab, 0.3 us of it in MFMAs (tools/gemm_micro.py small).
// With the split the DMA issue runs on the other wave of each SIMD, and the ring is 4 deep (one workgroup per CU: 147 KB),
// so three slabs are in flight.  One s_barrier per slab joins all 8 waves:
//   consumer kt:  read k-step-1 frags of slab kt | MFMA k-step 0 | lgkmcnt(0) | BARRIER kt | read k-step-0 frags of kt+1 | MFMA k-step 1
//   producer kt:  wait until slab kt+1 has landed (counted vmcnt)             | BARRIER kt | issue slab kt+4 into the stage of kt
// After BARRIER kt slab kt+1 is complete for everyone and nobody reads slab kt's stage any more.
// =====================================================================================================================
// "at most `ahead` slabs' worth of this wave's loads (LPT each) are still outstanding", ahead clamped to MAXA (counted vmcnt needs literals)
template <int LPT, int MAXA>
__device__ __forceinline__ void wait_slabs_ahead(int ahead) {
    if constexpr (MAXA == 0) {
        wait_vmcnt<0>();
    } else {
        if (ahead >= MAXA) wait_vmcnt<MAXA * LPT>();
        else wait_slabs_ahead<LPT, MAXA - 1>(ahead);
    }
}

// NST: ring depth (a power of two; 4 in every shipped instantiation — 8 stages measured +-0, see launch_cfg).  WPS: waves per SIMD the
// register allocation is bounded for — 4 lets TWO workgroups of the 64 x 64 tile share a CU (2 x 64 KB of LDS, 96 KB of slabs in flight)
// where a skinny projection has more tiles than CUs.
template <int BM, int BN, bool CONV, int NST = 4, int WPS = 2>
__global__ __launch_bounds__(2 * NT, WPS) void gemm4_kernel(const GemmParams p) {
    constexpr int BK3 = 64;
    static_assert((NST & (NST - 1)) == 0 && NST >= 4, "ring depth");
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "whole instructions per wave");
    constexpr int LPT = A_IT + B_IT;
    constexpr int STAGE = (BM + BN) * BK3;
    constexpr int CLD = BN + 8;
    static_assert(BM * CLD <= NST * STAGE, "epilogue tile must fit in the ring");
    __shared__ __attribute__((aligned(16))) half_t smem[NST * STAGE];
    __shared__ __attribute__((aligned(16))) float ln_mu[BM > BN ? BM : BN], ln_rs[BM > BN ? BM : BN];
    __shared__ __attribute__((aligned(16))) half_t bias_s[BN];
    __shared__ __attribute__((aligned(16))) float wsum_s[BN];

    const int lane = threadIdx.x & 63;
    const int wid8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wid8 >= 4;
    const int wid = wid8 & 3;
    const int tid = wid * 64 + lane;          // 0..255 inside the role
    const int z = blockIdx.z;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    int bid = xcd_remap(blockIdx.x, tiles * splitk);
    const int ks = bid / tiles;
    bid -= ks * tiles;
    const int tn_i = p.m_fastest ? bid / tiles_m : bid % tiles_n;
    const int tm_i = p.m_fastest ? bid % tiles_m : bid / tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int KT = (p.K + BK3 - 1) / BK3;
    const int kt_begin = (int)((long long)ks * KT / splitk), kt_end = (int)((long long)(ks + 1) * KT / splitk);

    if (producer) {
        // ------------------------------------------------------------------ producer: the v3 loader, nothing else
        const half_t* Ab = p.A + (long long)z * p.sA;
        const half_t* Wb = p.W + (long long)z * p.sW;
        const half_t* zp = reinterpret_cast<const half_t*>(g_zero_row);
        const int Cin = p.C1 + p.C2;
        int a_lc[A_IT];
        bool a_ok[A_IT];
        int a_img[A_IT], a_iy0[A_IT], a_ix0[A_IT];
        const half_t* a_ptr[A_IT];
        unsigned a_off[A_IT];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int q = tid + i * NT;
            const int row = q >> 3;
            a_lc[i] = (q & 7) ^ (row & 7);
            const int m = m0 + row;
            a_ok[i] = m < p.M;
            a_img[i] = a_iy0[i] = a_ix0[i] = 0;
            a_ptr[i] = zp;
            a_off[i] = 0;
            if (CONV) {
                const int hw = p.Ho * p.Wo;
                const int mm = a_ok[i] ? m : 0;
                const int img = mm / hw, rem = mm - img * hw;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                a_img[i] = img;
                a_iy0[i] = oy * p.stride - p.pad;
                a_ix0[i] = ox * p.stride - p.pad;
            } else {
                a_off[i] = (unsigned)(((long long)(a_ok[i] ? m : p.M - 1) * p.lda + a_lc[i] * 8) * 2);
            }
        }
        const half_t* a_base = Ab + (long long)kt_begin * BK3;
        int seg_left = 0;
        auto conv_seek = [&](int k0) {
            // (k0 beyond the taps: the second K segment — the 1x1 skip convolution's raw sources at the output pixel itself, gemm.h S1 / S2)
            const int K9 = p.ksize * p.ksize * Cin;
            const bool skp = k0 >= K9 && p.SC1 > 0;
            const int tap = skp ? 0 : k0 / Cin;
            const int c0 = skp ? k0 - K9 : k0 - tap * Cin;
            const int ky = skp ? p.pad : tap / p.ksize, kx = skp ? p.pad : tap - (tap / p.ksize) * p.ksize;
            const int Ca = skp ? p.SC1 : p.C1, Cb = skp ? p.SC2 : p.C2;
            const bool second = c0 >= Ca;
            const half_t* src = skp ? (second ? p.S2 : p.S1) : (second ? p.A2 : Ab);
            const int Cs = second ? Cb : Ca;
            const int cl = second ? c0 - Ca : c0;
            seg_left = ((second ? Ca + Cb : Ca) - c0) / BK3;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv && tap < p.ksize * p.ksize && c0 < Ca + Cb;
                int sy = iy, sx = ix;
                if (p.Hv == 2 * p.Hs && p.Wv == 2 * p.Ws) {
                    sy = iy >> 1;
                    sx = ix >> 1;
                } else if (p.Hv != p.Hs || p.Wv != p.Ws) {
                    sy = (int)((long long)iy * p.Hs / p.Hv);
                    sx = (int)((long long)ix * p.Ws / p.Wv);
                }
                a_ptr[i] = ok ? src + (((long long)a_img[i] * p.Hs + sy) * p.Ws + sx) * Cs + cl + a_lc[i] * 8 : zp + a_lc[i] * 8;
            }
        };
        unsigned b_off[B_IT];
        int b_lc[B_IT];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int q = tid + i * NT;
            const int row = q >> 3;
            b_lc[i] = (q & 7) ^ (row & 7);
            const int n = n0 + row < p.n_valid ? n0 + row : p.n_valid - 1;
            b_off[i] = (unsigned)(((long long)n * p.ldw + b_lc[i] * 8) * 2);
        }
        const half_t* b_base = Wb + (long long)kt_begin * BK3;
        const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
        auto issue = [&](int kt) {
            if (kt >= kt_end) return;
            const int st = (kt - kt_begin) & (NST - 1);
            const unsigned As = smem_base + (unsigned)(st * STAGE) * 2u + (unsigned)(wid * 64) * 16u;
            const unsigned Bs = As + (unsigned)(BM * BK3) * 2u;
            const int k0 = kt * BK3;
            if (k0 + BK3 <= p.K) {
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    if (CONV) glds16(a_ptr[i], As + (unsigned)(i * NT) * 16u);
                    else glds16s(a_off[i], a_base, As + (unsigned)(i * NT) * 16u);
                }
#pragma unroll
                for (int i = 0; i < B_IT; ++i) glds16s(b_off[i], b_base, Bs + (unsigned)(i * NT) * 16u);
            } else {
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    const half_t* src = CONV ? a_ptr[i] : reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(a_base) + a_off[i]);
                    glds16(k0 + a_lc[i] * 8 < p.K ? src : zp, As + (unsigned)(i * NT) * 16u);
                }
#pragma unroll
                for (int i = 0; i < B_IT; ++i) {
                    const half_t* src = reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(b_base) + b_off[i]);
                    glds16(k0 + b_lc[i] * 8 < p.K ? src : zp, Bs + (unsigned)(i * NT) * 16u);
                }
            }
            if (CONV) {
                if (--seg_left <= 0) {
                    conv_seek(k0 + BK3);
                } else {
#pragma unroll
                    for (int i = 0; i < A_IT; ++i) a_ptr[i] += BK3;
                }
            } else {
                a_base += BK3;
            }
            b_base += BK3;
        };
        // slab kt has landed once at most the loads of the slabs issued after it are outstanding (in-order completion)
        auto wait_slab = [&](int kt, int issued_after) {
            int ahead = kt_end - 1 - kt;
            if (ahead > issued_after) ahead = issued_after;
            wait_slabs_ahead<LPT, NST - 1>(ahead);
        };
        if (CONV) conv_seek(kt_begin * BK3);
#pragma unroll
        for (int t = 0; t < NST; ++t) issue(kt_begin + t);
        wait_slab(kt_begin, NST - 1);
        __builtin_amdgcn_s_barrier();                    // P: slab kt_begin is readable
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            wait_slab(kt + 1, NST - 2);                  // issued so far: up to kt + NST - 1
            __builtin_amdgcn_s_barrier();                // BARRIER kt
#ifdef LD_AB_BUILD
            if (p.dbg & 1) continue;                     // ablation (tools/gemm4_abl.py; wrong results, timing only): no DMA behind the prologue's NST slabs
#endif
            issue(kt + NST);
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                    // tail barrier (pairs with the consumers' before the epilogue)
        return;
    }

    // ---------------------------------------------------------------------- consumers
    {   // this tile's bias / LayerNorm-fold row sums into LDS (EpiStage; the consumers issue no LDS-DMA)
        f32x4 est_b, est_w;
        epi_stage_load<BN>(p, n0, tid, est_b, est_w);
        EPI_STAGE_WAIT(0, est_b, est_w);
        epi_stage_store<BN>(est_b, est_w, bias_s, wsum_s, tid);
    }
    if (p.ln_stat != nullptr) ln_prepare<BM, BN>(p, ln_mu, ln_rs, z, m0, n0, tid);   // (the slab loop's barriers publish it)
    const int wm0 = (wid >> 1) * WTM, wn0 = (wid & 1) * WTN;
    const int fr = lane & 15, fq = lane >> 4;
    const half_t* rd0 = smem + (wm0 + fr) * BK3 + ((fq ^ (fr & 7)) << 3);
    const half_t* rd1 = smem + (wm0 + fr) * BK3 + (((4 + fq) ^ (fr & 7)) << 3);
    const int b_rel = (BM + wn0 - wm0) * BK3;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto read_frags = [&](const half_t* rd, half8 (&fa)[TM], half8 (&fb)[TN]) {
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = as_half8(ld16(rd + b_rel + j * 16 * BK3));
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = as_half8(ld16(rd + i * 16 * BK3));
    };
    auto mma = [&](const half8 (&fa)[TM], const half8 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };
    __builtin_amdgcn_s_barrier();                        // P
    // (bias as the accumulators' start value: see gemm3_kernel)
    const bool bias_acc = splitk == 1 && p.ln_stat == nullptr && (p.act == 0 || p.act == 2);   // (GEGLU: value and gate biases alike, in the tile's column order)
    const bool bias_done = bias_acc || (splitk == 1 && p.ln_stat != nullptr && !p.ln_swapped && (p.act == 0 || p.act == 2));
    if (bias_acc) {
        const float inv_alpha = 1.0f / p.alpha;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const half4 bh = *reinterpret_cast<const half4*>(bias_s + wn0 + j * 16 + fq * 4);
            f32x4 bf;
#pragma unroll
            for (int r = 0; r < 4; ++r) bf[r] = (float)bh[r] * inv_alpha;
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = bf;
        }
    }
    read_frags(rd0, fa0, fb0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int st = 0;
#ifdef LD_AB_BUILD
    const bool no_rd = (p.dbg & 2) != 0, no_mm = (p.dbg & 4) != 0;   // ablations (tools/gemm4_abl.py): no fragment reads in the loop / no MFMAs
    if (no_rd) read_frags(rd1, fa1, fb1);
#else
    constexpr bool no_rd = false, no_mm = false;
#endif
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (!no_rd) read_frags(rd1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mm) mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // BARRIER kt
        const int flip = (st == NST - 1) ? -(NST - 1) * STAGE : STAGE;
        rd0 += flip;
        rd1 += flip;
        st = (st + 1) & (NST - 1);
        if (!no_rd) read_frags(rd0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (!no_mm) mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // tail: the ring is quiet, producers leave
#ifdef LD_AB_BUILD
    if (p.dbg & 8) return;                               // ablation: no epilogue
#endif

    if (splitk > 1) {
        float* part = p.partial + (long long)ks * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn0 + j * 16 + fq * 4;
                if (m < p.M && n < p.N) {
                    f32x4 v = acc[i][j];
                    v *= p.alpha;
                    *reinterpret_cast<f32x4*>(part + (long long)m * p.N + n) = v;
                }
            }
        }
        return;
    }
    half_t* Cs = smem;
    const EpiPre<BM, BN> pre = epi_prefetch_residual<BM, BN>(p, z, m0, n0, tid, bias_done);   // in flight while the tile is staged
    if (p.ln_stat != nullptr) ln_apply<TM, TN>(p, acc, ln_mu, ln_rs, m0, n0, wm0, wn0, fr, fq, wsum_s, bias_s, bias_done);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm0 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn0 + j * 16 + fq * 4;
            const f32x4 v = acc[i][j] * p.alpha;
            *reinterpret_cast<uint2*>(Cs + ml * CLD + nl) = make_uint2(pk2h(v[0], v[1]), pk2h(v[2], v[3]));
        }
    }
    __syncthreads();                                     // consumers only: the producers have exited
    float* scratch = p.stat_out != nullptr ? reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + (BM * CLD * 2 + 15) / 16 * 16) : nullptr;
    epilogue_tile<BM, BN>(p, Cs, z, m0, n0, tid, scratch, bias_s, pre);
}

// =====================================================================================================================
// v5: one 256 x 320 tile per workgroup, 8 waves (4 x 2, wave tile 64 x 160), 32-wide K steps in a 4-stage LDS-DMA ring, and the
// two wave groups (waves 0-3 / 4-7: the two waves of every SIMD) run HALF A STEP APART:
//     group 0:  | read k   | MFMA k   | read k+1 | MFMA k+1 | ...
//     group 1:  | (idle)   | read k   | MFMA k   | read k+1 | ...           ('|' = one s_barrier joining all 8 waves)
// so in every interval one wave of each SIMD issues 40 MFMAs (640 cycles) while its partner issues the 14 fragment reads
// of its next step and its share of the LDS-DMA for the step after next (4-5 one-KB pieces).  Fragments are single-buffered:
// the overlap comes from the partner wave, not from register double-buffering (160 accumulator + 56 fragment VGPRs at two
// waves per SIMD).  Against the 128 x 160 tile of v3 a step moves half the LDS-DMA pieces and 0.35 instead of 0.45
// fragment reads per MFMA, and nothing of the staging sits in front of the issuing wave's own MFMAs.
//   step k lives in stage k % 4; group 0 reads it in interval 2k, group 1 in 2k+1; it is overwritten (step k+4) from
//   interval 2k+2 on (three steps = 110 KB per CU in flight), and every wave waits for its own pieces of step k+1 (counted vmcnt) before the barrier that ends its
//   read phase k — hence before anybody reads step k+1.
// LDS rows are 64 bytes (4 chunks); chunk c of row r sits at physical chunk c ^ g[(r >> 2) & 3], g = {0, 2, 3, 1}: with the
// lane groups ds_read_b128 is served in ({0-3, 12-15, 20-27}, ...) the 16 lanes of a group then hit 16 distinct 16-byte slots.
// Epilogue: per wave, 16-row strips staged two at a time in the (then quiet) ring (rows padded to 328 bytes), 16-byte coalesced
// stores of 320-byte row segments with the same fused bias / row vector / activation / GEGLU / residual / LN-fold math as v3.
// Requirements (gemm_launch checks them): N % 320 == 0, K % 32 == 0, every split-K slice >= 2 steps, no ln_swapped.
// =====================================================================================================================
constexpr int V5_BM = 256, V5_BN = 320, V5_BK = 32, V5_NST = 4;
constexpr int V5_A_BYTES = V5_BM * 64, V5_STAGE_BYTES = (V5_BM + V5_BN) * 64;
constexpr int V5_EPI_LD = 164;                       // halfs per staged row: 320 data bytes + 8 pad (ds_write_b64 conflict-free)
constexpr int V5_EPI_BYTES = 16 * V5_EPI_LD * 2;     // one 16-row strip of a wave
constexpr int V5_SWZ = 0x78;                         // g[x] = (0x78 >> 2x) & 3 = {0, 2, 3, 1}

// one 16-row x 160-column strip of a wave's tile: staged fp16 values -> fused epilogue -> 16-byte global stores
// EPI (compile time: one epilogue per kernel instantiation keeps its code and its register demand small — with all three inlined
// into one kernel hipcc spilled 160 registers there and the LayerNorm-folded GEGLU of level 2 ran at 234 us instead of 140):
//   0 plain (bias / row vector / activation / residual), 1 plain + LayerNorm-fold statistics out, 2 GEGLU
template <int EPI>
__device__ __forceinline__ void v5_epilogue_strip(const GemmParams& p, half_t* Cs, int z, int m_base, int n_base, int lane, int part, bool bias_done) {
    const bool rows_full = m_base + 16 <= p.M;                          // (wave-uniform) every row of the strip exists: the branch-free paths
    // (GEGLU keeps the predicated loop: with 40 accumulators of the next strips still live, the batched / interleaved form of the 128 x 160
    // kernel's epilogue spills here and measured 14 % slower per launch at 4096 x 10240 x 1280)
    if (EPI == 2) {   // GEGLU: the wave's 160 columns are one [80 value | 80 gate] block -> 80 outputs
        if (rows_full) {
            // Round 5: whole strips take the batched form — every global / LDS operand of the strip requested first, then the stage-by-stage
            // GELUs of common.h (8 per chunk; with the one-transcendental GELU its live set is 40 registers: no spills next to the 80
            // accumulators of the strips still waiting, which is what ruled this form out with round 4's GELU: -14 % per launch then).
            // 160 chunk pairs over 64 lanes: two full rounds and one of 32 lanes (the others recompute chunk 0 and do not store).
            half_t* Cb = p.C + (long long)z * p.sC + (long long)m_base * p.ldc + n_base / 2;
            const bool hr = p.R != nullptr;
            const half_t* Rb = hr ? p.R + (long long)z * p.sR + (long long)m_base * p.ldr + n_base / 2 : nullptr;
            // (the value / gate biases are in the staged strip already: v5_finish adds them in fp32 before the rounding)
            uint4 rres[3], ca[3], cg[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int q0 = lane + it * 64;
                const int q = q0 < 160 ? q0 : 0;
                const int row = q / 10, cc = q - row * 10;
                rres[it] = hr ? ld16(Rb + (long long)row * p.ldr + cc * 8) : zero16();
                ca[it] = ld16(Cs + row * V5_EPI_LD + cc * 8);
                cg[it] = ld16(Cs + row * V5_EPI_LD + 80 + cc * 8);
            }
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int q0 = lane + it * 64;
                const int q = q0 < 160 ? q0 : 0;
                const int row = q / 10, cc = q - row * 10;
                float g[8];
                unpack8(cg[it], g);
                const unsigned aw[4] = {ca[it].x, ca[it].y, ca[it].z, ca[it].w};
                const f32x2 gp[4] = {{g[0], g[1]}, {g[2], g[3]}, {g[4], g[5]}, {g[6], g[7]}};
                unsigned ow[4];
                geglu8_staged(aw, gp, ow);
                uint4 packed = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                if (hr) packed = add8h(packed, rres[it]);
                if (q0 < 160) st16(Cb + (long long)row * p.ldc + cc * 8, packed);
            }
            return;
        }
        uint4 rba[3], rbg[3], rres[3];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int q = lane + it * 64;
            const int row = q / 10, cc = q - row * 10;
            const int m = m_base + row, nv = n_base + cc * 8;
            const bool ok = q < 160 && m < p.M;
            rba[it] = (ok && !bias_done) ? ld16(p.bias_n + nv) : zero16();
            rbg[it] = (ok && !bias_done) ? ld16(p.bias_n + nv + 80) : zero16();
            rres[it] = (ok && p.R != nullptr) ? ld16(p.R + (long long)z * p.sR + (long long)m * p.ldr + n_base / 2 + cc * 8) : zero16();
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int q = lane + it * 64;
            const int row = q / 10, cc = q - row * 10;
            const int m = m_base + row;
            if (q < 160 && m < p.M) {
                float a[8], g[8], ba[8], bg[8], r[8];
                unpack8(ld16(Cs + row * V5_EPI_LD + cc * 8), a);
                unpack8(ld16(Cs + row * V5_EPI_LD + 80 + cc * 8), g);
                unpack8(rba[it], ba);
                unpack8(rbg[it], bg);
                unpack8(rres[it], r);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (a[j] + ba[j]) * gelu_f(g[j] + bg[j]) + r[j];
                st16(p.C + (long long)z * p.sC + (long long)m * p.ldc + n_base / 2 + cc * 8, pack8(a));
            }
        }
        return;
    }
    const bool hb = p.bias_n != nullptr && !bias_done, hv = p.rowvec != nullptr, hr = p.R != nullptr;
    if (rows_full && p.bias_m == nullptr && p.act == 0 && bias_done) {
        // branch-free, all five chunks of a lane requested as one batch, and in PACKED fp16 (round 5, as the 128 x 160 kernel's tile epilogue:
        // the bias is in the staged strip already — v5_finish adds it in fp32 before the one rounding — so a chunk is strip (+ time-embedding
        // row) (+ residual) by v_pk_add_f16, exact sums rounded once, and the LayerNorm-fold row statistics come from v_dot2_f32_f16)
        half_t* Cb = p.C + (long long)z * p.sC + (long long)m_base * p.ldc + n_base;
        const half_t* Rb = hr ? p.R + (long long)z * p.sR + (long long)m_base * p.ldr + n_base : nullptr;
        float s1[5], s2[5];
        uint4 rv[5], rres[5], cv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int q = lane + k * 64;
            const int row = q / 20, cc = q - row * 20;
            rv[k] = hv ? ld16(p.rowvec + (long long)((m_base + row) / p.rows_per_vec) * p.ldrv + n_base + cc * 8) : zero16();
            rres[k] = hr ? ld16(Rb + (long long)row * p.ldr + cc * 8) : zero16();
            cv[k] = ld16(Cs + row * V5_EPI_LD + cc * 8);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int q = lane + k * 64;
            const int row = q / 20, cc = q - row * 20;
            uint4 packed = cv[k];
            if (hv) packed = add8h(packed, rv[k]);
            if (hr) packed = add8h(packed, rres[k]);
            st16(Cb + (long long)row * p.ldc + cc * 8, packed);
            if (EPI == 1 && p.stat_out != nullptr) {   // LN-fold producer: row statistics of the stored fp16 values
                const half2v one2 = {(half_t)1.f, (half_t)1.f};
                const half2v h0 = __builtin_bit_cast(half2v, packed.x), h1 = __builtin_bit_cast(half2v, packed.y);
                const half2v h2 = __builtin_bit_cast(half2v, packed.z), h3 = __builtin_bit_cast(half2v, packed.w);
                float a1 = __builtin_amdgcn_fdot2(h1, one2, __builtin_amdgcn_fdot2(h0, one2, 0.f, false), false);
                float a2 = __builtin_amdgcn_fdot2(h1, h1, __builtin_amdgcn_fdot2(h0, h0, 0.f, false), false);
                s1[k] = __builtin_amdgcn_fdot2(h3, one2, __builtin_amdgcn_fdot2(h2, one2, a1, false), false);
                s2[k] = __builtin_amdgcn_fdot2(h3, h3, __builtin_amdgcn_fdot2(h2, h2, a2, false), false);
            }
        }
        if (EPI == 1 && p.stat_out != nullptr) {   // chunk partials -> LDS (the strip has been consumed) -> one lane per row, in chunk order
            float* sc = reinterpret_cast<float*>(Cs);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 5; ++it) *reinterpret_cast<float2*>(sc + (lane + it * 64) * 2) = make_float2(s1[it], s2[it]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 16) {
                float a = 0.f, b = 0.f;
#pragma unroll 4
                for (int c = 0; c < 20; ++c) {
                    const float2 t = *reinterpret_cast<const float2*>(sc + (lane * 20 + c) * 2);
                    a += t.x;
                    b += t.y;
                }
                *reinterpret_cast<float2*>(p.stat_out + ((long long)part * p.M + m_base + lane) * 2) = make_float2(a, b);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
    uint4 rb[5], rv[5], rres[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int q = lane + it * 64;
        const int row = q / 20, cc = q - row * 20;
        const int m = m_base + row, n = n_base + cc * 8;
        const bool ok = m < p.M;
        rb[it] = (ok && hb) ? ld16(p.bias_n + n) : zero16();
        rv[it] = (ok && hv) ? ld16(p.rowvec + (long long)(m / p.rows_per_vec) * p.ldrv + n) : zero16();
        rres[it] = (ok && hr) ? ld16(p.R + (long long)z * p.sR + (long long)m * p.ldr + n) : zero16();
    }
    float s1[5], s2[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int q = lane + it * 64;
        const int row = q / 20, cc = q - row * 20;
        const int m = m_base + row, n = n_base + cc * 8;
        s1[it] = s2[it] = 0.f;
        if (m < p.M) {
            float v[8], b[8], e[8], r[8];
            unpack8(ld16(Cs + row * V5_EPI_LD + cc * 8), v);
            unpack8(rb[it], b);
            unpack8(rv[it], e);
            unpack8(rres[it], r);
            const float bm = p.bias_m != nullptr ? (float)p.bias_m[m] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = v[j] + b[j] + bm + e[j];
                if (p.act == 1) t = silu_f(t);
                else if (p.act == 3) t = quick_gelu_f(t);
                v[j] = t + r[j];
            }
            const uint4 packed = pack8(v);
            st16(p.C + (long long)z * p.sC + (long long)m * p.ldc + n, packed);
            if (EPI == 1 && p.stat_out != nullptr) {   // LN-fold producer: row statistics of what was actually stored (the fp16 values)
                float f[8];
                unpack8(packed, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s1[it] += f[j];
                    s2[it] += f[j] * f[j];
                }
            }
        }
    }
    if (EPI == 1 && p.stat_out != nullptr) {   // chunk partials -> LDS (the strip has been consumed) -> one lane per row sums them in chunk order
        float* sc = reinterpret_cast<float*>(Cs);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int q = lane + it * 64;
            sc[q * 2] = s1[it];
            sc[q * 2 + 1] = s2[it];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16 && m_base + lane < p.M) {
            float a = 0.f, b = 0.f;
            for (int c = 0; c < 20; ++c) {
                a += sc[(lane * 20 + c) * 2];
                b += sc[(lane * 20 + c) * 2 + 1];
            }
            float* o = p.stat_out + ((long long)part * p.M + m_base + lane) * 2;
            o[0] = a;
            o[1] = b;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// tail shared by the 256 x 320 tile kernels (v5 / v6): split-K slab store, or the staged fused epilogue (two 16-row strips at a
// time through this wave's 10.5 KB of the — by now quiet — LDS ring)
template <int EPI, bool LNC>   // EPI: see v5_epilogue_strip; LNC: LayerNorm-fold consumer (ln_mu / ln_rs valid)
__device__ __forceinline__ void v5_finish(const GemmParams& p, f32x4 (&acc)[4][10], char* smem5, const float* ln_mu, const float* ln_rs, int z, int m0,
                                          int n0, int wm0, int wn0, int wid, int lane, int ks, int splitk, int tn_i) {
    constexpr int TM = 4, TN = 10;
    const int fr = lane & 15, fq = lane >> 4;
    const int m_w = m0 + wm0, n_w = n0 + wn0;
#ifdef LD_AB_BUILD
    if (p.dbg & 4096) return;                                       // ablation (tools/conv6_abl.py; wrong results, timing only): no epilogue
#endif
    if (splitk > 1) {
        float* part = p.partial + (long long)ks * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m_w + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n_w + j * 16 + fq * 4;
                if (m < p.M) {
                    f32x4 v = acc[i][j];
                    v *= p.alpha;
                    *reinterpret_cast<f32x4*>(part + (long long)m * p.N + n) = v;
                }
            }
        }
        return;
    }
    // the ring is quiet (every wave is past its last fragment read and DMA wait): each wave stages two 16-row strips at a time in
    // its own 10.5 KB of it, so at most half of the accumulators are live next to the epilogue's prefetch registers
    half_t* Cs = reinterpret_cast<half_t*>(smem5 + wid * 2 * V5_EPI_BYTES);
    const int part = tn_i * 2 + (wid & 1);                          // LN-fold statistics: one part per 160-column half tile
    const bool ln = LNC && p.ln_stat != nullptr;
    // plain epilogues (no activation) and GEGLU: the bias is added HERE, in fp32 before the one rounding to fp16, and the strips add none
    const bool bias_done = EPI == 2 || p.act == 0;                 // (GEGLU: value and gate biases alike)
    const bool add_b = bias_done && p.bias_n != nullptr;
    // (always a load: an absent bias reads the zero page — a select around a load makes hipcc branch and wait per load; per strip, from L1 after
    // the first: a batch held for all four strips costs 20 registers next to the 160 accumulators and spilled)
    const half_t* bsrc = (add_b ? p.bias_n + n_w : reinterpret_cast<const half_t*>(g_zero_row)) + fq * 4;
    auto stage = [&](auto I, half_t* dst) {                         // literal strip index: the accumulators stay in registers
        constexpr int i = decltype(I)::value;
        // LN-fold consumer: acc <- rstd * (acc - mu * wsum) in fp32, strip by strip (keeps the live registers low)
        const float mu = ln ? ln_mu[wm0 + i * 16 + fr] : 0.f, rs = ln ? ln_rs[wm0 + i * 16 + fr] : 1.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = acc[i][j];
            if (ln) {
                const f32x4 ws = *reinterpret_cast<const f32x4*>(p.ln_wsum + n_w + j * 16 + fq * 4);
                v = (v - mu * ws) * rs;
            }
            const half4 bh = *reinterpret_cast<const half4*>(bsrc + j * 16);   // (L1-resident after the first strip; GEGLU: [80 value | 80 gate] biases, the strip's column order)
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = v[r] * p.alpha + (float)bh[r];
            *reinterpret_cast<uint2*>(dst + fr * V5_EPI_LD + j * 16 + fq * 4) = make_uint2(pk2h(o[0], o[1]), pk2h(o[2], o[3]));
        }
    };
    half_t* Cs1 = Cs + 16 * V5_EPI_LD;
    stage(std::integral_constant<int, 0>{}, Cs);
    stage(std::integral_constant<int, 1>{}, Cs1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // same wave, in-order LDS: the strips are complete
    __builtin_amdgcn_sched_barrier(0);
    v5_epilogue_strip<EPI>(p, Cs, z, m_w, n_w, lane, part, bias_done);
    v5_epilogue_strip<EPI>(p, Cs1, z, m_w + 16, n_w, lane, part, bias_done);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // strips consumed before the next pair overwrites them
    __builtin_amdgcn_sched_barrier(0);
    stage(std::integral_constant<int, 2>{}, Cs);
    stage(std::integral_constant<int, 3>{}, Cs1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    v5_epilogue_strip<EPI>(p, Cs, z, m_w + 32, n_w, lane, part, bias_done);
    v5_epilogue_strip<EPI>(p, Cs1, z, m_w + 48, n_w, lane, part, bias_done);
}

// Epilogue of the halo-tile kernel's narrower tiles (256 x 256: the VAE's N = 256 / 512 convolutions; written for any width 16 TN per
// wave): wave tile 64 x WN, WN = 16 TN.  Convolutions carry no LayerNorm fold and no
// GEGLU, so this is the plain epilogue only (bias / bias_m / row vector / SiLU / residual), written once for every WN: two 16-row strips
// at a time through the wave's slice of the (quiet) LDS, operands of a strip requested as one batch, predicated stores.
template <int TM, int TN>
__device__ __forceinline__ void v6_finish(const GemmParams& p, f32x4 (&acc)[TM][TN], char* smem5, int m0, int n0, int wm0, int wn0, int wid, int lane, int ks,
                                          int splitk, int gimg, int gchunk) {
    static_assert(TM == 4 || TM == 8, "wave tile of 64 or 128 rows");
    constexpr int WN = TN * 16, LD = WN + 4, STRIP_BYTES = 16 * LD * 2;
    constexpr int CPR = WN / 8, TOT = 16 * CPR, ITS = (TOT + 63) / 64;
    const int fr = lane & 15, fq = lane >> 4;
    const int m_w = m0 + wm0, n_w = n0 + wn0;
    if (splitk > 1) {
        float* part = p.partial + (long long)ks * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m_w + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (m < p.M) {
                    f32x4 v = acc[i][j];
                    v *= p.alpha;
                    *reinterpret_cast<f32x4*>(part + (long long)m * p.N + n_w + j * 16 + fq * 4) = v;
                }
            }
        }
        return;
    }
    half_t* Cs0 = reinterpret_cast<half_t*>(smem5 + wid * 2 * STRIP_BYTES);
    const bool hb = p.bias_n != nullptr, hv = p.rowvec != nullptr, hr = p.R != nullptr;
    // GroupNorm statistics of the OUTPUT (GemmParams::gn_part, host-checked: whole tiles of one image, groups of 4 or of whole 8-channel
    // chunks): a lane always handles the same 8-channel chunk, so it sums what it stores (the fp16-rounded values) over its rows —
    // (sum, sum of squares) of channels 0-3 / 4-7 apart when a group is 4 channels wide — and the tile's partials are put together below
    const bool gne = p.gn_part != nullptr, g4 = p.N == 128;
    float gs[4] = {0.f, 0.f, 0.f, 0.f};
    // plain epilogues (no activation, no per-row bias): the bias is added HERE, in fp32 before the one rounding to fp16, and the strips work
    // in packed fp16 (round 5, as v5_finish: the fp32 form cost ~70 vector instructions per 16-byte chunk — a quarter of the N = 128 convolutions'
    // launch time at K = 1152)
    const bool packed_ok = p.act == 0 && p.bias_m == nullptr;
    const half_t* bsrc = ((packed_ok && hb) ? p.bias_n + n_w : reinterpret_cast<const half_t*>(g_zero_row)) + fq * 4;   // (always a load: zero page without a bias)
    auto stage = [&](auto I, half_t* dst) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const half4 bh = *reinterpret_cast<const half4*>(bsrc + j * 16);
            f32x4 v = acc[i][j] * p.alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)bh[r];
            *reinterpret_cast<uint2*>(dst + fr * LD + j * 16 + fq * 4) = make_uint2(pk2h(v[0], v[1]), pk2h(v[2], v[3]));
        }
    };
    auto strip = [&](const half_t* Cs, int m_base) {
        if (packed_ok) {
            uint4 rv[ITS], rres[ITS], cv[ITS];
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int q0 = lane + it * 64;
                const int q = (TOT % 64 == 0 || q0 < TOT) ? q0 : 0;
                const int row = q / CPR, cc = q - row * CPR;
                const int m = m_base + row < p.M ? m_base + row : p.M - 1;
                const int n = n_w + cc * 8;
                rv[it] = hv ? ld16(p.rowvec + (long long)(m / p.rows_per_vec) * p.ldrv + n) : zero16();
                rres[it] = hr ? ld16(p.R + (long long)m * p.ldr + n) : zero16();
                cv[it] = ld16(Cs + row * LD + cc * 8);
            }
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int q0 = lane + it * 64;
                const int q = (TOT % 64 == 0 || q0 < TOT) ? q0 : 0;
                const int row = q / CPR, cc = q - row * CPR;
                uint4 packed = cv[it];
                if (hv) packed = add8h(packed, rv[it]);
                if (hr) packed = add8h(packed, rres[it]);
                if ((TOT % 64 == 0 || q0 < TOT) && m_base + row < p.M && n_w + cc * 8 < p.n_valid) {
                    st16(p.C + (long long)(m_base + row) * p.ldc + n_w + cc * 8, packed);
                    if (gne) {   // v_dot2_f32_f16 on the packed pairs: 8 instructions per chunk
                        const half2v one2 = {(half_t)1.f, (half_t)1.f};
                        const half2v h0 = __builtin_bit_cast(half2v, packed.x), h1 = __builtin_bit_cast(half2v, packed.y);
                        const half2v h2 = __builtin_bit_cast(half2v, packed.z), h3 = __builtin_bit_cast(half2v, packed.w);
                        gs[0] = __builtin_amdgcn_fdot2(h1, one2, __builtin_amdgcn_fdot2(h0, one2, gs[0], false), false);     // channels 0-3
                        gs[1] = __builtin_amdgcn_fdot2(h1, h1, __builtin_amdgcn_fdot2(h0, h0, gs[1], false), false);
                        gs[2] = __builtin_amdgcn_fdot2(h3, one2, __builtin_amdgcn_fdot2(h2, one2, gs[2], false), false);     // channels 4-7
                        gs[3] = __builtin_amdgcn_fdot2(h3, h3, __builtin_amdgcn_fdot2(h2, h2, gs[3], false), false);
                    }
                }
            }
            return;
        }
        uint4 rb[ITS], rv[ITS], rres[ITS], cv[ITS];
        half_t rm[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int q0 = lane + it * 64;
            const int q = (TOT % 64 == 0 || q0 < TOT) ? q0 : 0;
            const int row = q / CPR, cc = q - row * CPR;
            const int m = m_base + row < p.M ? m_base + row : p.M - 1;
            const int n = n_w + cc * 8;
            rb[it] = hb ? ld16(p.bias_n + n) : zero16();
            rv[it] = hv ? ld16(p.rowvec + (long long)(m / p.rows_per_vec) * p.ldrv + n) : zero16();
            rres[it] = hr ? ld16(p.R + (long long)m * p.ldr + n) : zero16();
            rm[it] = p.bias_m != nullptr ? p.bias_m[m] : (half_t)0.f;
            cv[it] = ld16(Cs + row * LD + cc * 8);
        }
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int q0 = lane + it * 64;
            const int q = (TOT % 64 == 0 || q0 < TOT) ? q0 : 0;
            const int row = q / CPR, cc = q - row * CPR;
            float v[8], b[8], e[8], r[8];
            unpack8(cv[it], v);
            unpack8(rb[it], b);
            unpack8(rv[it], e);
            unpack8(rres[it], r);
            const float bm = (float)rm[it];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = v[j] + b[j] + bm + e[j];
                if (p.act == 1) t = silu_f(t);
                else if (p.act == 3) t = quick_gelu_f(t);
                v[j] = t + r[j];
            }
            if ((TOT % 64 == 0 || q0 < TOT) && m_base + row < p.M && n_w + cc * 8 < p.n_valid) {
                const uint4 packed = pack8(v);
                st16(p.C + (long long)(m_base + row) * p.ldc + n_w + cc * 8, packed);
                if (gne) {   // v_dot2_f32_f16 on the packed pairs: 8 instructions per chunk
                    const half2v one2 = {(half_t)1.f, (half_t)1.f};
                    const half2v h0 = __builtin_bit_cast(half2v, packed.x), h1 = __builtin_bit_cast(half2v, packed.y);
                    const half2v h2 = __builtin_bit_cast(half2v, packed.z), h3 = __builtin_bit_cast(half2v, packed.w);
                    gs[0] = __builtin_amdgcn_fdot2(h1, one2, __builtin_amdgcn_fdot2(h0, one2, gs[0], false), false);     // channels 0-3
                    gs[1] = __builtin_amdgcn_fdot2(h1, h1, __builtin_amdgcn_fdot2(h0, h0, gs[1], false), false);
                    gs[2] = __builtin_amdgcn_fdot2(h3, one2, __builtin_amdgcn_fdot2(h2, one2, gs[2], false), false);     // channels 4-7
                    gs[3] = __builtin_amdgcn_fdot2(h3, h3, __builtin_amdgcn_fdot2(h2, h2, gs[3], false), false);
                }
            }
        }
    };
    half_t* Cs1 = Cs0 + 16 * LD;
    auto pair = [&](auto P) {                                            // strips 2P, 2P + 1 (literal indices: the accumulators stay in registers)
        constexpr int i0 = 2 * decltype(P)::value;
        stage(std::integral_constant<int, i0>{}, Cs0);
        stage(std::integral_constant<int, i0 + 1>{}, Cs1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // same wave, in-order LDS: the strips are complete
        __builtin_amdgcn_sched_barrier(0);
        strip(Cs0, m_w + 16 * i0);
        strip(Cs1, m_w + 16 * i0 + 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // strips consumed before the next pair overwrites them
        __builtin_amdgcn_sched_barrier(0);
    };
    pair(std::integral_constant<int, 0>{});
    pair(std::integral_constant<int, 1>{});
    if constexpr (TM == 8) {
        pair(std::integral_constant<int, 2>{});
        pair(std::integral_constant<int, 3>{});
    }
    if (gne && 64 % CPR == 0) {   // (workgroup-uniform; a lane keeps its chunk over the strips only when CPR divides 64: the host asks for it at BN = 256 / 128 only)
        if (!g4) {   // groups of whole chunks: the two halves of the chunk belong together
            gs[0] += gs[2];
            gs[1] += gs[3];
        }
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) gs[k] += __shfl_xor(gs[k], o, 64);
        }
        float* wp = reinterpret_cast<float*>(smem5 + 8 * 2 * STRIP_BYTES);      // [8 waves][CPR chunks][4], behind every wave's strips
        if (lane < CPR) *reinterpret_cast<float4*>(wp + (wid * CPR + lane) * 4) = make_float4(gs[0], gs[1], gs[2], gs[3]);
        __syncthreads();
        // one thread per group of the tile: the four waves of its column half top to bottom, the group's chunks left to right (fixed order)
        const int cpg = p.N / 32, tid = wid * 64 + lane;
        if (tid < 2 * WN / cpg) {
            const int col = tid * cpg, wn = col / WN, cf = (col - wn * WN) >> 3;
            const int nch = cpg >= 8 ? cpg >> 3 : 1, part = (cpg == 4 && (col & 4)) ? 2 : 0;
            float s = 0.f, ss = 0.f;
            for (int wmi = 0; wmi < 4; ++wmi)
                for (int c = 0; c < nch; ++c) {
                    const float* e = wp + ((wmi * 2 + wn) * CPR + cf + c) * 4 + part;
                    s += e[0];
                    ss += e[1];
                }
            float* o = p.gn_part + (((long long)gimg * p.gn_P + gchunk) * 32 + (n0 + col) / cpg) * 2;
            o[0] = s;
            o[1] = ss;
        }
    }
}

template <bool CONV, int EPI>
__global__ __launch_bounds__(512, 2) void gemm5_kernel(const GemmParams p) {
    constexpr int TM = 4, TN = 10;
    __shared__ __attribute__((aligned(16))) char smem5[V5_NST * V5_STAGE_BYTES];
    static_assert(8 * 2 * V5_EPI_BYTES <= V5_NST * V5_STAGE_BYTES, "epilogue staging must fit in the ring");
    __shared__ __attribute__((aligned(16))) float ln_mu[V5_BM], ln_rs[V5_BM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grp1 = wid >= 4;
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 160;
    const int z = blockIdx.z;
    const int tiles_m = (p.M + V5_BM - 1) / V5_BM, tiles_n = p.N / V5_BN;
    const int tiles = tiles_m * tiles_n;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    int bid = xcd_remap(blockIdx.x, tiles * splitk);
    const int ks = bid / tiles;
    bid -= ks * tiles;
    int tn_i = bid % tiles_n, tm_i = bid / tiles_n;
    if (!CONV && p.xcd_gm > 0) {
        // plain GEMM, XCD-blocked tile order: XCD x (a contiguous run of tiles / 8 logical ids) owns the block (x / gn, x % gn) of a gm x gn grid
        // over the tile matrix and walks it M-fastest — the workgroups that run together on one XCD share a few W tiles and A panels through
        // its L2.  (N-fastest over whole M panels made every XCD stream ALL of W once per M panel: 433 MB fetched per launch for the
        // 26 MB matrix of the level-2 GEGLU, profiles/pmc_traffic.json round 3.)
        const int gn = 8 / p.xcd_gm, per = tiles >> 3, x = bid / per, l = bid - x * per;
        const int bm_t = tiles_m / p.xcd_gm, bn_t = tiles_n / gn;
        tm_i = (x / gn) * bm_t + l % bm_t;
        tn_i = (x % gn) * bn_t + l / bm_t;
    }
#ifdef LD_AB_BUILD
    const int m0 = (p.dbg & 64) ? 0 : tm_i * V5_BM, n0 = (p.dbg & 64) ? 0 : tn_i * V5_BN;
#else
    const int m0 = tm_i * V5_BM, n0 = tn_i * V5_BN;
#endif
    const int KT = p.K / V5_BK;
    const int kt_begin = (int)((long long)ks * KT / splitk), kt_end = (int)((long long)(ks + 1) * KT / splitk);
    const int nk = kt_end - kt_begin;                              // >= 2 (gemm_launch)

    const half_t* Ab = p.A + (long long)z * p.sA;
    const half_t* Wb = p.W + (long long)z * p.sW;
    const half_t* zp = reinterpret_cast<const half_t*>(g_zero_row);
    const int Cin = p.C1 + p.C2;

    // ---- loader state: 2 A pieces and 2 (waves 4-7) or 3 (waves 0-3) B pieces per wave and step; a piece = 16 rows x 64 bytes
    const int prow = lane >> 2;                                    // row inside a piece (piece rows start at multiples of 16)
    const int lchunk = (lane & 3) ^ ((V5_SWZ >> (2 * ((prow >> 2) & 3))) & 3);   // logical chunk this lane fetches
    const int a_row0 = wid * 32 + prow;                            // + 16 for the second piece
    const half_t* a_ptr[2];
    unsigned a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + a_row0 + i * 16;
        a_ptr[i] = zp;
        a_off[i] = CONV ? 0u : (unsigned)(((long long)(m < p.M ? m : p.M - 1) * p.lda + lchunk * 8) * 2);
    }
    const half_t* a_base = Ab + (long long)kt_begin * V5_BK;        // wave-uniform (plain GEMM)
    int seg_left = 0;
    int k_issue = kt_begin * V5_BK;                                 // K index of the next step to issue
    auto conv_seek = [&](int k0) {
        // (k0 beyond the taps: the second K segment — the 1x1 skip convolution's raw sources at the output pixel itself, gemm.h S1 / S2)
        const int K9 = p.ksize * p.ksize * Cin;
        const bool skp = k0 >= K9 && p.SC1 > 0;
        const int tap = skp ? 0 : k0 / Cin;
        const int c0 = skp ? k0 - K9 : k0 - tap * Cin;
        const int ky = skp ? p.pad : tap / p.ksize, kx = skp ? p.pad : tap - (tap / p.ksize) * p.ksize;
        const int Ca = skp ? p.SC1 : p.C1, Cb = skp ? p.SC2 : p.C2;
        const bool second = c0 >= Ca;
        const half_t* src = skp ? (second ? p.S2 : p.S1) : (second ? p.A2 : Ab);
        const int Cs = second ? Cb : Ca;
        const int cl = second ? c0 - Ca : c0;
        seg_left = ((second ? Ca + Cb : Ca) - c0) / V5_BK;
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + a_row0 + i * 16;
            const int mm = m < p.M ? m : 0;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
            const bool ok = m < p.M && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv && tap < p.ksize * p.ksize && c0 < Ca + Cb;
            int sy = iy, sx = ix;
            if (p.Hv == 2 * p.Hs && p.Wv == 2 * p.Ws) {
                sy = iy >> 1;
                sx = ix >> 1;
            } else if (p.Hv != p.Hs || p.Wv != p.Ws) {
                sy = (int)((long long)iy * p.Hs / p.Hv);
                sx = (int)((long long)ix * p.Ws / p.Wv);
            }
            a_ptr[i] = ok ? src + (((long long)img * p.Hs + sy) * p.Ws + sx) * Cs + cl + lchunk * 8 : zp + lchunk * 8;
        }
    };
    unsigned b_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int piece = i < 2 ? wid * 2 + i : 16 + (wid & 3);
        const int row = piece * 16 + prow;
        const int n = n0 + row < p.n_valid ? n0 + row : p.n_valid - 1;
        b_off[i] = (unsigned)(((long long)n * p.ldw + lchunk * 8) * 2);
    }
    const half_t* b_base = Wb + (long long)kt_begin * V5_BK;        // wave-uniform

    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem5);
    unsigned st_issue = 0;                                          // byte offset of the stage the next issued step goes to
    auto issue = [&]() {
        const unsigned As = smem_base + st_issue + (unsigned)(wid * 2) * 1024u;
        const unsigned Bs = smem_base + st_issue + (unsigned)V5_A_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (CONV) glds16(a_ptr[i], As + (unsigned)i * 1024u);
            else glds16s(a_off[i], a_base, As + (unsigned)i * 1024u);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16s(b_off[i], b_base, Bs + (unsigned)(wid * 2 + i) * 1024u);
        if (!grp1) glds16s(b_off[2], b_base, Bs + (unsigned)(16 + wid) * 1024u);
        k_issue += V5_BK;
        if (CONV) {
            if (--seg_left <= 0) {
                conv_seek(k_issue);
            } else {
                a_ptr[0] += V5_BK;
                a_ptr[1] += V5_BK;
            }
        } else {
            a_base += V5_BK;
        }
        b_base += V5_BK;
        st_issue = st_issue == (unsigned)((V5_NST - 1) * V5_STAGE_BYTES) ? 0u : st_issue + (unsigned)V5_STAGE_BYTES;
    };
    // "my pieces of every step but the n newest ones issued have landed" (in-order completion; 5 or 4 pieces per step and wave)
    auto wait_all_but = [&](int n) {
        if (!grp1) {
            if (n >= 2) wait_vmcnt<10>();
            else if (n == 1) wait_vmcnt<5>();
            else wait_vmcnt<0>();
        } else {
            if (n >= 2) wait_vmcnt<8>();
            else if (n == 1) wait_vmcnt<4>();
            else wait_vmcnt<0>();
        }
    };
    if (CONV) conv_seek(k_issue);

    const int fr = lane & 15, fq = lane >> 4;
    // fragment read bases (bytes into stage 0): A rows wm0 + 16 i + fr, B rows wn0 + 16 j + fr; (row >> 2) & 3 == (fr >> 2) & 3
    const unsigned rchunk = (unsigned)(fq ^ ((V5_SWZ >> (2 * ((fr >> 2) & 3))) & 3)) << 4;
    const char* rdA = smem5 + (wm0 + fr) * 64 + rchunk;
    const char* rdB = smem5 + V5_A_BYTES + (wn0 + fr) * 64 + rchunk;
    int st_read = 0;                                                // stage index of the step this wave reads next

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 fa[TM], fb[TN];

    // ---- prologue: steps 0 .. 2 in flight, step 0 landed and published, group 1 one barrier behind
    issue();
    issue();
    if (nk > 2) issue();
    if (EPI != 0 && p.ln_stat != nullptr) ln_prepare<V5_BM, V5_BN>(p, ln_mu, ln_rs, z, m0, n0, tid);   // (the loop's barriers publish it)
    wait_all_but(nk > 2 ? 2 : 1);
    __builtin_amdgcn_s_barrier();
    if (grp1) __builtin_amdgcn_s_barrier();

    for (int k = 0; k < nk; ++k) {
        // ------------------------------------------------ read phase (the partner wave of this SIMD is in its MFMA phase)
#ifdef LD_AB_BUILD
        if (p.dbg & 32) __builtin_amdgcn_s_setprio(2);
        if (k + 3 < nk && !(p.dbg & 1) && !(p.dbg & 8)) issue();
        if (!(p.dbg & 2) || k == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = as_half8(ld16(rdB + j * 1024));
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = as_half8(ld16(rdA + i * 1024));
        }
        if (p.dbg & 8) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (k + 3 < nk && !(p.dbg & 1)) issue();
        }
        if (p.dbg & 32) __builtin_amdgcn_s_setprio(0);
#else
        if (k + 3 < nk) issue();                                    // step k+3 -> the stage step k-1 left (both groups are done with it)
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = as_half8(ld16(rdB + j * 1024));
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = as_half8(ld16(rdA + i * 1024));
#endif
        {
            const int d = st_read == V5_NST - 1 ? -(V5_NST - 1) * V5_STAGE_BYTES : V5_STAGE_BYTES;
            rdA += d;
            rdB += d;
            st_read = st_read == V5_NST - 1 ? 0 : st_read + 1;
        }
        // my pieces of step k+1 have landed (steps k+2, k+3, if issued, may stay in flight); the barrier publishes them
#ifdef LD_AB_BUILD
        if (p.dbg & 1) wait_vmcnt<0>();
        else
#endif
            wait_all_but(k + 3 < nk ? 2 : (k + 2 < nk ? 1 : 0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------ MFMA phase (the partner reads / stages)
#ifdef LD_AB_BUILD
        if (!(p.dbg & 16))
#endif
            __builtin_amdgcn_s_setprio(1);
#ifdef LD_AB_BUILD
        if (p.dbg & 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[0], acc[0][j], 0, 0, 0);
        } else
#endif
        {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (!grp1) __builtin_amdgcn_s_barrier();                        // group 0 waits out group 1's last MFMA phase: every wave ran 2 nk + 2 barriers

    v5_finish<EPI, EPI != 0>(p, acc, smem5, ln_mu, ln_rs, z, m0, n0, wm0, wn0, wid, lane, ks, splitk, tn_i);
}

// =====================================================================================================================
// v6: 3x3 stride-1 convolution on the v5 skeleton (256 x 320 tile, 8 waves, two wave groups half a step apart), with the A
// operand taken from a HALO tile: the 256 output pixels of a tile are 256 / W whole image rows, so for one 32-channel slab the
// (256 / W + 2) x (W + 2) input pixels they touch are copied to LDS ONCE (LDS-DMA, image border = zero page) and the nine taps
// read their fragments from it at literal offsets — K runs slab-major, tap-minor.  Against v5's implicit im2col the A share of
// the LDS-DMA falls from 16 pieces per 32-wide step to ceil(HP / 16) pieces per NINE steps (W = 64: 25), i.e. 36 -> 22.8 pieces
// per step in total, and every input byte leaves L2 once per tile instead of nine times.
// Weights stay in the checkpoint-derived [Cout][tap][Cin] order: the B pointer just walks  +Cin per tap, +32 - 8 Cin per slab.
// The halo is double-buffered (slab s+1 is fetched during the nine steps of slab s); its rows are 64 bytes, unswizzled: the
// four A fragment reads of a step are 2-way bank-conflicted (of 14 reads; the LDS port is ~25 % busy).
// Requirements (gemm_launch): ksize 3, stride 1, pad 1, no resize, Wo == W in {16, 32, 64, 128}, Ho * Wo % 256 == 0,
// C1 % 32 == 0, C2 % 32 == 0, N % 320 == 0; a split over K is a split over slabs.
// =====================================================================================================================
// Round 3: (i) the tile is W pixels wide but the IMAGE may be wider (W = 128 only: p.Wo = 256, 512, 1024 ... — a tile is then TR rows
// of one 128-pixel column band; the VAE's 256- and 512-pixel-row stages), (ii) BM = 512 (four rows of a 128-pixel band, wave tile
// 128 x BN/2) gives the N = 128 convolutions of the VAE's last level 32 MFMAs per phase instead of 16, (iii) UP: the input is the
// nearest-2x upsampling of the source (Upsample / Upsample1, LD.py:3498-3511, 5114-5152): halo pixel (y, x) comes from source pixel
// (y >> 1, x >> 1) — only the loader's address changes.
template <int W, bool GN, int BN = V5_BN, int BM = V5_BM, bool UP = false>
                                            // GN: GroupNorm (+SiLU) of the input fused into the halo (separate instantiation: the plain conv keeps its
                                            // registers); BN: tile width 320 (the UNet's N = 320 k), 256 (the VAE's N = 256 / 512) or 128 (its N = 128)
__global__ __launch_bounds__(512, 2) void conv6_kernel(const GemmParams p) {
    constexpr int TM = BM / 64, TN = BN / 32;
    static_assert(BN == 320 || BN == 256 || BN == 160 || BN == 128 || BN == 32, "tile width");
    static_assert(BM == 256 || (BM == 512 && W == 128), "tile height: 256 pixels, or four rows of a 128-pixel band");
    static_assert(!GN || BN == V5_BN || (W == 128 && ((BN == 256 && BM == 256) || (BN == 128 && BM == 512))),
                  "the fused GroupNorm only pays where the output is one tile wide: the UNet's N = 320, the VAE's N = 256 / 128 at >= 256-pixel rows");
    static_assert(!(GN && UP), "no caller");
    constexpr int BPIECES = BN / 16, NB_ALL = BPIECES / 8, NB_EXTRA = BPIECES % 8;   // B pieces of a step: NB_ALL per wave + one more for waves < NB_EXTRA
    constexpr int TR = BM / W, HW2 = W + 2, HP = (TR + 2) * HW2;       // tile rows, halo row pitch (pixels), halo pixels
    constexpr int NH = ((HP + 15) / 16 + 7) / 8;                      // halo LDS-DMA pieces per wave and slab (uniform: spare pieces copy zeros)
    constexpr int HBYTES = NH * 8 * 1024;                              // one halo buffer
    constexpr int BSTAGE = BN * 64, NSTB = 4;                          // B ring: 4 stages of BN rows x 64 bytes
    constexpr int RING0 = 2 * HBYTES;                                  // byte offset of the B ring
    __shared__ __attribute__((aligned(16))) char smem5[2 * HBYTES + NSTB * BSTAGE];
    static_assert(2 * HBYTES + NSTB * BSTAGE <= 163840, "LDS");
    static_assert(!GN || NH <= 7, "fused GroupNorm: my (<= 7) pieces of the next slab are normalised in one go in the read phase of tap 3 (28 temporaries)");
    // fused GroupNorm: every wave keeps the 32 scales + 32 shifts of the slab being normalised in 256 bytes of LDS.  Where the
    // halo buffers and the B ring already take all 160 KB (W = 128) the tables live in spare piece slots of halo buffer 0 and the
    // spare (all-zero) pieces of every wave are sent to the last slot instead
    constexpr int HPIECES = (HP + 15) / 16;
    constexpr bool TBL_IN_HALO = 2 * HBYTES + NSTB * BSTAGE + 2048 > 163840;
    static_assert(!TBL_IN_HALO || NH * 8 - HPIECES >= 3, "two table slots and a dump slot");
    __shared__ __attribute__((aligned(16))) float gn_lds[(GN && !TBL_IN_HALO) ? 8 * 64 : 4];
    static_assert(8 * 2 * (BN == V5_BN ? V5_EPI_BYTES : 16 * (BN / 2 + 4) * 2) + 8 * (BN / 16) * 16 <= 2 * HBYTES + NSTB * BSTAGE,
                  "epilogue staging (+ the GroupNorm partials of the output) must fit");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grp1 = wid >= 4;
    const int wm = wid >> 1;
    const int wm0 = wm * (BM / 4), wn0 = (wid & 1) * (BN / 2);
    const int tiles_m = p.M / BM, tiles_n = p.N / BN;
    const int tiles = tiles_m * tiles_n;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    int bid = xcd_remap(blockIdx.x, tiles * splitk);
    const int ks = bid / tiles;
    bid -= ks * tiles;
    const int tn_i = bid % tiles_n, tm_i = bid / tiles_n;
    const int n0 = tn_i * BN;
    const int Cin = p.C1 + p.C2;
    const int NS = Cin / 32;                                           // channel slabs
    const int s_begin = (int)((long long)ks * NS / splitk), s_end = (int)((long long)(ks + 1) * NS / splitk);
    const int nk = (s_end - s_begin) * 9;                              // 32-wide steps of this workgroup (>= 9)

    const half_t* zp = reinterpret_cast<const half_t*>(g_zero_row);
    // this tile = rows row0 .. row0 + TR - 1, columns col0 .. col0 + W - 1 of image img (W < 128: the image is W wide, col0 = 0)
    const int Wimg = W == 128 ? p.Wo : W;
    const int HWo = p.Ho * Wimg;
    const int bands = Wimg / W, tiles_img = (p.Ho / TR) * bands;
    const int img = tm_i / tiles_img, t_in = tm_i - img * tiles_img;
    const int row0 = (t_in / bands) * TR, col0 = (t_in - (t_in / bands) * bands) * W;

    // ---- halo loader state: piece j of this wave covers halo pixels (wid + 8 j) * 16 .. + 15; lane -> (pixel, 16-byte chunk)
    int hpix[NH];                                                      // source pixel index inside the image, or -1 (border / spare)
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int hp = (wid + 8 * j) * 16 + (lane >> 2);
        const int hy = hp / HW2, hx = hp - hy * HW2;
        const int iy = row0 + hy - 1, ix = col0 + hx - 1;
        const bool in = hp < HP && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)Wimg;
        hpix[j] = !in ? -1 : UP ? (iy >> 1) * (Wimg >> 1) + (ix >> 1) : iy * Wimg + ix;
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem5);
    auto issue_halo = [&](int s, int buf) {                            // channel slab s (32 channels of the concatenated input) -> halo buffer buf
        const int c0 = s * 32;
        const bool second = c0 >= p.C1;
        const half_t* src = (second ? p.A2 : p.A) + (long long)img * p.Hs * p.Ws * (second ? p.C2 : p.C1) + (second ? c0 - p.C1 : c0) + (lane & 3) * 8;
        const int Cs = second ? p.C2 : p.C1;
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            const half_t* g = hpix[j] >= 0 ? src + (long long)hpix[j] * Cs : zp;
            const int slot = (GN && TBL_IN_HALO && wid + 8 * j >= HPIECES) ? NH * 8 - 1 : wid + 8 * j;
            glds16(g, smem_base + (unsigned)(buf * HBYTES) + (unsigned)slot * 1024u);
        }
    };
    // ---- fused GroupNorm (+SiLU) of the input (GN): y = x * scale[img][c] + shift[img][c], applied to the halo IN LDS, each lane on
    // the 16-byte chunks it copied itself (so only its own DMA wait orders it).  The 32 scales and 32 shifts of a slab are fetched by
    // ONE untracked load per lane (lane l: entry l of [scale | shift]) and parked in this wave's 256-byte LDS table.
    // Border pixels stay zero: the convolution pads the NORMALISED tensor.
    float gn_tbl = 0.f;
    auto gn_load = [&](int s) {          // issued BEFORE the halo pieces of the same slab: their wait covers it (in-order completion)
        const float* src = ((lane & 32) ? p.gn_shift : p.gn_scale) + (long long)img * Cin + s * 32 + (lane & 31);
        asm volatile("global_load_dword %0, %1, off" : "=&v"(gn_tbl) : "v"(src) : "memory");
    };
    float* const gn_mine = (TBL_IN_HALO ? reinterpret_cast<float*>(smem5 + HPIECES * 1024) : gn_lds) + wid * 64;
    auto gn_apply_slab = [&](int buf) {
        // runs at the head of a read phase, fenced off from the fragment reads behind it: the 56 fragment registers are dead there,
        // so the temporaries below cost no accumulator spills.  All NH pieces in one go: their LDS reads overlap each other.
        __builtin_amdgcn_sched_barrier(0);
        gn_mine[lane] = gn_tbl;          // same wave reads it back: in-order LDS, no barrier
        const float* tp = gn_mine + (lane & 3) * 8;
        H8 io[NH];
#pragma unroll
        for (int j = 0; j < NH; ++j) io[j].u = ld16(smem5 + buf * HBYTES + (wid + 8 * j) * 1024 + lane * 16);
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(tp), sc1 = *reinterpret_cast<const f32x4*>(tp + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(tp + 32), sh1 = *reinterpret_cast<const f32x4*>(tp + 36);
#pragma unroll
        for (int j = 0; j < NH; ++j) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (float)io[j].e[e] * (e < 4 ? sc0[e & 3] : sc1[e & 3]) + (e < 4 ? sh0[e & 3] : sh1[e & 3]);
                if (p.gn_silu) v *= __builtin_amdgcn_rcpf(1.0f + __expf(-v));   // SiLU with v_rcp_f32 (1 ulp; rounded to fp16 anyway)
                io[j].e[e] = (half_t)v;
            }
            if (hpix[j] >= 0) st16(smem5 + buf * HBYTES + (wid + 8 * j) * 1024 + lane * 16, io[j].u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- B loader state (as v5): NB_ALL pieces per wave and step, one more for waves < NB_EXTRA (BN = 320: 2 + waves 0-3)
    const int prow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((V5_SWZ >> (2 * ((prow >> 2) & 3))) & 3);
    const bool b_extra = wid < NB_EXTRA;                                // (wave-uniform)
    unsigned b_off[NB_ALL + 1];
#pragma unroll
    for (int i = 0; i < NB_ALL + 1; ++i) {
        const int piece = i < NB_ALL ? wid * NB_ALL + i : 8 * NB_ALL + (wid % (NB_EXTRA > 0 ? NB_EXTRA : 1));
        const int n = n0 + piece * 16 + prow;
        b_off[i] = (unsigned)(((long long)n * p.ldw + lchunk * 8) * 2);
    }
    const half_t* b_base = p.W + (long long)s_begin * 32;              // step (slab s_begin, tap 0); wave-uniform
    int b_tap = 0;
    unsigned st_issue = 0;                                             // byte offset (inside the B ring) of the stage the next step goes to
    auto issue_b = [&]() {
        const unsigned Bs = smem_base + (unsigned)RING0 + st_issue;
#pragma unroll
        for (int i = 0; i < NB_ALL; ++i) glds16s(b_off[i], b_base, Bs + (unsigned)(wid * NB_ALL + i) * 1024u);
        if (NB_EXTRA > 0 && b_extra) glds16s(b_off[NB_ALL], b_base, Bs + (unsigned)(8 * NB_ALL + wid) * 1024u);
        if (b_tap == 8) {
            b_tap = 0;
            b_base += 32 - 8 * Cin;
        } else {
            ++b_tap;
            b_base += Cin;
        }
        st_issue = st_issue == (unsigned)((NSTB - 1) * BSTAGE) ? 0u : st_issue + (unsigned)BSTAGE;
    };
    // "every LDS-DMA of mine but the n newest steps' B pieces (+ the halo pieces when they sit among those) has landed"
    auto wait_keep = [&](int steps, bool halo) {
        constexpr int PX = NB_ALL + 1, PA = NB_ALL;                      // pieces per step of a wave with / without the extra piece
        if (NB_EXTRA > 0 && b_extra) {
            if (steps >= 2) { if (halo) wait_vmcnt<2 * PX + NH>(); else wait_vmcnt<2 * PX>(); }
            else if (steps == 1) { if (halo) wait_vmcnt<PX + NH>(); else wait_vmcnt<PX>(); }
            else wait_vmcnt<0>();
        } else {
            if (steps >= 2) { if (halo) wait_vmcnt<2 * PA + NH>(); else wait_vmcnt<2 * PA>(); }
            else if (steps == 1) { if (halo) wait_vmcnt<PA + NH>(); else wait_vmcnt<PA>(); }
            else wait_vmcnt<0>();
        }
    };

    // ---- fragment read bases: A = halo pixel of output pixel (wm0 + 16 i + fr) at tap (0,0), B as v5
    const int fr = lane & 15, fq = lane >> 4;
    const int oyw = wm0 / W, oxw = wm0 - oyw * W;                      // first output pixel of this wave inside the tile
    // the wave's BM / 4 output pixels are consecutive rows of the [M][N] output (one image row segment, or whole rows of a narrow image)
    const int m0 = img * HWo + (row0 + oyw) * Wimg + col0 + oxw - wm0;   // so that m0 + wm0 is the wave's first output row
    const char* rdA = smem5 + ((oyw * HW2 + oxw + fr) * 64 + fq * 16);
    const unsigned rchunk = (unsigned)(fq ^ ((V5_SWZ >> (2 * ((fr >> 2) & 3))) & 3)) << 4;
    const char* rdB = smem5 + RING0 + (wn0 + fr) * 64 + rchunk;
    int st_read = 0;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 fa[TM], fb[TN];

    // ---- prologue: halo of the first slab and B steps 0..2 in flight; step 0 + halo landed and published; group 1 one barrier behind
    if (GN) gn_load(s_begin);
    issue_halo(s_begin, 0);
    issue_b();
    issue_b();
    issue_b();
    wait_keep(2, false);
    if (GN) gn_apply_slab(0);
    __builtin_amdgcn_s_barrier();
    if (grp1) __builtin_amdgcn_s_barrier();

    int k = 0;
    for (int s = s_begin; s < s_end; ++s) {
        const int hb = (s - s_begin) & 1;
        const char* rdAs = rdA + hb * HBYTES;
#pragma unroll
        for (int t = 0; t < 9; ++t, ++k) {
            // ------------------------------------------------ read phase (the partner wave of this SIMD is in its MFMA phase)
            if (t == 0 && s + 1 < s_end) {
                if (GN) gn_load(s + 1);
                issue_halo(s + 1, hb ^ 1);                             // the other buffer was last read in slab s-1: free for everyone
            }
            if (GN && t == 3 && s + 1 < s_end) gn_apply_slab(hb ^ 1);   // my table load and halo pieces of slab s+1 landed at tap 2's wait
            if (k + 3 < nk) issue_b();
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = as_half8(ld16(rdB + j * 1024));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // output pixel block i of this wave: 16 pixels of one image row; literal offset of its tap-(ky,kx) halo pixels
                const int pi = i * 16;                                   // (wm0 % W + 16 i) stays inside the row: W % 16 == 0 and wm0 % 16 == 0
                const int oy = (W >= 64) ? 0 : pi / W, ox = (W >= 64) ? pi : pi % W;
                fa[i] = as_half8(ld16(rdAs + ((oy + t / 3) * HW2 + ox + t % 3) * 64));
            }
            {
                const int d = st_read == NSTB - 1 ? -(NSTB - 1) * BSTAGE : BSTAGE;
                rdB += d;
                st_read = st_read == NSTB - 1 ? 0 : st_read + 1;
            }
            // my B pieces of step k+1 (and, from tap 2 on, the next slab's halo pieces) have landed; the barrier publishes them
            wait_keep(k + 3 < nk ? 2 : (k + 2 < nk ? 1 : 0), t < 2 && s + 1 < s_end);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ------------------------------------------------ MFMA phase (the partner reads / stages)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (GN)   // in-place form pinned in asm: with the fused-GroupNorm code around, hipcc otherwise renames the accumulators
                              // between the unrolled taps (D != C) and spills them inside this phase
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(fb[j]), "v"(fa[i]));
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (!grp1) __builtin_amdgcn_s_barrier();                            // group 0 waits out group 1's last MFMA phase
    if (GN) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the asm MFMAs are invisible to hipcc's hazard recogniser: let the last ones retire before VALU reads the accumulators
    if constexpr (BN == V5_BN) v5_finish<0, false>(p, acc, smem5, nullptr, nullptr, 0, m0, n0, wm0, wn0, wid, lane, ks, splitk, tn_i);
    else v6_finish<TM, TN>(p, acc, smem5, m0, n0, wm0, wn0, wid, lane, ks, splitk, img, t_in);
}

// =====================================================================================================================
// v7: short-K row-panel GEMM (K = 320: every projection of the level-0 transformer blocks).  One workgroup owns 256 rows and ALL
// of N.  Its A panel never touches LDS: each wave loads the MFMA fragments of its 32 rows once (20 x 16 bytes per lane, 80 VGPRs)
// and keeps them for the whole launch.  W streams through a 2-stage LDS-DMA ring in 80-row tiles (51 KB: the full K of 80 output
// columns), so per 80-column step a wave issues 100 MFMAs against 7 DMA pieces and 50 fragment reads.
// The two wave groups (waves 0-3 / 4-7: the two waves of every SIMD) run HALF A STEP APART, as in v5, but here the second phase of
// a step is its EPILOGUE: while one wave of a SIMD issues the MFMAs of step j the other finishes and stores step j-1 (bias,
// LayerNorm fold, GEGLU, residual, row statistics) — the per-tile prologue + epilogue that costs the 128 x 160 kernel 60 % of its
// time at K = 320 is hidden behind the matrix pipe, and A is read from HBM exactly once.
//     group 0:  | MFMA j   | EPI j    | MFMA j+1 | EPI j+1  | ...
//     group 1:  | (idle)   | MFMA j   | EPI j    | MFMA j+1 | ...           ('|' = s_barrier joining all 8 waves)
//   step j lives in stage j & 1.  Group 0 issues its share of step j+1 in EPI j, group 1 its share of step j+2 in EPI j (the stage is
//   free by then for both); every interval ends with the DMA retired + lgkmcnt(0), so a step is complete one barrier before its first reader.
// The loop is LDS-bandwidth bound (tools/gemm7_phases.py: with an epilogue that staged its tile through LDS an interval took 4200
// clocks against 2000 for the fragment reads + DMA writes alone), so the epilogue stays OUT of LDS: the 16 W rows an MFMA tile
// reads are chosen such that a lane's accumulators of two neighbouring tiles are 8 CONSECUTIVE output columns —
//     tile jj < 4, MFMA index c  <-  W row 32 (jj >> 1) + 8 (c >> 2) + 4 (jj & 1) + (c & 3);   tile 4: row 64 + (c & 3) + 8 ((c >> 2) & 1) + 4 (c >> 3)
// — and results leave as 16-byte (tile pairs) / 8-byte (tile 4) stores straight from the accumulator layout: 64 + 64 + 32 bytes per
// row and step.  W rows are 640 bytes; chunk c of row r sits at physical chunk (c & ~7) | ((c & 7) ^ key(r)), key(r) = (r & 3) | ((r >> 3) & 1) << 2:
// the 8 rows a lane group reads together (r = x, x+1, x+2, x+3, x+8, .. x+11) have 8 distinct keys -> conflict-free ds_read_b128.
// GEGLU: steps alternate value / gate blocks of 80 columns; the value step's result waits as packed fp16 in 20 VGPRs.
// =====================================================================================================================
constexpr int V7_BM = 256, V7_K = 320, V7_KS = V7_K / 32, V7_NB = 80;
constexpr int V7_PIECES = 56, V7_STAGE_BYTES = V7_PIECES * 1024;      // 80 rows x 640 bytes = 50 pieces, padded to 7 per wave

template <bool GEGLU, bool LN>
__global__ __launch_bounds__(512, 2) void gemm7_kernel(const GemmParams p) {
    constexpr int TM = 2, TN = 5;
    __shared__ __attribute__((aligned(16))) char smem7[2 * V7_STAGE_BYTES];
#ifdef LD_AB_BUILD
    // phase timestamps of waves 0 and 4 of one workgroup (tools/gemm7_phases.py): [wave group][step < 32][8 stamps], low 32 bits of s_memtime
    __shared__ unsigned ts7[2 * 32 * 8];
    const bool ts_on = (p.dbg & 2048) && blockIdx.x == 9 && (threadIdx.x & 255) == 0;
    auto stamp = [&](int it_, int k) {
        if (ts_on && it_ < 32) ts7[((threadIdx.x >> 8) * 32 + it_) * 8 + k] = (unsigned)__builtin_readcyclecounter();
    };
#else
    auto stamp = [&](int, int) {};
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grp1 = wid >= 4;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * V7_BM, mw = m0 + wid * 32;              // this wave's 32 rows
    const int NS = p.N / V7_NB;                                         // steps (80 W rows each)

    // ---- A fragments: rows mw + 16 i + fr, k = 32 ks + 8 fq .. + 7 (rows past M are clamped; their outputs are never stored)
    // Round 5: requested in the prologue BEHIND the first W pieces and in k-step order, and not waited for there (the prologue's counted wait
    // leaves these 20 loads in flight): the first step's MFMAs start on k-step 0 while the later k-steps of the 164 KB panel are still
    // arriving — before, every workgroup sat through its whole panel load (all 256 at once: ~7 of a 32 us launch) before its first MFMA.
    half8 fa[TM][V7_KS];
    auto load_a = [&]() {
        const half_t* ar[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mw + i * 16 + fr;
            ar[i] = p.A + (long long)(m < p.M ? m : p.M - 1) * p.lda + fq * 8;
        }
#pragma unroll
        for (int ks = 0; ks < V7_KS; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i][ks] = as_half8(ld16(ar[i] + ks * 32));
    };
    // ---- LayerNorm fold (consumer): (mu, rstd) of my rows from the producer's per-part (sum, sum of squares).  The whole affine part of
    // the epilogue,  v = rstd alpha (acc - mu wsum) + bias,  is folded into the accumulators' START value  bias / (rstd alpha) - mu wsum
    // (set at the head of a step's MFMA phase, which has vector-issue slack), so the epilogue is one multiply by rstd alpha.
    float rs_a[TM], inv_a[TM], mu_a[TM];
    auto ln_fill = [&]() {   // (prologue, behind the first W pieces: its loads are waited for at once — together with those pieces, which the prologue needs anyway)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            rs_a[i] = p.alpha;
            mu_a[i] = 0.f;
            if (LN) {
                const int m = mw + i * 16 + fr;
                float s1 = 0.f, s2 = 0.f;
                if (m < p.M) ln_sum_parts(p.ln_stat + (long long)m * 2, (long long)p.ln_rows * 2, p.ln_parts, s1, s2);
                const float mu = s1 * p.ln_inv_c;
                mu_a[i] = mu;
                rs_a[i] = rsqrtf(fmaxf(s2 * p.ln_inv_c - mu * mu, 0.f) + p.ln_eps) * p.alpha;     // (rows past M: finite garbage, never stored)
            }
            inv_a[i] = 1.0f / rs_a[i];
        }
    };
    auto key = [](int r) { return (r & 3) | (((r >> 3) & 1) << 2); };
    // ---- W loader: piece (wid + 8 i) of a stage, lane l -> LDS byte o = piece * 1024 + 16 l -> row o / 640, physical chunk (o % 640) / 16
    unsigned w_off[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int o = (wid + 8 * i) * 1024 + lane * 16;
        int row = o / 640;
        const int pc = (o - row * 640) >> 4;
        const int lc = (pc & ~7) | ((pc & 7) ^ key(row));
        if (row > V7_NB - 1) row = V7_NB - 1;                           // the 6 padding pieces re-read the last row (never read back)
        w_off[i] = (unsigned)(((long long)row * p.ldw + lc * 8) * 2);
    }
    // piece 50 (the first padding piece, issued by wave 2) carries the step's 80 bias halfs (bytes 51200 ..) and 80 LayerNorm-fold row
    // sums (bytes 51456 ..): the epilogue reads them from LDS instead of waiting on small global loads every step
    const char* aux_ptr = reinterpret_cast<const char*>(g_zero_row);
    int aux_step = 0;
    if (lane < 10 && p.bias_n != nullptr) {
        aux_ptr = reinterpret_cast<const char*>(p.bias_n + lane * 8);
        aux_step = V7_NB * 2;
    } else if (lane >= 16 && lane < 36 && p.ln_wsum != nullptr) {
        aux_ptr = reinterpret_cast<const char*>(p.ln_wsum + (lane - 16) * 4);
        aux_step = V7_NB * 4;
    }
    const char* p7 = wid == 2 ? aux_ptr : reinterpret_cast<const char*>(p.W) + w_off[6];   // every wave's 7th piece, as a per-lane pointer
    const long long step7 = wid == 2 ? (long long)aux_step : (long long)V7_NB * p.ldw * 2;
    // Step order: workgroup b walks the N / 80 steps starting at step j0(b) and wraps, so that the workgroups of an XCD (b, b + 8, ..)
    // do not all ask its L2 for the same W lines at the same moment.  GEGLU rotates by (value, gate) pairs.
    const int j0 = GEGLU ? 2 * (int)((blockIdx.x >> 3) % (unsigned)(NS >> 1)) : (int)((blockIdx.x >> 3) % (unsigned)NS);
    const unsigned smem_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)smem7);
    int n_issued = 0;
    auto issue = [&]() {
        int js = j0 + n_issued;
        if (js >= NS) js -= NS;
        const half_t* w_base = p.W + (long long)js * V7_NB * p.ldw;      // wave-uniform: W row block of step js
        const unsigned dst = smem_base + (unsigned)((n_issued & 1) * V7_STAGE_BYTES) + (unsigned)wid * 1024u;
#pragma unroll
        for (int i = 0; i < 6; ++i) glds16s(w_off[i], w_base, dst + (unsigned)(8 * i) * 1024u);
        glds16(reinterpret_cast<const half_t*>(p7 + js * step7), dst + 48u * 1024u);   // per-lane pointer form: wave 2 fetches bias / row sums here
        ++n_issued;
    };
    // fragment read bases: the lane that supplies MFMA index fr reads W row  const(jj) + 8 (fr >> 2) + (fr & 3)  (tiles 0..3) or
    // 64 + (fr & 3) + 8 ((fr >> 2) & 1) + 4 (fr >> 3)  (tile 4); both have key (fr & 3) | ((fr >> 2) & 1) << 2
    const int kf = (fr & 3) | (((fr >> 2) & 1) << 2);
    const char* rdP = smem7 + (8 * (fr >> 2) + (fr & 3)) * 640;          // + (32 (jj >> 1) + 4 (jj & 1)) * 640 per tile
    const char* rdL = smem7 + (64 + (fr & 3) + 8 * ((fr >> 2) & 1) + 4 * (fr >> 3)) * 640;
    int chunk_lo[2];                                                     // (fq ^ key) and ((4 + fq) ^ key): the low 3 bits for even / odd ks
    chunk_lo[0] = ((fq ^ kf) & 7) << 4;
    chunk_lo[1] = (((4 + fq) ^ kf) & 7) << 4;
    // ---- epilogue addressing (accumulator layout): rows mw + 16 i + fr; tile pair P -> columns 32 P + 8 fq .. + 7, tile 4 -> 64 + c8 .. + 3
    const int c8 = 64 + 8 * (fq & 1) + 4 * (fq >> 1);
    int o_c[TM], o_r[TM];                                                // element offsets relative to (row mw, column n_out)
    bool row_ok[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mw + 16 * i + fr;
        row_ok[i] = m < p.M;
        o_c[i] = (16 * i + fr) * p.ldc;
        o_r[i] = ((row_ok[i] ? m : p.M - 1) - mw) * p.ldr;
    }

    f32x4 acc[TM][TN];
    unsigned vh[GEGLU ? TM * TN * 2 : 1];                               // GEGLU: the finished value block, packed fp16, waits for its gate block

    // ---- one MFMA phase: this step's 80 W rows x K = 320 against my A fragments
    auto mfma_step = [&](int stage) {
        const char* TP = rdP + stage * V7_STAGE_BYTES;
        const char* TL = rdL + stage * V7_STAGE_BYTES;
        auto rd = [&](int j, int ks) {
            const int co = ((ks >> 1) << 7) + chunk_lo[ks & 1];
            return as_half8(ld16(j < 4 ? TP + (32 * (j >> 1) + 4 * (j & 1)) * 640 + co : TL + co));
        };
        half8 fb[2][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = rd(j, 0);
        {   // accumulators start at  bias / (rstd alpha) - mu wsum  of my columns (tile jj < 4 -> 32 (jj >> 1) + 8 fq + 4 (jj & 1) .. + 3, tile 4 -> c8 .. + 3)
            const char* aux = smem7 + stage * V7_STAGE_BYTES + 50 * 1024;   // this step's bias (halfs; zeros when there is none) and, 256 bytes on, LayerNorm-fold row sums (floats)
            const uint4 b01 = ld16(aux + (8 * fq) * 2), b23 = ld16(aux + (32 + 8 * fq) * 2);
            const uint2 b4 = *reinterpret_cast<const uint2*>(aux + c8 * 2);
            const uint2 bt[TN] = {make_uint2(b01.x, b01.y), make_uint2(b01.z, b01.w), make_uint2(b23.x, b23.y), make_uint2(b23.z, b23.w), b4};
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                const half4 bh = __builtin_bit_cast(half4, bt[jj]);
                f32x4 bf, ws = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) bf[r] = (float)bh[r];
                if (LN) ws = *reinterpret_cast<const f32x4*>(aux + 256 + (jj < 4 ? 32 * (jj >> 1) + 8 * fq + 4 * (jj & 1) : c8) * 4);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][jj] = LN ? bf * inv_a[i] - mu_a[i] * ws : bf * inv_a[i];
            }
        }
#pragma unroll
        for (int ks = 0; ks < V7_KS; ++ks) {
#ifdef LD_AB_BUILD
            if ((p.dbg & 128) && ks > 0) break;
#endif
            if (ks + 1 < V7_KS) {
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[(ks + 1) & 1][j] = rd(j, ks + 1);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[ks & 1][j], fa[i][ks], acc[i][j], 0, 0, 0);
        }
    };
    // group 0: its share of step j+1 (stage free since group 1's MFMA j-1); group 1: its share of step j+2 (stage free since its own MFMA j)
    auto issue_next = [&](int j) {
        if (!grp1) {
            if (j + 1 < NS) issue();
        } else {
            if (j + 2 < NS) issue();
        }
    };
    const bool full_tile = m0 + V7_BM <= p.M;
    // ---- one epilogue phase: step j (columns n_out .. n_out + 79 of the output; W / bias rows nb .. nb + 79); returns the number of
    // store instructions it left as the youngest vector-memory operations of this wave.  Branch-free, every LDS / global read of a
    // phase issued as one batch: tile-by-tile read-wait-convert chains measured at twice the MFMA phase they are meant to hide behind.
    auto epilogue = [&](int it_, int j) -> int {                         // it_: position in this workgroup's walk (stage parity), j: the step
#ifdef LD_AB_BUILD
        if (p.dbg & 256) {
            issue_next(it_);
            return 0;
        }
#endif
        const int nb = j * V7_NB;                                        // row block of W / bias / wsum
        const int n_out = GEGLU ? (j >> 1) * V7_NB : nb;
        // The W pieces of a later step go out FIRST: they then have the whole epilogue to land, and the closing wait of the interval still
        // finds them older than this epilogue's output stores.  (The stage they overwrite is free: see issue_next; the bias / row-sum
        // piece this epilogue reads belongs to wave 2's share, which group 0 re-issues one interval later.)
        issue_next(it_);
        const bool has_res = !GEGLU && p.R != nullptr;
        uint4 r16[TM][2];                                                // residual, requested now, added after the activation
        uint2 r8[TM];
        if (has_res) {                                                   // (uniform; rows past M read row M - 1, their results are never stored; GEGLU: no residual here)
            const half_t* Rb = p.R + (long long)mw * p.ldr + n_out;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                r16[i][0] = ld16(Rb + o_r[i] + 8 * fq);
                r16[i][1] = ld16(Rb + o_r[i] + 32 + 8 * fq);
                r8[i] = *reinterpret_cast<const uint2*>(Rb + o_r[i] + c8);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                r16[i][0] = r16[i][1] = zero16();
                r8[i] = make_uint2(0u, 0u);
            }
        }
        auto affine = [&](int i, int jj) { return acc[i][jj] * rs_a[i]; };   // (bias and the LayerNorm shift went into the accumulators' start value)
        if (GEGLU && (j & 1) == 0) {                                     // value block: park it (one uniform branch, not one per tile:
#pragma unroll                                                           //  the gate step below must stay ONE basic block, see there)
            for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const f32x4 v = affine(i, jj);
                    vh[(i * TN + jj) * 2] = pk2h(v[0], v[1]);
                    vh[(i * TN + jj) * 2 + 1] = pk2h(v[2], v[3]);
                }
            stamp(it_, 3);
            return 0;
        }
        uint2 h[TM][TN];
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            f32x4 v[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) v[i] = affine(i, jj);
            if (GEGLU) {                                                 // gate step: out = value * gelu(gate), 8 at a time (common.h: geglu8_staged)
                const unsigned aw[4] = {vh[jj * 2], vh[jj * 2 + 1], vh[(TN + jj) * 2], vh[(TN + jj) * 2 + 1]};
                const f32x2 gp[4] = {{v[0][0], v[0][1]}, {v[0][2], v[0][3]}, {v[1][0], v[1][1]}, {v[1][2], v[1][3]}};
                unsigned ow[4];
#ifdef LD_AB_BUILD
                if (p.dbg & 512) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const half2v ah = __builtin_bit_cast(half2v, aw[k]);
                        ow[k] = pk2h((float)ah[0] * gp[k][0], (float)ah[1] * gp[k][1]);
                    }
                } else
#endif
                    geglu8_staged(aw, gp, ow);
                h[0][jj] = make_uint2(ow[0], ow[1]);
                h[1][jj] = make_uint2(ow[2], ow[3]);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    h[i][jj].x = pk2h(v[i][0], v[i][1]);
                    h[i][jj].y = pk2h(v[i][2], v[i][3]);
                }
            }
        }
        stamp(it_, 3);
        // Vector-memory operations retire in order: the residual is waited for with the builtin (which hipcc's waitcnt pass models: it
        // then adds no wait of its own), unconditionally (under `if (R)` the model still holds the loads outstanding on the merged path
        // and parks its own waits further down).  The W pieces issued above are older and retire with it — they have had the whole
        // finish to land; the output stores below stay the youngest operations, so the interval's closing wait can leave them in flight.
        if (!GEGLU) __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
        stamp(it_, 4);
        uint4 o16[TM][2];
        uint2 o8[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            o16[i][0] = make_uint4(h[i][0].x, h[i][0].y, h[i][1].x, h[i][1].y);
            o16[i][1] = make_uint4(h[i][2].x, h[i][2].y, h[i][3].x, h[i][3].y);
            o8[i] = h[i][4];
            if (has_res) {                                               // packed fp16 adds: the finished tile is fp16 already
                o16[i][0] = add8h(o16[i][0], r16[i][0]);
                o16[i][1] = add8h(o16[i][1], r16[i][1]);
                const uint4 t = add8h(make_uint4(o8[i].x, o8[i].y, 0u, 0u), make_uint4(r8[i].x, r8[i].y, 0u, 0u));
                o8[i] = make_uint2(t.x, t.y);
            }
        }
        if (!GEGLU && p.stat_out != nullptr) {   // LN-fold producer: (sum, sum of squares) of the fp16 results per row: 20 columns per lane, then across the 4 lanes of a row
            const half2v one2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned w[10] = {o16[i][0].x, o16[i][0].y, o16[i][0].z, o16[i][0].w, o16[i][1].x, o16[i][1].y, o16[i][1].z, o16[i][1].w, o8[i].x, o8[i].y};
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 10; ++e) {
                    const half2v hv = __builtin_bit_cast(half2v, w[e]);
                    s1 = __builtin_amdgcn_fdot2(hv, one2, s1, false);
                    s2 = __builtin_amdgcn_fdot2(hv, hv, s2, false);
                }
                // lanes fr, fr + 16, fr + 32, fr + 48 hold one row: two swap-and-add steps leave the row total in all four
                auto a1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
                auto a2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
                s1 = __uint_as_float(a1[0]) + __uint_as_float(a1[1]);
                s2 = __uint_as_float(a2[0]) + __uint_as_float(a2[1]);
                a1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
                a2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
                s1 = __uint_as_float(a1[0]) + __uint_as_float(a1[1]);
                s2 = __uint_as_float(a2[0]) + __uint_as_float(a2[1]);
                if (fq == 0 && row_ok[i]) *reinterpret_cast<float2*>(p.stat_out + ((long long)j * p.M + mw + 16 * i + fr) * 2) = make_float2(s1, s2);
            }
        }
#ifdef LD_AB_BUILD
        if (p.dbg & 1024) return 0;                                      // ablation: no output stores
#endif
        half_t* Cb = p.C + (long long)mw * p.ldc + n_out;
        if (full_tile) {                                                 // exactly 6 store instructions: the closing wait leaves them in flight
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                st16(Cb + o_c[i] + 8 * fq, o16[i][0]);
                st16(Cb + o_c[i] + 32 + 8 * fq, o16[i][1]);
                *reinterpret_cast<uint2*>(Cb + o_c[i] + c8) = o8[i];
            }
            return 6;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
            if (row_ok[i]) {
                st16(Cb + o_c[i] + 8 * fq, o16[i][0]);
                st16(Cb + o_c[i] + 32 + 8 * fq, o16[i][1]);
                *reinterpret_cast<uint2*>(Cb + o_c[i] + c8) = o8[i];
            }
        return 0;
    };
    auto end_interval = [&](int keep_stores) {
        if (keep_stores == 6) wait_vmcnt<6>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: step 0 (everyone) and group 1's share of step 1 in flight; step 0 landed and published; group 1 one barrier behind
    issue();
    if (grp1 && NS > 1) issue();
    ln_fill();
    load_a();
    // every W piece issued above (and the LayerNorm statistics) is older than the TM * V7_KS loads of the A panel: this leaves exactly those in flight
    wait_vmcnt<TM * V7_KS>();
    __builtin_amdgcn_s_barrier();
    if (grp1) __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int it_ = 0; it_ < NS; ++it_) {
        int j = j0 + it_;
        if (j >= NS) j -= NS;
        stamp(it_, 0);
        mfma_step(it_ & 1);
        stamp(it_, 1);
        end_interval(st);                                                // my DMA share is older than the last epilogue's stores: those may stay in flight
        stamp(it_, 2);
        // The epilogue runs at raised priority: on this chip a VALU stream and an MFMA stream of the two waves of a SIMD take the SUM of
        // their times when the MFMA wave has (equal or higher) priority — it holds the vector issue port while the matrix pipe is busy —
        // and the MAX when the VALU wave has priority (tools/micro/coexec.hip, profiles/README.md).
        __builtin_amdgcn_s_setprio(2);
        st = epilogue(it_, j);
        __builtin_amdgcn_s_setprio(0);
        stamp(it_, 5);
        end_interval(st);
        stamp(it_, 6);
    }
    if (!grp1) __builtin_amdgcn_s_barrier();                            // group 0 waits out group 1's last epilogue: every wave ran 2 NS + 2 barriers
#ifdef LD_AB_BUILD
    if ((p.dbg & 2048) && blockIdx.x == 9 && p.partial != nullptr) {
        __syncthreads();
        unsigned* out = reinterpret_cast<unsigned*>(p.partial);
        for (int q = threadIdx.x; q < 2 * 32 * 8; q += 512) out[q] = ts7[q];
    }
#endif
}

// split-K second pass: sum the fp32 slabs and run the same epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, int bn) {
    const int out_n = p.act == 2 ? p.N / 2 : p.N;
    const int cpr = out_n / 8;
    const long long total = (long long)p.M * cpr;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(q / cpr), cc = (int)(q - (long long)m * cpr);
        float v[8];
        if (p.act == 2) {
            const int half_bn = bn / 2;
            const int no = cc * 8;
            const int tile = no / half_bn, within = no - tile * half_bn;
            const int nv = tile * bn + within, ng = nv + half_bn;
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int s = 0; s < p.splitk; ++s) {
                const float* base = p.partial + ((long long)s * p.M + m) * p.N;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a[j] += base[nv + j];
                    g[j] += base[ng + j];
                }
            }
            float ba[8], bg[8];
            unpack8(ld16(p.bias_n + nv), ba);
            unpack8(ld16(p.bias_n + ng), bg);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (a[j] + ba[j]) * gelu_f(g[j] + bg[j]);
            epilogue_store8(p, 0, m, no, 0, v);
        } else {
            const int n = cc * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            // four slabs per batch of loads, summed in slab order (a split of 16 read one slab after the other is 16 memory latencies in a
            // row: the reduce launches of the batch-1 step were latency-, not bandwidth-bound)
            const float* base = p.partial + (long long)m * p.N + n;
            const long long slab = (long long)p.M * p.N;
            int s = 0;
            for (; s + 4 <= p.splitk; s += 4) {
                f32x4 x0[4], x1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x0[u] = *reinterpret_cast<const f32x4*>(base + (s + u) * slab);
                    x1[u] = *reinterpret_cast<const f32x4*>(base + (s + u) * slab + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] += x0[u][j];
                        v[4 + j] += x1[u][j];
                    }
            }
            for (; s < p.splitk; ++s) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + s * slab), x1 = *reinterpret_cast<const f32x4*>(base + s * slab + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] += x0[j];
                    v[4 + j] += x1[j];
                }
            }
            epilogue_store8(p, 0, m, n, n, v);
        }
    }
}

// split-K second pass that also emits the GroupNorm partial statistics of its output (GemmParams::gn_part): block = (pixel chunk, image,
// slab of 8 groups), a thread owns one 8-channel chunk and strides over the chunk's pixels — thread mapping, LDS layout and summation
// order are those of gn_stats_kernel (norm.hip), so the partials (and the normalised tensor) are bit-identical to the two-launch path.
__global__ __launch_bounds__(256) void splitk_reduce_gn_kernel(const GemmParams p) {
    __shared__ float csum[2048], csq[2048];   // [rows_par][slab channels]
    const int C = p.N, cpg = C / 32;
    const int CS = C / 4, CHS = CS >> 3;
    const int n = blockIdx.y, pc = blockIdx.x, slab = blockIdx.z, tid = threadIdx.x;
    const int rows_par = 256 / CHS;
    const int cc = tid % CHS, prow = tid / CHS;
    const int c0 = slab * CS + cc * 8;
    const int p_begin = pc * p.gn_ppb, p_end = min(p.gn_HW, p_begin + p.gn_ppb);
    const long long slab_stride = (long long)p.M * p.N;
    if (prow < rows_par) {
        float s[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
        for (int pix = p_begin + prow; pix < p_end; pix += rows_par) {
            const int m = n * p.gn_HW + pix;
            const float* base = p.partial + (long long)m * p.N + c0;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            int sp = 0;
            for (; sp + 4 <= p.splitk; sp += 4) {
                f32x4 x0[4], x1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x0[u] = *reinterpret_cast<const f32x4*>(base + (sp + u) * slab_stride);
                    x1[u] = *reinterpret_cast<const f32x4*>(base + (sp + u) * slab_stride + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] += x0[u][j];
                        v[4 + j] += x1[u][j];
                    }
            }
            for (; sp < p.splitk; ++sp) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + sp * slab_stride), x1 = *reinterpret_cast<const f32x4*>(base + sp * slab_stride + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] += x0[j];
                    v[4 + j] += x1[j];
                }
            }
            epilogue_store8(p, 0, m, c0, c0, v);                       // (v comes back as the values before the fp16 rounding)
            float f[8];
            unpack8(pack8(v), f);                                      // statistics of what was stored: the fp16 values, as gn_stats_kernel reads them
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += f[j];
                ss[j] += f[j] * f[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            csum[prow * CS + cc * 8 + j] = s[j];
            csq[prow * CS + cc * 8 + j] = ss[j];
        }
    }
    __syncthreads();
    {
        const int g = tid >> 5, sub = tid & 31;                  // 8 groups x 32 lanes
        const int cnt = rows_par * cpg;
        float a = 0.f, b = 0.f;
        for (int i = sub; i < cnt; i += 32) {
            const int pr = i / cpg, c = g * cpg + (i - pr * cpg);
            a += csum[pr * CS + c];
            b += csq[pr * CS + c];
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (sub == 0) {
            float* o = p.gn_part + (((long long)n * p.gn_P + pc) * 32 + slab * 8 + g) * 2;
            o[0] = a;
            o[1] = b;
        }
    }
}

// plain GEMMs with at most this many workgroups take the producer/consumer kernel (measured at B=1: 256 -> 169.3, 512 -> 167.8,
// 768 -> 166.2, 1280 -> 164.8 steps/s)
constexpr int V4_MAX_BLOCKS = 256;

// name of the kernel instantiation the last gemm_launch on this thread dispatched (ld_unet_profile groups by it)
thread_local const char* t_last_kernel = "";
const char* intern_name(const std::string& s) {   // stable storage for composed names (a handful per process)
    static std::mutex mu;
    static std::set<std::string> pool;
    std::lock_guard<std::mutex> lock(mu);
    return pool.insert(s).first->c_str();
}

static bool two_wg_ok();   // (A/B hook, below)
static bool m_fastest_ok();
static bool skinny_conv1_ok();

// conv1_2wg: gemm_launch's decision to run a 1x1 convolution on 64 x 64 tiles UNSPLIT on the two-workgroups-per-CU producer / consumer kernel
// (its skinny_conv1 rule and the older few-tile rule behind the same A/B switch) — decided THERE, not re-derived here
template <int BM, int BN>
void launch_cfg(const GemmParams& p, hipStream_t s, bool deep = false, bool conv1_2wg = false) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int sk = p.splitk > 1 ? p.splitk : 1;
    dim3 grid(tiles * sk, 1, p.batch);
    // producer/consumer kernel: wins where a plain GEMM leaves at most one workgroup per CU (batch-1 step: +5.6 % whole step,
    // same box A/B); loses on convs and wherever two v3 workgroups share a CU.
    if constexpr (BM == 64 && BN == 64) {
        // skinny projections with more tiles than CUs: the producer / consumer kernel with two workgroups per CU (see gemm4_kernel, WPS).
        // Measured per launch inside the batch-1 forward (tools/ab_launches.py 1 0 2048, profiles/r05_ab_gemm4_rings.txt): 2048 x 640 x 640
        // (320 tiles, 10 slabs) 14.3 -> 11.7 us against the 2-stage kernel; 8192 x 320 x 320 (640 tiles, 5 slabs) 12.3 -> 14.4: short K stays.
        // An 8-stage ring for <= 256 tiles measured +-0 (512 x 1280 x 1280: 13.6 vs 13.4 us): these launches are not short of bytes in
        // flight — an ablated kernel that only runs its prologue and barriers takes 4.7 of 8.3 us (tools/gemm4_abl.py).
        const long long blocks = (long long)tiles * sk * p.batch;
        if (!p.conv && two_wg_ok() && blocks > V4_MAX_BLOCKS && blocks <= 512 && p.K >= 640) {
            t_last_kernel = "gemm4_kernel<64,64,plain,2wg>";
            hipLaunchKernelGGL((gemm4_kernel<64, 64, false, 4, 4>), grid, dim3(2 * NT), 0, s, p);
            return;
        }
        if (conv1_2wg) {
            t_last_kernel = "gemm4_kernel<64,64,conv,2wg>";
            hipLaunchKernelGGL((gemm4_kernel<64, 64, true, 4, 4>), grid, dim3(2 * NT), 0, s, p);
            return;
        }
    }
    if (!p.conv && (long long)tiles * sk * p.batch <= V4_MAX_BLOCKS) {
        static const std::string name = "gemm4_kernel<" + std::to_string(BM) + "," + std::to_string(BN) + ",plain>";
        t_last_kernel = name.c_str();
        hipLaunchKernelGGL((gemm4_kernel<BM, BN, false>), grid, dim3(2 * NT), 0, s, p);
    } else if (p.conv) {
        static const std::string name = "gemm3_kernel<" + std::to_string(BM) + "," + std::to_string(BN) + ",conv>";
        t_last_kernel = name.c_str();
        if constexpr (BM == 64 && BN == 160) {
            if (deep) {   // 4-stage ring, one workgroup per CU (three slabs in flight): see the rule in gemm_launch
                hipLaunchKernelGGL((gemm3_kernel<BM, BN, true, 4>), grid, dim3(NT), 0, s, p);
                return;
            }
        }
        hipLaunchKernelGGL((gemm3_kernel<BM, BN, true, 2>), grid, dim3(NT), 0, s, p);
    } else {
        static const std::string name = "gemm3_kernel<" + std::to_string(BM) + "," + std::to_string(BN) + ",plain>";
        t_last_kernel = name.c_str();
        hipLaunchKernelGGL((gemm3_kernel<BM, BN, false, 2>), grid, dim3(NT), 0, s, p);
    }
}

}  // namespace

#ifdef LD_AB_BUILD
// tuning hook of the A/B build (tools/gemm_sweep.py): force tile height / split-K for every following launch; 0 = automatic.
// Not part of the shipped library: `make AB=1` builds libld_mi355x_ab.so with it.
static int g_force_bm = 0, g_force_sk = 0, g_no_v5 = 0;
// per-shape override (tools/ab_shape.py: candidates are timed launch by launch INSIDE the forward): M, N, K -> tile height, split
static int g_shape_ovr[5] = {0, 0, 0, 0, 0};
extern "C" void ld_debug_gemm_shape_override(int M, int N, int K, int bm, int splitk) {
    g_shape_ovr[0] = M; g_shape_ovr[1] = N; g_shape_ovr[2] = K; g_shape_ovr[3] = bm; g_shape_ovr[4] = splitk;
}
extern "C" void ld_debug_gemm_override(int bm, int splitk) {
    g_force_bm = bm;
    g_force_sk = splitk;
}
extern "C" void ld_debug_gemm_no_v5(int off) { g_no_v5 = off; }
static int g_v5_dbg = 0;
extern "C" void ld_debug_gemm_v5_dbg(int bits) { g_v5_dbg = bits; }   // 1: no DMA issue in the loop, 2: no fragment reads, 4: 4 of 40 MFMAs
#endif

namespace {
bool skinny_conv1_ok() {
#ifdef LD_AB_BUILD
    return (g_no_v5 & 8192) == 0;    // A/B: bit 8192 keeps the split 64 x 160 tiles for the skinny 1x1 convolutions
#else
    return true;
#endif
}
bool m_fastest_ok() {
#ifdef LD_AB_BUILD
    return (g_no_v5 & 4096) == 0;    // A/B: bit 4096 keeps the N-fastest tile order everywhere
#else
    return true;
#endif
}
bool two_wg_ok() {
#ifdef LD_AB_BUILD
    return (g_no_v5 & 2048) == 0;    // A/B: bit 2048 keeps round 4's 2-stage kernel for the 64 x 64 tiles beyond 256 workgroups
#else
    return true;
#endif
}
}  // namespace

// the split-K second pass of a launch: the plain reduce, or the one that also emits GroupNorm partials (GemmParams::gn_part)
static void launch_splitk_reduce(const GemmParams& p, int bn, hipStream_t stream) {
    // (N >= 256: below that gn_stats_kernel uses fewer, wider channel slabs — norm.hip gn_slabs — and this kernel's four-slab order would
    // no longer reproduce its partials bit for bit)
    const bool gn = p.gn_part != nullptr && p.act != 2 && p.batch == 1 && p.gn_P > 0 && p.gn_HW > 0 && p.M % p.gn_HW == 0 && p.N % 32 == 0 && p.N <= 8192 && p.N >= 256 &&
                    p.ldc == p.N && (p.N / 4) % 8 == 0 && p.N / 32 <= 256;
    if (gn) {
        hipLaunchKernelGGL(splitk_reduce_gn_kernel, dim3(p.gn_P, p.M / p.gn_HW, 4), dim3(256), 0, stream, p);
        if (p.gn_part_done != nullptr) *p.gn_part_done = p.gn_P;
        t_last_kernel = intern_name(std::string(t_last_kernel) + "+splitk_reduce_gn_kernel");
        return;
    }
    const int out_n = p.act == 2 ? p.N / 2 : p.N;
    const long long total = (long long)p.M * (out_n / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p, bn);
    t_last_kernel = intern_name(std::string(t_last_kernel) + "+splitk_reduce_kernel");
}

bool gemm_ln_fold_available() { return true; }
const char* gemm_last_kernel_name() { return t_last_kernel; }

// Measured (tools/gemm5_ab.py AB_MODE=v7, tools/ab_launches.py; profiles/README.md): against the 128 x 160 kernel the row-panel kernel runs the
// plain K = 320 projections 1.3..1.6x faster (27 vs 36 us at 65536 x 320 with residual, 37 vs 60 us at 65536 x 640) and, since its epilogue
// left LDS and its GELUs are issued interleaved (common.h: geglu8_staged), the level-0 GEGLU as well.
static bool v7_geglu_enabled() {
#ifdef LD_AB_BUILD
    return (g_no_v5 & 16) == 0;      // A/B: bit 16 sends GEGLU back to the 128 x 160 kernel
#else
    return true;
#endif
}

// smallest K of a GEGLU the 256 x 320 kernel takes (round 4: 1280 — its strip epilogue ran the GELUs unbatched)
static int v5_geglu_min_k() {
#ifdef LD_AB_BUILD
    return (g_no_v5 & 1024) ? 640 : 1280;      // A/B: bit 1024 also sends the K = 640 GEGLU (level 1) there
#else
    return 1280;
#endif
}

// does this convolution run on the halo-tile kernel (v6), and with which tile (rows x columns), split over K and loader (UP)?
struct V6Plan {
    int sk = 1, bn = 0, bm = V5_BM, wc = 0;   // wc: tile width in pixels (16 / 32 / 64 / 128; an image wider than 128 is cut into 128-pixel bands)
    bool up = false;
};
static bool v6_plan(const GemmParams& p, V6Plan* out) {
    const bool same = p.Hv == p.Hs && p.Wv == p.Ws;
    const bool up = p.Hv == 2 * p.Hs && p.Wv == 2 * p.Ws;           // exact nearest 2x (Upsample: LD.py:3498-3511; Upsample1 when the skip is 2x)
    if (!(p.conv && p.ksize == 3 && p.stride == 1 && (p.pad < 0 || p.pad == 1) && (same || up) && p.Ho == p.Hv && p.Wo == p.Wv &&
          p.C1 % 32 == 0 && p.C2 % 32 == 0 && p.bm == 0 && p.bn == 0 && p.splitk == 0 && p.batch == 1 && p.act != 2))
        return false;
    if (up && (p.C2 != 0 || p.gn_scale != nullptr)) return false;
    if (p.SC1 > 0) return false;                                     // (a second K segment: the tap-major kernels)
    const int wc = (p.Wo == 16 || p.Wo == 32 || p.Wo == 64 || p.Wo == 128) ? p.Wo : (p.Wo > 128 && p.Wo % 128 == 0) ? 128 : 0;
    if (wc == 0) return false;
#ifdef LD_AB_BUILD
    if (g_no_v5 & 2) return false;
    if ((g_no_v5 & 256) && (up || p.Wo > 128)) return false;      // A/B: round-2 coverage only
#endif
    // tile: 320 columns for the UNet's N = 320 k; 256 for the VAE's N = 256 / 512 at >= 64-pixel rows (VAE decode b=8 29.9 -> 28.8 ms);
    // 512 pixels x 128 columns for its N = 128 (a 256 x 128 tile would leave a wave 16 MFMAs per phase: the step's fixed cost dominates).
    // (256 x 160 tiles for level 1 — 256 unsplit tiles instead of 128 + split — measured a net loss per launch inside the forward:
    // 16384 x 640 x 5760 737 vs 748 us per 6 launches, but 317 vs 291 / 217 vs 213 / 176 vs 170 us at K = 17280 / 11520 / 8640: a
    // 64 x 80 wave tile reads 1.3x the LDS bytes per MFMA and halves the MFMAs a step's fixed cost is spread over.)
    int bn = 0, bm = V5_BM;
    if (p.N % V5_BN == 0) {
        bn = V5_BN;
    } else if (p.N % 256 == 0 && wc >= 64 && (p.gn_scale == nullptr || (p.N == 256 && wc == 128 && !up))   // (fused GroupNorm: one tile wide, 128-pixel bands)
#ifdef LD_AB_BUILD
               && !(g_no_v5 & 64)
#endif
    ) {
        bn = 256;
    } else if (p.N % 128 == 0 && wc == 128 && (p.gn_scale == nullptr || (p.N == 128 && !up))
#ifdef LD_AB_BUILD
               && !(g_no_v5 & 256)
#endif
    ) {
        bn = 128;
        bm = 512;
    } else if (p.N == 32 && wc == 128 && p.gn_scale == nullptr && !up) {
        // a <= 8-channel output convolution (the VAE's conv_out, weights zero-padded to 32 rows by the caller, n_valid = 8): 8 MFMAs per
        // wave and phase — the step's fixed cost dominates, but the halo tile is read once instead of nine times per pixel
        bn = 32;
        bm = 512;
    } else {
        return false;
    }
    if (p.Ho % (bm / wc) != 0 || p.M % bm != 0) return false;       // whole tiles: bm / wc image rows each
    const long long tm = p.M / bm;
    const long long t6 = tm * (p.N / bn);
    const int NS = (p.C1 + p.C2) / 32;
    int sk6 = 1;
    // (split over K only from K = 8640 on: per launch inside the UNet forward (tools/ab_launches.py) the split + reduce pair loses to
    // the unsplit 128 x 160 kernel at 16384 x 640 x 5760 — 133 vs 124 us — and wins from 8640 on: 171 vs 176, 208 vs 227, 285 vs 329 us)
    if (t6 < 192 && p.partial != nullptr && p.K >= 8640) {
        sk6 = (int)((256 + t6 - 1) / t6);
        const int cap = p.K / 2560;
        if (sk6 > cap) sk6 = cap;
        if (sk6 > NS) sk6 = NS;
        while (sk6 > 1 && (size_t)sk6 * p.M * p.N * sizeof(float) > p.partial_bytes) --sk6;
    }
#ifdef LD_AB_BUILD
    if (p.partial != nullptr && bn == V5_BN) {   // split-factor sweep of the halo convolution (tools/conv6_split_sweep.py): LD_V6_SK_W<width> = slices over K
        static const char* names[4] = {"LD_V6_SK_W16", "LD_V6_SK_W32", "LD_V6_SK_W64", "LD_V6_SK_W128"};
        const char* e = getenv(names[wc == 16 ? 0 : wc == 32 ? 1 : wc == 64 ? 2 : 3]);
        if (e != nullptr) {
            int want = atoi(e);
            if (want < 1) want = 1;
            if (want > NS) want = NS;
            while (want > 1 && (size_t)want * p.M * p.N * sizeof(float) > p.partial_bytes) --want;
            sk6 = want;
        }
    }
#endif
    out->sk = sk6;
    out->bn = bn;
    out->bm = bm;
    out->wc = wc;
    out->up = up;
    return t6 * sk6 >= 192;
}

bool gemm_conv_takes_skip_segment(const GemmParams& p) {
    V6Plan pl;
    return p.conv && p.ksize == 3 && p.stride == 1 && p.Hv == p.Hs && p.Wv == p.Ws && p.SC1 == 0 && !conv8_plan(p, nullptr) && !v6_plan(p, &pl);
}

bool gemm_conv_takes_halo_tile(const GemmParams& pin) {
    GemmParams p = pin;
    V6Plan pl;
    return v6_plan(p, &pl);
}

bool gemm_conv_fuses_groupnorm(const GemmParams& p) {
#ifdef LD_AB_BUILD
    if (g_no_v5 & 4) return false;   // A/B: keep the two-pass GroupNorm in front of the halo kernel
#endif
    // Measured (tools/gnconv_ab.py, same process): fused vs two-pass GroupNorm + the same halo conv: +4 % at N = 320 (level 0, one N
    // tile per M tile), +-0 % at N = 640, -3 % at N = 1280 — every N tile of an M tile normalises the same halo again, so the fusion
    // only pays where the output is one tile wide.
    // Round 5: also the VAE decoder's one-tile-wide stages — N = 256 at 256-pixel rows, N = 128 at 512-pixel rows (128-pixel bands) — where a
    // two-pass GroupNorm writes and re-reads 134 - 537 MB per convolution.
    V6Plan pl;
    if (!v6_plan(p, &pl) || pl.up) return false;
    if (p.N == V5_BN && pl.bn == V5_BN) return true;
#ifdef LD_AB_BUILD
    if (g_no_v5 & 16384) return false;   // A/B: the UNet's N = 320 case only
#endif
    // (the 128 -> 3 output convolution — 8 MFMAs per phase — measured a loss with its norm_out fused: VAE decode +0.15 ms, profiles/r05_ab_vae_gn_fused.txt)
    return pl.wc == 128 && ((p.N == 256 && pl.bn == 256 && pl.bm == V5_BM) || (p.N == 128 && pl.bn == 128 && pl.bm == 512));
}

int gemm_launch(const GemmParams& pin, hipStream_t stream) {
    GemmParams p = pin;
#ifdef LD_AB_BUILD
    p.dbg = g_v5_dbg;
    if (g_force_bm) p.bm = g_force_bm;
    if (g_force_sk) p.splitk = g_force_sk;
    if (g_shape_ovr[0] == p.M && g_shape_ovr[1] == p.N && g_shape_ovr[2] == p.K && p.batch == 1 && p.gn_scale == nullptr) {
        p.bm = g_shape_ovr[3];
        p.splitk = g_shape_ovr[4];
    }
#endif
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.A == nullptr || p.W == nullptr || p.C == nullptr) return LD_ERR_ARG;
    if ((p.N & 7) || (p.K & 7) || (p.ldw & 7) || (p.ldc & 7)) return LD_ERR_SHAPE;
    if (p.conv) {
        const int Cin = p.C1 + p.C2;
        if (p.ksize != 1 && p.ksize != 3) return LD_ERR_ARG;
        if (p.pad < 0) p.pad = p.ksize >> 1;
        if (p.SC1 < 0 || p.SC2 < 0 || (p.SC1 == 0 && p.SC2 != 0)) return LD_ERR_ARG;
        if (Cin <= 0 || (p.C1 % 64) || (p.C2 % 64) || (p.SC1 % 64) || (p.SC2 % 64) || p.K != p.ksize * p.ksize * Cin + p.SC1 + p.SC2) return LD_ERR_SHAPE;
        if (p.SC1 > 0) {   // second K segment (gemm.h): raw sources of the output's size, read at the output pixel itself
            if (p.S1 == nullptr || (p.SC2 > 0 && p.S2 == nullptr)) return LD_ERR_ARG;
            if (p.stride != 1 || p.Hv != p.Hs || p.Wv != p.Ws || p.Ho != p.Hv || p.Wo != p.Wv || p.pad != p.ksize / 2 || p.SC1 > 32768 || p.SC2 > 32768) return LD_ERR_SHAPE;
        }
        if (p.C1 > 32768 || p.C2 > 32768) return LD_ERR_SHAPE;   // the stepped zero row (g_zero_row) covers one tap's channel run
        if (p.C2 > 0 && p.A2 == nullptr) return LD_ERR_ARG;
        if (p.M % (p.Ho * p.Wo)) return LD_ERR_SHAPE;
        if (p.batch != 1) return LD_ERR_ARG;
    } else if (p.lda & 7) {
        return LD_ERR_SHAPE;
    }
    if (p.act == 2 && (p.bias_n == nullptr || (p.N & 15))) return LD_ERR_ARG;
    if (p.R != nullptr && (p.ldr & 7)) return LD_ERR_SHAPE;

    if (p.n_valid <= 0 || p.n_valid > p.N) p.n_valid = p.N;
    // ---- conv8 (row-resident, weights streamed once, in-launch slab reduction): the two-image 16x16 / 8x8 levels of a batch-1 step
    if (conv8_plan(p, nullptr)) {
        static const char* names[4][2] = {{"conv8_kernel<W8>", "conv8_kernel<W8,up>"}, {"conv8_kernel<W16>", "conv8_kernel<W16,up>"},
                                          {"conv8_kernel<W32>", "conv8_kernel<W32,up>"}, {"conv8_kernel<W64>", "conv8_kernel<W64,up>"}};
        t_last_kernel = names[p.Wo == 8 ? 0 : p.Wo == 16 ? 1 : p.Wo == 32 ? 2 : 3][p.Hv == 2 * p.Hs ? 1 : 0];
        if (p.gn_part != nullptr && p.gn_part_done != nullptr) *p.gn_part_done = conv8_gn_chunks(p);
        return conv8_launch(p, stream);
    }
    // ---- v6 (halo-tile 3x3 convolution on the v5 skeleton): stride-1 convs whose tiles are whole image rows and fill the chip
    V6Plan pl;
    if (v6_plan(p, &pl)) {
        const int sk6 = pl.sk, bn6 = pl.bn, wc = pl.wc;
        const long long t6 = (long long)(p.M / pl.bm) * (p.N / bn6);
        p.splitk = sk6;
        p.pad = 1;
        if (bn6 != 32) p.n_valid = p.N;                              // (the 32-column tile stores only the caller's n_valid columns)
        dim3 grid((unsigned)(t6 * sk6), 1, 1);
        const bool wide = p.Wo > 128;
        if (sk6 == 1 && p.gn_part != nullptr) {
            // the generic epilogue (v6_finish) also writes the GroupNorm partial statistics of the OUTPUT, one chunk per tile of an image
            const int cpg = p.N / 32, tiles_img = (p.Ho / (pl.bm / wc)) * (p.Wo / wc);
            const bool ok = (bn6 == 256 || bn6 == 128) && p.N % 32 == 0 && (p.N == 128 || cpg % 8 == 0) && bn6 % cpg == 0 && p.n_valid == p.N && p.act == 0;
            if (ok) {
                p.gn_P = tiles_img;
                if (p.gn_part_done != nullptr) *p.gn_part_done = tiles_img;
            } else {
                p.gn_part = nullptr;
            }
        }
        if (p.gn_scale != nullptr && bn6 != V5_BN) {
            if (p.gn_shift == nullptr || pl.up || wc != 128 || p.N != bn6 || !((bn6 == 256 && pl.bm == V5_BM) || (bn6 == 128 && pl.bm == 512))) return LD_ERR_ARG;
            if (bn6 == 256) {
                t_last_kernel = "conv6_kernel<W128,halo+groupnorm,256>";
                hipLaunchKernelGGL((conv6_kernel<128, true, 256>), grid, dim3(512), 0, stream, p);
            } else {
                t_last_kernel = "conv6_kernel<W128,halo+groupnorm,128x512>";
                hipLaunchKernelGGL((conv6_kernel<128, true, 128, 512>), grid, dim3(512), 0, stream, p);
            }
        } else if (p.gn_scale != nullptr) {
            if (p.gn_shift == nullptr || pl.up) return LD_ERR_ARG;
            t_last_kernel = wc == 16 ? "conv6_kernel<W16,halo+groupnorm>" : wc == 32 ? "conv6_kernel<W32,halo+groupnorm>"
                          : wc == 64 ? "conv6_kernel<W64,halo+groupnorm>" : "conv6_kernel<W128,halo+groupnorm>";
            switch (wc) {
                case 16: hipLaunchKernelGGL((conv6_kernel<16, true>), grid, dim3(512), 0, stream, p); break;
                case 32: hipLaunchKernelGGL((conv6_kernel<32, true>), grid, dim3(512), 0, stream, p); break;
                case 64: hipLaunchKernelGGL((conv6_kernel<64, true>), grid, dim3(512), 0, stream, p); break;
                default: hipLaunchKernelGGL((conv6_kernel<128, true>), grid, dim3(512), 0, stream, p); break;
            }
        } else if (bn6 == 128) {
            t_last_kernel = "conv6_kernel<W128,halo,128x512>";
            hipLaunchKernelGGL((conv6_kernel<128, false, 128, 512>), grid, dim3(512), 0, stream, p);
        } else if (bn6 == 32) {
            t_last_kernel = "conv6_kernel<W128,halo,32x512>";
            hipLaunchKernelGGL((conv6_kernel<128, false, 32, 512>), grid, dim3(512), 0, stream, p);
        } else if (bn6 == 256) {
            if (pl.up) {
                t_last_kernel = wc == 64 ? "conv6_kernel<W64,halo,256,up>" : "conv6_kernel<W128,halo,256,up>";
                if (wc == 64) hipLaunchKernelGGL((conv6_kernel<64, false, 256, V5_BM, true>), grid, dim3(512), 0, stream, p);
                else hipLaunchKernelGGL((conv6_kernel<128, false, 256, V5_BM, true>), grid, dim3(512), 0, stream, p);
            } else {
                t_last_kernel = wc == 64 ? "conv6_kernel<W64,halo,256>" : "conv6_kernel<W128,halo,256>";
                if (wc == 64) hipLaunchKernelGGL((conv6_kernel<64, false, 256>), grid, dim3(512), 0, stream, p);
                else hipLaunchKernelGGL((conv6_kernel<128, false, 256>), grid, dim3(512), 0, stream, p);
            }
        } else if (pl.up) {
            t_last_kernel = wc == 16 ? "conv6_kernel<W16,halo,up>" : wc == 32 ? "conv6_kernel<W32,halo,up>" : wc == 64 ? "conv6_kernel<W64,halo,up>" : "conv6_kernel<W128,halo,up>";
            switch (wc) {
                case 16: hipLaunchKernelGGL((conv6_kernel<16, false, V5_BN, V5_BM, true>), grid, dim3(512), 0, stream, p); break;
                case 32: hipLaunchKernelGGL((conv6_kernel<32, false, V5_BN, V5_BM, true>), grid, dim3(512), 0, stream, p); break;
                case 64: hipLaunchKernelGGL((conv6_kernel<64, false, V5_BN, V5_BM, true>), grid, dim3(512), 0, stream, p); break;
                default: hipLaunchKernelGGL((conv6_kernel<128, false, V5_BN, V5_BM, true>), grid, dim3(512), 0, stream, p); break;
            }
        } else {
            t_last_kernel = wc == 16 ? "conv6_kernel<W16,halo>" : wc == 32 ? "conv6_kernel<W32,halo>" : wc == 64 ? "conv6_kernel<W64,halo>" : "conv6_kernel<W128,halo>";
            switch (wc) {
                case 16: hipLaunchKernelGGL((conv6_kernel<16, false>), grid, dim3(512), 0, stream, p); break;
                case 32: hipLaunchKernelGGL((conv6_kernel<32, false>), grid, dim3(512), 0, stream, p); break;
                case 64: hipLaunchKernelGGL((conv6_kernel<64, false>), grid, dim3(512), 0, stream, p); break;
                default: hipLaunchKernelGGL((conv6_kernel<128, false>), grid, dim3(512), 0, stream, p); break;
            }
        }
        (void)wide;
        if (sk6 > 1) launch_splitk_reduce(p, 160, stream);
        return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
    }
    if (p.gn_scale != nullptr) return LD_ERR_ARG;   // only the halo kernel applies a fused GroupNorm (ask gemm_conv_fuses_groupnorm first)
    // ---- v7 (row-panel kernel, A fragments in registers): the K = 320 projections whose 256-row panels fill the chip
    if (!p.conv && p.K == V7_K && p.batch == 1 && !p.ln_swapped && p.bias_m == nullptr && p.rowvec == nullptr && p.bm == 0 && (p.bn == 0 || p.bn == 160) &&
        p.splitk == 0 && (p.n_valid <= 0 || p.n_valid >= p.N) && p.N % (p.act == 2 ? 160 : V7_NB) == 0 && (p.M + V7_BM - 1) / V7_BM >= 192 &&
        (p.act == 0 || (p.act == 2 && p.bias_n != nullptr && p.stat_out == nullptr && p.R == nullptr && v7_geglu_enabled()))
#ifdef LD_AB_BUILD
        && !(g_no_v5 & 8)
#endif
    ) {
        if (p.stat_parts_out != nullptr) *p.stat_parts_out = p.N / V7_NB;
        dim3 grid((unsigned)((p.M + V7_BM - 1) / V7_BM), 1, 1);
        const bool ln = p.ln_stat != nullptr;
        if (p.act == 2) {
            t_last_kernel = "gemm7_kernel<256,K320,geglu>";
            if (ln) hipLaunchKernelGGL((gemm7_kernel<true, true>), grid, dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((gemm7_kernel<true, false>), grid, dim3(512), 0, stream, p);
        } else {
            t_last_kernel = "gemm7_kernel<256,K320,plain>";
            if (ln) hipLaunchKernelGGL((gemm7_kernel<false, true>), grid, dim3(512), 0, stream, p);
            else hipLaunchKernelGGL((gemm7_kernel<false, false>), grid, dim3(512), 0, stream, p);
        }
        return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
    }
    // ---- v5 (256 x 320 tile, 8 waves, staggered wave groups): whenever its tiles (x an optional split over K) fill the chip
    {
        // measured per launch inside the UNet forward against v3 (tools/ab_launches.py, profiles/README.md): +5..10 % on the 3x3 convolutions
        // the halo kernel cannot take and on the K = 1280 GEGLU; plain GEMMs / 1x1 convs with one tile per CU and K <= 1600 LOSE (16384 x
        // 1280 x 640: 69 vs 50 us, 65536 x 320 x 1600: 108 vs 103 us — nothing overlaps a short tile's prologue and epilogue), and so does
        // a split over K (4096 x 1280 x 11520: the 128 x 160 kernel's own split is 3 % ahead)
        const bool conv3 = p.conv && p.ksize == 3;
        const bool shape_ok = !p.ln_swapped && p.bm == 0 && (p.bn == 0 || p.bn == 160) && p.N % V5_BN == 0 && p.K % V5_BK == 0 &&
                              p.K >= (p.act == 2 ? v5_geglu_min_k() : conv3 ? 640 : 2560) && (p.n_valid == p.N || p.n_valid <= 0 || p.n_valid > p.N) &&
                              p.splitk == 0 && p.M >= 1024;
#ifdef LD_AB_BUILD
        if (shape_ok && !(g_no_v5 & 1)) {
#else
        if (shape_ok) {
#endif
            const long long t5 = (long long)((p.M + V5_BM - 1) / V5_BM) * (p.N / V5_BN) * p.batch;
            if (t5 >= 192 && !(p.act == 2 && p.stat_out != nullptr)) {
                if (p.stat_parts_out != nullptr) *p.stat_parts_out = 2 * (p.N / V5_BN);
                p.splitk = 1;
                p.bn = 160;
                if (p.n_valid <= 0 || p.n_valid > p.N) p.n_valid = p.N;
                dim3 grid((unsigned)t5, 1, 1);
                grid.x = (unsigned)(((p.M + V5_BM - 1) / V5_BM) * (p.N / V5_BN));
                grid.z = (unsigned)p.batch;
                const bool ln = p.ln_stat != nullptr || p.stat_out != nullptr;
                if (p.conv) {
                    if (ln || p.act == 2) return LD_ERR_ARG;          // (no caller: convolutions carry neither the LayerNorm fold nor GEGLU)
                    t_last_kernel = "gemm5_kernel<256,320,conv>";
                    hipLaunchKernelGGL((gemm5_kernel<true, 0>), grid, dim3(512), 0, stream, p);
                } else {
                    t_last_kernel = "gemm5_kernel<256,320,plain>";
                    {   // XCD-blocked tile order (see the kernel): the split of the 8 XCDs over (M, N) that moves the fewest bytes, A once per
                        // N group and W once per M group
                        const int tm5 = (p.M + V5_BM - 1) / V5_BM, tn5 = p.N / V5_BN;
                        p.xcd_gm = 0;
                        if (p.batch == 1 && (tm5 * tn5) % 8 == 0 && p.M % V5_BM == 0) {
                            double best = 0;
                            for (int gm = 1; gm <= 8; gm <<= 1) {
                                const int gn = 8 / gm;
                                if (tm5 % gm || tn5 % gn) continue;
                                const double cost = (double)p.M * gn + (double)p.N * gm;     // x K x 2 bytes
                                if (p.xcd_gm == 0 || cost < best) {
                                    best = cost;
                                    p.xcd_gm = gm;
                                }
                            }
                        }
#ifdef LD_AB_BUILD
                        if (g_no_v5 & 512) p.xcd_gm = 0;
#endif
                    }
                    if (p.act == 2) hipLaunchKernelGGL((gemm5_kernel<false, 2>), grid, dim3(512), 0, stream, p);
                    else if (ln) hipLaunchKernelGGL((gemm5_kernel<false, 1>), grid, dim3(512), 0, stream, p);
                    else hipLaunchKernelGGL((gemm5_kernel<false, 0>), grid, dim3(512), 0, stream, p);
                }
                return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
            }
        }
    }
    int bn = p.bn ? p.bn : gemm_pick_bn(p.N);
    // Skinny plain GEMMs (the batch-1 step's M = 128..2048 projections): with <= 128 tiles of 64 x 160 most CUs idle while
    // each busy one streams 28.7 KB per slab through its one LDS-DMA path; 64 x 64 tiles spread the same work over 2.5x more CUs
    // at 16 KB per slab: 512x1280x1280 12.8 -> 8.0 us, 128x1280x1280 12.4 -> 7.6 us; batch-1 step +6 % (same box), batch 8 neutral.
    // Measured no better: the same for convs (their split-K already fills the chip: 156.5 vs 163.0 steps/s), 32-row tiles
    // below it (166.2 vs 165.8); thresholds 128 / 256 / 512: 162.0 / 163.6 / 164.2 at B=1, 49.25 / 49.38 / 48.48 at B=8.
    constexpr int skinny_max = 256;
    const bool skinny_ok = !p.conv && p.act != 2 && p.bn == 0 && p.bm == 0 && (p.N % 64) == 0;
    if (skinny_ok) {
        const long long t160 = (long long)((p.M + 63) / 64) * ((p.N + bn - 1) / bn) * p.batch;
        if (t160 <= skinny_max && (p.N + bn - 1) / bn <= 8) bn = 64;      // (wide outputs keep the 160-column tile: 512 x 2560 x 1280 -9 % in the forward)
    }
    // Round 5: the batch-1 step's 1x1 contractions over two sources (ResBlock skip_connection on the concatenated input, the MLP-out fold at
    // K = 3200) take the same 64 x 64 tiles UNSPLIT on the producer / consumer kernel with two workgroups per CU, instead of 64 x 160 tiles
    // split over K + a reduce launch.  Per shape inside the batch-1 forward (tools/ab_launches.py 1 0 8192, profiles/r05_ab_skinny_conv1.txt):
    // 2048 x 640 x 3200 32.2 -> 27.7 us, 2048 x 640 x 1920 26.1 -> 19.4, 512 x 1280 x 2560 25.3 -> 21.4; K = 6400 (100 slabs in one
    // workgroup) loses: 31.8 -> 37.1 us, and as two slices of 3200 + reduce 31.5 -> 34.4 us, so K stops at 3200.  (Without a split there are no GroupNorm partials from the reduce pass: the next
    // GroupNorm runs its own statistics pass — counted in the whole-forward A/B: 5.225 -> 5.207 ms before K was capped.)
    bool skinny_conv1 = false;
    if (skinny_conv1_ok() && p.conv && p.ksize == 1 && p.act == 0 && p.bn == 0 && p.bm == 0 && p.splitk == 0 && p.batch == 1 && (p.N % 64) == 0 && p.SC1 == 0) {
        const long long t64 = (long long)((p.M + 63) / 64) * (p.N / 64);
        if (t64 > 128 && t64 <= 512 && p.K >= 1280 && p.K <= 3200) {
            bn = 64;
            skinny_conv1 = true;
        }

    }
    if (bn != 128 && bn != 160 && bn != 64) return LD_ERR_ARG;
    if (p.act == 2 && (p.N % bn)) return LD_ERR_SHAPE;
    const int tiles_n = (p.N + bn - 1) / bn;
    const int KT = (p.K + BK - 1) / BK;          // 64-wide K slabs
    // Tile height and split-K, fitted to a per-shape sweep of every contraction of the SD1.5 UNet at UNet batch 2 and 16
    // (tools/gemm_sweep.py, profiles/README.md): bigger tiles win whenever they fill the chip; a split only pays when each
    // slice keeps >= ~640 of K (its second pass is a ~6 us launch plus fp32 slab traffic); short-K problems prefer
    // 64-row tiles and no split; long-K problems prefer 128-row tiles and a split up to ~2 blocks per CU.
    const int tiles128 = ((p.M + 127) / 128) * tiles_n * p.batch;
    const bool can_split = p.batch == 1 && p.partial != nullptr;
    const int sk_cap = p.K / 640 < 1 ? 1 : (p.K / 640 > 16 ? 16 : p.K / 640);   // (640: in-forward sweep, tools/ab_shape.py — 128 x 1280 x 2560: split 4 -14 %)
    int bm = p.bm, sk = p.splitk;
    if (bm == 0) {
        // (second and third line re-fitted launch by launch INSIDE the forward, tools/ab_shape.py: an isolated sweep keeps a shape's
        // weights in L2 / MALL and mis-ranks the candidates.  512 x 1280 x 11520: 64-row tiles + split 8 -11 % against 128 / 15;
        // 4096 x 1280 x 1280: 128-row tiles -6 %)
        if (tiles128 >= 512) bm = 128;
        else if (p.K >= 4096 && can_split && (tiles128 >= 128 || (p.K >= 8192 && tiles128 >= 64))) bm = 128;
        else if (!p.conv && tiles128 >= 256) bm = 128;
        else bm = 64;
    }
    if (bn == 64) bm = 64;
    if (skinny_conv1) sk = 1;
    if (bm != 64 && bm != 128) return LD_ERR_ARG;
    const int tiles = ((p.M + bm - 1) / bm) * tiles_n;
    // Convolutions on 64 x 160 tiles with very few tiles (<= 32: the 8 x 8 level of a batch-1 step) or exactly one round of them (256 .. 511)
    // run the 4-stage ring with ONE workgroup per CU and a split aimed at 256 workgroups: three slabs in flight per workgroup hide the
    // HBM latency of their cold weights better than two co-resident 2-stage workgroups (per launch inside the batch-1 forward,
    // tools/ab_launches.py: 128 x 1280 x 11520 24.3 -> 22.2 us, 8192 x 320 x 1600 32 -> 26 us; 64 .. 128 tiles with long K lose 10 %).
    bool deep = p.conv && bm == 64 && bn == 160 && p.batch == 1 && p.splitk == 0 && (tiles <= 32 || (tiles >= 256 && tiles < 512));
#ifdef LD_AB_BUILD
    if (g_no_v5 & 128) deep = false;
#endif
    if (sk == 0) {
        sk = 1;
        if (can_split && tiles * p.batch < 512) {
            sk = ((deep ? 256 : 512) + tiles - 1) / tiles;
            if (sk > sk_cap) sk = sk_cap;
        }
    }
    if (sk > 1) {
        if (p.batch != 1 || p.partial == nullptr) return LD_ERR_ARG;
        while (sk > 1 && (size_t)sk * p.M * p.N * sizeof(float) > p.partial_bytes) --sk;
        if (sk > KT) sk = KT;
    }
    if (p.stat_out != nullptr || p.ln_stat != nullptr) {   // LN fold: v3 / v4 kernels, whole K in one workgroup
        if (p.act == 2 && p.stat_out != nullptr) return LD_ERR_ARG;
        sk = 1;
        if (p.stat_parts_out != nullptr) *p.stat_parts_out = tiles_n;
    }
    p.splitk = sk;
    p.bn = bn;
    // measured (profiles/r01_b): n-fastest wins on every SD1.5 shape — the 9 taps of a 3x3 conv and the N tiles of one
    // M panel re-read the same activations through the XCD's L2, which matters more than re-streaming the weights
    // Round 5: a plain GEMM with few M panels and a large weight matrix (the batch-1 step's M = 512 GEGLU: 4 panels x 26 MB) walks its tiles
    // M-fastest, so the panels of one N tile run together on one XCD and the weight tile leaves HBM once instead of once per panel
    // (profiles/pmc_traffic.json round 4: 113 MB per launch for 26 MB of weights)
    if (p.m_fastest < 0) {
        const int tiles_m = (p.M + bm - 1) / bm;
        p.m_fastest = (!p.conv && p.batch == 1 && tiles_m >= 2 && tiles_m <= 8 && (long long)p.N * p.K * 2 >= (8ll << 20) && m_fastest_ok()) ? 1 : 0;
    }

    // every unsplit 1x1 convolution that ends on 64 x 64 tiles with at most two workgroups per CU takes the 2wg kernel: the skinny_conv1 shapes
    // and the few-tile shapes of the older skinny_max rule alike; A/B bit 8192 (skinny_conv1_ok) restores round 4's route for all of them
    const bool conv1_2wg = bn == 64 && skinny_conv1_ok() && p.conv && p.ksize == 1 && sk <= 1 && (long long)tiles * p.batch <= 512;
    if (bn == 64) launch_cfg<64, 64>(p, stream, false, conv1_2wg);
    else if (bm == 128 && bn == 160) launch_cfg<128, 160>(p, stream);
    else if (bm == 128 && bn == 128) launch_cfg<128, 128>(p, stream);
    else if (bm == 64 && bn == 160) launch_cfg<64, 160>(p, stream, deep);
    else launch_cfg<64, 128>(p, stream);

    if (sk > 1) launch_splitk_reduce(p, bn, stream);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}
