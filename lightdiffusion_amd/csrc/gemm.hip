// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950.
//
// Tile: BM x BN x 64, 4 waves (2x2), each wave (BM/2)x(BN/2) as 16x16x32 f16 MFMA tiles.
// Staging: global -> registers (issued before the MFMA phase of the current tile, T14 split) -> LDS, two LDS
// stages, ONE barrier per K-step.  LDS tiles are [rows][64 halfs] with the 16-byte chunk index XOR-swizzled
// by (row & 7): ds_read_b128 fragment reads are bank-conflict free (cdna guide T2).
// MFMA operands are swapped (a := W fragment, b := A fragment) so that each lane ends up with 4 consecutive
// output channels of one output row -> 8-byte LDS writes / 16-byte split-K stores in the epilogue.
// Epilogue: accumulators -> fp16 tile in LDS -> row-wise 16-byte coalesced stores with the fused bias /
// time-embedding broadcast / SiLU / GEGLU / residual.
#include "gemm.h"

namespace {

constexpr int BK = 64;
constexpr int NT = 256;

__device__ __forceinline__ void epilogue_store8(const GemmParams& p, int z, int m, int n_out, int n_bias, float (&v)[8]) {
    // v already holds alpha*acc (and, for GEGLU, the gated product with biases applied)
    if (p.bias_n != nullptr && p.act != 2) {
        float b[8];
        unpack8(ld16(p.bias_n + n_bias), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += b[j];
    }
    if (p.bias_m != nullptr) {
        const float bm = (float)p.bias_m[m];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bm;
    }
    if (p.rowvec != nullptr) {
        float b[8];
        unpack8(ld16(p.rowvec + (long long)(m / p.rows_per_vec) * p.ldrv + n_out), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += b[j];
    }
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
    }
    if (p.R != nullptr) {
        float r[8];
        unpack8(ld16(p.R + (long long)z * p.sR + (long long)m * p.ldr + n_out), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    st16(p.C + (long long)z * p.sC + (long long)m * p.ldc + n_out, pack8(v));
}

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const GemmParams p) {
    constexpr int WTM = BM / 2, WTN = BN / 2;   // wave tile
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = (BM * 8) / NT;         // 16-byte chunks per thread per A tile
    constexpr int B_IT = (BN * 8 + NT - 1) / NT;
    constexpr int STAGE = (BM + BN) * BK;       // halfs per stage
    constexpr int CLD = BN + 8;                 // epilogue tile row stride (halfs)
    static_assert(BM * CLD <= 2 * STAGE, "epilogue tile must fit in the staging LDS");
    __shared__ __attribute__((aligned(16))) half_t smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * WTM, wn0 = (wid & 1) * WTN;
    const int z = blockIdx.z;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    const int splitk = p.splitk > 1 ? p.splitk : 1;
    int bid = xcd_remap(blockIdx.x, tiles * splitk);
    const int ks = bid / tiles;
    bid -= ks * tiles;
    // n fastest: consecutive blocks (one XCD) sweep the N tiles of one M panel -> A panel stays in that L2
    const int tn_i = bid % tiles_n, tm_i = bid / tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;

    const int KT = (p.K + BK - 1) / BK;
    const int kt_begin = (int)((long long)ks * KT / splitk), kt_end = (int)((long long)(ks + 1) * KT / splitk);

    const half_t* Ab = p.A + (long long)z * p.sA;
    const half_t* A2b = p.A2;
    const half_t* Wb = p.W + (long long)z * p.sW;
    const int Cin = p.C1 + p.C2;

    // ---- per-thread staging coordinates (fixed over the K loop)
    int a_row[A_IT];
    bool a_ok[A_IT];
    long long a_base[A_IT];   // plain: element offset of the row;  conv: image index
    int a_iy0[A_IT], a_ix0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int q = tid + i * NT;
        const int row = q >> 3;
        a_row[i] = row;
        const int m = m0 + row;
        a_ok[i] = m < p.M;
        if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / hw, rem = mm - img * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_base[i] = img;
            a_iy0[i] = oy * p.stride - (p.ksize >> 1);
            a_ix0[i] = ox * p.stride - (p.ksize >> 1);
        } else {
            a_base[i] = (long long)m * p.lda;
            a_iy0[i] = a_ix0[i] = 0;
        }
    }
    const int ch = tid & 7;   // chunk within the 64-wide K slab (same for every i: NT % 8 == 0)

    uint4 ra[A_IT], rb[B_IT];

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        if (CONV) {
            const int tap = k0 / Cin;
            const int c0 = k0 - tap * Cin;
            const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
            const bool second = c0 >= p.C1;
            const half_t* src = second ? A2b : Ab;
            const int Cs = second ? p.C2 : p.C1;
            const int cl = (second ? c0 - p.C1 : c0) + ch * 8;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
                int sy = iy, sx = ix;
                if (p.Hv == 2 * p.Hs && p.Wv == 2 * p.Ws) {   // exact 2x nearest upsample
                    sy = iy >> 1;
                    sx = ix >> 1;
                } else if (p.Hv != p.Hs || p.Wv != p.Ws) {    // general nearest resize: src = floor(dst * in / out)
                    sy = (int)((long long)iy * p.Hs / p.Hv);
                    sx = (int)((long long)ix * p.Ws / p.Wv);
                }
                const long long off = ((a_base[i] * p.Hs + sy) * p.Ws + sx) * Cs + cl;
                ra[i] = ok ? ld16(src + off) : zero16();
            }
        } else {
            const int kc = k0 + ch * 8;
            const bool kok = kc < p.K;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = (a_ok[i] && kok) ? ld16(Ab + a_base[i] + kc) : zero16();
        }
        const int kc = k0 + ch * 8;
        const bool kok = kc < p.K;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int q = tid + i * NT;
            const int row = q >> 3;
            const bool ok = (q < BN * 8) && (n0 + row < p.n_valid) && kok;
            rb[i] = ok ? ld16(Wb + (long long)(n0 + row) * p.ldw + kc) : zero16();
        }
    };

    auto store_tiles = [&](int stage) {
        half_t* As = smem + stage * STAGE;
        half_t* Bs = As + BM * BK;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) st16(As + a_row[i] * BK + ((ch ^ (a_row[i] & 7)) << 3), ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int q = tid + i * NT;
            const int row = q >> 3;
            if (q < BN * 8) st16(Bs + row * BK + ((ch ^ (row & 7)) << 3), rb[i]);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;

    if (kt_begin < kt_end) {
        load_tiles(kt_begin);
        store_tiles(0);
    }
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int stage = (kt - kt_begin) & 1;
        const bool more = kt + 1 < kt_end;
        if (more) load_tiles(kt + 1);   // global loads in flight under the MFMA phase below
        const half_t* As = smem + stage * STAGE;
        const half_t* Bs = As + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 bf[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 16 + fr;
                bf[j] = as_half8(ld16(Bs + row * BK + (((kk * 4 + fq) ^ (row & 7)) << 3)));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 16 + fr;
                const half8 af = as_half8(ld16(As + row * BK + (((kk * 4 + fq) ^ (row & 7)) << 3)));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af, acc[i][j], 0, 0, 0);
            }
        }
        if (more) store_tiles(stage ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  D = Wfrag x Afrag^T: lane holds rows n = fq*4 + r (r = 0..3) of column m = fr.
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
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm0 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn0 + j * 16 + fq * 4;
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[i][j][r] * p.alpha);
            *reinterpret_cast<half4*>(Cs + ml * CLD + nl) = h;
        }
    }
    __syncthreads();

    if (p.act == 2) {
        constexpr int CPR = BN / 16;   // output chunks per row (BN/2 columns)
        for (int q = tid; q < BM * CPR; q += NT) {
            const int row = q / CPR, cc = q - row * CPR;
            const int m = m0 + row;
            const int nv = n0 + cc * 8, ng = nv + BN / 2;
            if (m < p.M && ng < p.N) {
                float a[8], g[8], ba[8], bg[8];
                unpack8(ld16(Cs + row * CLD + cc * 8), a);
                unpack8(ld16(Cs + row * CLD + BN / 2 + cc * 8), g);
                unpack8(ld16(p.bias_n + nv), ba);
                unpack8(ld16(p.bias_n + ng), bg);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (a[j] + ba[j]) * gelu_f(g[j] + bg[j]);
                epilogue_store8(p, z, m, n0 / 2 + cc * 8, 0, a);
            }
        }
    } else {
        constexpr int CPR = BN / 8;
        for (int q = tid; q < BM * CPR; q += NT) {
            const int row = q / CPR, cc = q - row * CPR;
            const int m = m0 + row, n = n0 + cc * 8;
            if (m < p.M && n < p.N) {
                float v[8];
                unpack8(ld16(Cs + row * CLD + cc * 8), v);
                epilogue_store8(p, z, m, n, n, v);
            }
        }
    }
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
            for (int s = 0; s < p.splitk; ++s) {
                const float* base = p.partial + ((long long)s * p.M + m) * p.N + n;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(base), x1 = *reinterpret_cast<const f32x4*>(base + 4);
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

template <int BM, int BN>
void launch_cfg(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int sk = p.splitk > 1 ? p.splitk : 1;
    dim3 grid(tiles * sk, 1, p.batch), block(NT);
    if (p.conv)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, true>), grid, block, 0, s, p);
    else
        hipLaunchKernelGGL((gemm_kernel<BM, BN, false>), grid, block, 0, s, p);
}

}  // namespace

int gemm_launch(const GemmParams& pin, hipStream_t stream) {
    GemmParams p = pin;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.A == nullptr || p.W == nullptr || p.C == nullptr) return LD_ERR_ARG;
    if ((p.N & 7) || (p.K & 7) || (p.ldw & 7) || (p.ldc & 7)) return LD_ERR_SHAPE;
    if (p.conv) {
        const int Cin = p.C1 + p.C2;
        if (p.ksize != 1 && p.ksize != 3) return LD_ERR_ARG;
        if (Cin <= 0 || (p.C1 % 64) || (p.C2 % 64) || p.K != p.ksize * p.ksize * Cin) return LD_ERR_SHAPE;
        if (p.C2 > 0 && p.A2 == nullptr) return LD_ERR_ARG;
        if (p.M % (p.Ho * p.Wo)) return LD_ERR_SHAPE;
        if (p.batch != 1) return LD_ERR_ARG;
    } else if (p.lda & 7) {
        return LD_ERR_SHAPE;
    }
    if (p.act == 2 && (p.bias_n == nullptr || (p.N & 15))) return LD_ERR_ARG;
    if (p.R != nullptr && (p.ldr & 7)) return LD_ERR_SHAPE;

    if (p.n_valid <= 0 || p.n_valid > p.N) p.n_valid = p.N;
    int bn = p.bn ? p.bn : gemm_pick_bn(p.N);
    if (bn != 128 && bn != 160) return LD_ERR_ARG;
    if (p.act == 2 && (p.N % bn)) return LD_ERR_SHAPE;
    const int tiles_n = (p.N + bn - 1) / bn;
    int bm = p.bm;
    if (bm == 0) bm = (((p.M + 127) / 128) * tiles_n * p.batch >= 256) ? 128 : 64;
    if (bm != 64 && bm != 128) return LD_ERR_ARG;
    const int tiles = ((p.M + bm - 1) / bm) * tiles_n;
    const int KT = (p.K + BK - 1) / BK;

    int sk = p.splitk;
    if (sk == 0) {   // auto: fill ~2 blocks per CU when the tile grid alone cannot
        sk = 1;
        if (p.batch == 1 && p.partial != nullptr && tiles < 384 && KT >= 8) {
            sk = (512 + tiles - 1) / tiles;
            if (sk > KT / 4) sk = KT / 4;
            if (sk > 32) sk = 32;
            if (sk < 1) sk = 1;
        }
    }
    if (sk > 1) {
        if (p.batch != 1 || p.partial == nullptr) return LD_ERR_ARG;
        while (sk > 1 && (size_t)sk * p.M * p.N * sizeof(float) > p.partial_bytes) --sk;
        if (sk > KT) sk = KT;
    }
    p.splitk = sk;
    p.bn = bn;

    if (bm == 128 && bn == 160) launch_cfg<128, 160>(p, stream);
    else if (bm == 128 && bn == 128) launch_cfg<128, 128>(p, stream);
    else if (bm == 64 && bn == 160) launch_cfg<64, 160>(p, stream);
    else launch_cfg<64, 128>(p, stream);

    if (sk > 1) {
        const int out_n = p.act == 2 ? p.N / 2 : p.N;
        const long long total = (long long)p.M * (out_n / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p, bn);
    }
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}
