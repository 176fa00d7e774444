// GroupNorm(32) (+SiLU) and LayerNorm for NHWC / token-major fp16 activations.  HBM-bound kernels:
// 16-byte loads/stores, fp32 statistics, wave-level reductions.
//
// GroupNorm takes up to two NHWC sources that are *virtually concatenated* along channels (the UNet's
// th.cat([h, hs.pop()], 1), LD.py:5749) and writes one contiguous normalised tensor, so the concat is never
// materialised on its own.  Two launches: (1) partial sums per (image, pixel-chunk, group); (2) finish the
// statistics (tiny, L2-resident) and apply scale/shift (+SiLU) as one FMA per element.
#include "kernels.h"

namespace {

constexpr int GN_THREADS = 256;
// channel slabs (grid.z) of 32 / slabs groups each: 4 from C = 256 on (a block's pixel row segment is then >= 128 bytes); fewer for
// narrow tensors so that a segment stays a whole 128-byte line — with four 64-byte slabs the VAE's C = 128 GroupNorms ran at 3.1 TB/s
// against 5.3 TB/s for C = 256 / 512 (profiles/r03a_vae512_launches.txt).  The UNet's tensors (C >= 320) keep 4: summation order unchanged.
__host__ __device__ inline int gn_slabs(int C) { return C >= 256 ? 4 : C >= 128 ? 2 : 1; }

struct GnArgs {
    const half_t* x1;
    const half_t* x2;
    int C1, C2, HW, P, ppb;   // P pixel-chunks per image, ppb pixels per chunk (the grid of the statistics and of the apply pass)
    int Pstat;                // chunks per image of `partial` (= P when gn_stats_kernel wrote it; a producer's epilogue may use another count)
    int slabs;                // gn_slabs(C1 + C2)
    float* partial;           // [N][Pstat][32][2]
    const half_t* gamma;
    const half_t* beta;
    half_t* y;
    float eps;
    int silu;
};

__device__ __forceinline__ const half_t* gn_src(const GnArgs& a, int n, int pix, int c) {
    return c < a.C1 ? a.x1 + ((long long)n * a.HW + pix) * a.C1 + c
                    : a.x2 + ((long long)n * a.HW + pix) * a.C2 + (c - a.C1);
}

// Block = (pixel chunk, image, slab of 8 groups = C/4 channels = C/32 16-byte chunks).  A thread owns ONE 8-channel
// chunk (fixed over its pixel loop, so scale/shift live in registers) and strides over the pixels of the chunk.
// No float atomics: every LDS slot has one writer and the reductions run in a fixed order (bitwise reproducible).
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const GnArgs a) {
    __shared__ float csum[2048], csq[2048];   // [rows_par][slab channels], rows_par * CS <= 256 * 8
    const int C = a.C1 + a.C2, cpg = C / 32;
    const int CS = C / a.slabs, CHS = CS >> 3;                  // slab channels / chunks
    const int gps = 32 / a.slabs, lpg = GN_THREADS / gps;       // groups per slab, lanes per group in the final reduction (32 / 16 / 8)
    const int n = blockIdx.y, pc = blockIdx.x, slab = blockIdx.z, tid = threadIdx.x;
    const int rows_par = GN_THREADS / CHS;
    const int cc = tid % CHS, prow = tid / CHS;
    const int c0 = slab * CS + cc * 8;
    const int p_begin = pc * a.ppb, p_end = min(a.HW, p_begin + a.ppb);
    if (prow < rows_par) {
        float s[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
        // four pixels per batch of loads (one load, wait, accumulate per iteration left a wave a single request in flight), summed in
        // pixel order as before
        const half_t* src = gn_src(a, n, p_begin + prow, c0);
        const long long pstep = (long long)rows_par * (c0 < a.C1 ? a.C1 : a.C2);
        int pix = p_begin + prow;
        for (; pix + 3 * rows_par < p_end; pix += 4 * rows_par, src += 4 * pstep) {
            uint4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = ld16(src + u * pstep);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
                unpack8(r[u], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s[j] += v[j];
                    ss[j] += v[j] * v[j];
                }
            }
        }
        for (; pix < p_end; pix += rows_par, src += pstep) {
            float v[8];
            unpack8(ld16(src), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += v[j];
                ss[j] += v[j] * v[j];
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
        const int g = tid / lpg, sub = tid - g * lpg;            // gps groups x lpg lanes (4 slabs: 8 x 32)
        const int cnt = rows_par * cpg;
        float s = 0.f, ss = 0.f;
        for (int i = sub; i < cnt; i += lpg) {
            const int pr = i / cpg, c = g * cpg + (i - pr * cpg);
            s += csum[pr * CS + c];
            ss += csq[pr * CS + c];
        }
        for (int o = 1; o < lpg; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            ss += __shfl_xor(ss, o, 64);
        }
        if (sub == 0) {
            float* o = a.partial + (((long long)n * a.P + pc) * 32 + slab * gps + g) * 2;
            o[0] = s;
            o[1] = ss;
        }
    }
}

__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const GnArgs a) {
    __shared__ float mean[32], rstd[32];
    const int C = a.C1 + a.C2, cpg = C / 32;
    const int CS = C / a.slabs, CHS = CS >> 3;
    const int gps = 32 / a.slabs, lpg = GN_THREADS / gps;
    const int n = blockIdx.y, pc = blockIdx.x, slab = blockIdx.z, tid = threadIdx.x;
    {   // finish the statistics of this slab's groups: lpg lanes per group sweep the P partial slabs in a fixed order
        const int g = tid / lpg, sub = tid - g * lpg;
        float s = 0.f, ss = 0.f;
        const float* pp = a.partial + ((long long)n * a.Pstat * 32 + slab * gps + g) * 2;
        for (int i = sub; i < a.Pstat; i += 4 * lpg) {   // four chunks per batch of loads (a producer's epilogue may have written up to 1024 chunks per image)
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int iu = i + u * lpg;
                v[u] = *reinterpret_cast<const float2*>(pp + (long long)(iu < a.Pstat ? iu : sub) * 64);
                if (iu >= a.Pstat) v[u] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += v[u].x;
                ss += v[u].y;
            }
        }
        for (int o = 1; o < lpg; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            ss += __shfl_xor(ss, o, 64);
        }
        if (sub == 0) {
            const float cnt = (float)cpg * (float)a.HW;
            const float mu = s / cnt;
            const float var = fmaxf(ss / cnt - mu * mu, 0.f);
            mean[g] = mu;
            rstd[g] = rsqrtf(var + a.eps);
        }
    }
    __syncthreads();
    const int rows_par = GN_THREADS / CHS;
    const int cc = tid % CHS, prow = tid / CHS;
    if (prow >= rows_par) return;
    const int c0 = slab * CS + cc * 8;
    const int p_begin = pc * a.ppb, p_end = min(a.HW, p_begin + a.ppb);
    float ga[8], be[8], sc[8], sh[8];
    unpack8(ld16(a.gamma + c0), ga);
    unpack8(ld16(a.beta + c0), be);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (cc * 8 + j) / cpg;                        // group within the slab
        sc[j] = rstd[g] * ga[j];
        sh[j] = be[j] - mean[g] * sc[j];
    }
    const half_t* src = gn_src(a, n, p_begin + prow, c0);
    const long long pstep = (long long)rows_par * (c0 < a.C1 ? a.C1 : a.C2);
    half_t* dst = a.y + ((long long)n * a.HW + p_begin + prow) * C + c0;
    const long long dstep = (long long)rows_par * C;
    auto norm8 = [&](uint4 raw) {
        float v[8];
        unpack8(raw, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = v[j] * sc[j] + sh[j];
            if (a.silu) v[j] = silu_f(v[j]);
        }
        return pack8(v);
    };
    int pix = p_begin + prow;
    for (; pix + 3 * rows_par < p_end; pix += 4 * rows_par, src += 4 * pstep, dst += 4 * dstep) {   // four pixels per batch of loads
        uint4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = ld16(src + u * pstep);
#pragma unroll
        for (int u = 0; u < 4; ++u) st16(dst + u * dstep, norm8(r[u]));
    }
    for (; pix < p_end; pix += rows_par, src += pstep, dst += dstep) st16(dst, norm8(ld16(src)));
}

// GroupNorm statistics only, finished into per-(image, channel) scale / shift for a consumer that applies them itself (the halo
// convolution, gemm.h gn_scale / gn_shift): same reduction order and the same  sc = rstd * gamma,  sh = beta - mean * sc  as
// gn_apply_kernel, so the fused path reproduces the two-pass one bit for bit.  One block per (image, slab of 8 groups).
__global__ __launch_bounds__(GN_THREADS) void gn_finalize_kernel(const GnArgs a, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float mean[32], rstd[32];
    const int C = a.C1 + a.C2, cpg = C / 32, CS = C / a.slabs;
    const int gps = 32 / a.slabs, lpg = GN_THREADS / gps;
    const int n = blockIdx.x, slab = blockIdx.y, tid = threadIdx.x;
    {
        const int g = tid / lpg, sub = tid - g * lpg;
        float s = 0.f, ss = 0.f;
        const float* pp = a.partial + ((long long)n * a.Pstat * 32 + slab * gps + g) * 2;
        for (int i = sub; i < a.Pstat; i += 4 * lpg) {   // four chunks per batch of loads (a producer's epilogue may have written up to 1024 chunks per image)
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int iu = i + u * lpg;
                v[u] = *reinterpret_cast<const float2*>(pp + (long long)(iu < a.Pstat ? iu : sub) * 64);
                if (iu >= a.Pstat) v[u] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += v[u].x;
                ss += v[u].y;
            }
        }
        for (int o = 1; o < lpg; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            ss += __shfl_xor(ss, o, 64);
        }
        if (sub == 0) {
            const float cnt = (float)cpg * (float)a.HW;
            const float mu = s / cnt;
            const float var = fmaxf(ss / cnt - mu * mu, 0.f);
            mean[g] = mu;
            rstd[g] = rsqrtf(var + a.eps);
        }
    }
    __syncthreads();
    for (int c = tid; c < CS; c += GN_THREADS) {
        const int g = c / cpg, cg = slab * CS + c;
        const float sc = rstd[g] * (float)a.gamma[cg];
        scale[(long long)n * C + cg] = sc;
        shift[(long long)n * C + cg] = (float)a.beta[cg] - mean[g] * sc;
    }
}

// LayerNorm over the last dim: one wave per row, row held in registers (C <= 64 lanes * 4 chunks * 8 = 2048).
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const half_t* __restrict__ gamma,
                                                         const half_t* __restrict__ beta, half_t* __restrict__ y,
                                                         int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int CH = C >> 3;
    float v[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = lane + i * 64;
        if (cc < CH) {
            unpack8(ld16(x + (long long)row * C + cc * 8), v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        }
    }
    const float mu = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = lane + i * 64;
        if (cc < CH) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mu;
                ss += d * d;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = lane + i * 64;
        if (cc < CH) {
            float ga[8], be[8];
            unpack8(ld16(gamma + cc * 8), ga);
            unpack8(ld16(beta + cc * 8), be);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = (v[i][j] - mu) * rs * ga[j] + be[j];
            st16(y + (long long)row * C + cc * 8, pack8(v[i]));
        }
    }
}

// row softmax in place over fp16 scores (VAE mid-block attention: one head, L = H*W keys)
// `valid` <= cols: columns valid..cols-1 are padding (scores ignored, probabilities written as zeros).
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* __restrict__ s, int cols, int valid, long long ld) {
    __shared__ float red[4];
    half_t* row = s + (long long)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int CH = cols >> 3;
    float mx = -INFINITY;
    for (int c = tid; c < CH; c += 256) {
        float v[8];
        unpack8(ld16(row + c * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, c * 8 + j < valid ? v[j] : -INFINITY);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = tid; c < CH; c += 256) {
        float v[8];
        unpack8(ld16(row + c * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += c * 8 + j < valid ? __expf(v[j] - mx) : 0.f;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = tid; c < CH; c += 256) {
        float v[8];
        unpack8(ld16(row + c * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = c * 8 + j < valid ? __expf(v[j] - mx) * inv : 0.f;
        st16(row + c * 8, pack8(v));
    }
}

}  // namespace

size_t groupnorm_workspace_bytes(int n_img, int HW) {
    int P = gn_num_chunks(n_img, HW);
    const int p8 = HW / 16 < 256 ? HW / 16 : 256;               // a producer's epilogue may write 16-pixel chunks (conv8: HW / 16 per image, up to 64 x 64 images)
    if (p8 > P) P = p8;
    if (HW / 256 > P) P = HW / 256;                             // ... or one chunk per 256-pixel tile (the halo convolution's generic epilogue)
    return (size_t)n_img * P * 32 * 2 * sizeof(float);
}

int groupnorm_stats_launch(const half_t* x1, int C1, const half_t* x2, int C2, int n_img, int HW, float* partial, hipStream_t stream) {
    const int C = C1 + C2;
    if (x1 == nullptr || partial == nullptr) return LD_ERR_ARG;
    if (C % 32 || C1 % 8 || C2 % 8 || C > 8192 || C1 <= 0 || (C2 > 0 && x2 == nullptr)) return LD_ERR_SHAPE;
    GnArgs a;
    a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.HW = HW;
    a.P = a.Pstat = gn_num_chunks(n_img, HW);
    a.ppb = (HW + a.P - 1) / a.P;
    a.partial = partial; a.gamma = nullptr; a.beta = nullptr; a.y = nullptr; a.eps = 0.f; a.silu = 0;
    a.slabs = gn_slabs(C);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a.P, n_img, a.slabs), dim3(GN_THREADS), 0, stream, a);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int groupnorm_launch(const half_t* x1, int C1, const half_t* x2, int C2, int n_img, int HW, const half_t* gamma,
                     const half_t* beta, float eps, int silu, half_t* y, float* partial, hipStream_t stream, int stats_ready) {
    const int C = C1 + C2;
    if (x1 == nullptr || y == nullptr || partial == nullptr || gamma == nullptr || beta == nullptr) return LD_ERR_ARG;
    if (C % 32 || C1 % 8 || C2 % 8 || C > 8192 || C1 <= 0 || (C2 > 0 && x2 == nullptr)) return LD_ERR_SHAPE;
    GnArgs a;
    a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.HW = HW;
    a.P = gn_num_chunks(n_img, HW);
    a.Pstat = stats_ready > 0 ? stats_ready : a.P;            // stats_ready: chunk count of the partials a producer wrote
    a.ppb = (HW + a.P - 1) / a.P;
    a.partial = partial; a.gamma = gamma; a.beta = beta; a.y = y; a.eps = eps; a.silu = silu;
    a.slabs = gn_slabs(C);
    dim3 grid(a.P, n_img, a.slabs);
    if (!stats_ready) hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(GN_THREADS), 0, stream, a);   // (ready: the producer's split-K reduce wrote `partial`, gemm.h gn_part)
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(GN_THREADS), 0, stream, a);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int groupnorm_scale_shift_launch(const half_t* x1, int C1, const half_t* x2, int C2, int n_img, int HW, const half_t* gamma, const half_t* beta,
                                 float eps, float* partial, float* scale, float* shift, hipStream_t stream, int stats_ready) {
    const int C = C1 + C2;
    if (x1 == nullptr || partial == nullptr || gamma == nullptr || beta == nullptr || scale == nullptr || shift == nullptr) return LD_ERR_ARG;
    if (C % 32 || C1 % 8 || C2 % 8 || C > 8192 || C1 <= 0 || (C2 > 0 && x2 == nullptr)) return LD_ERR_SHAPE;
    GnArgs a;
    a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.HW = HW;
    a.P = gn_num_chunks(n_img, HW);
    a.Pstat = stats_ready > 0 ? stats_ready : a.P;
    a.ppb = (HW + a.P - 1) / a.P;
    a.partial = partial; a.gamma = gamma; a.beta = beta; a.y = nullptr; a.eps = eps; a.silu = 0;
    a.slabs = gn_slabs(C);
    if (!stats_ready) hipLaunchKernelGGL(gn_stats_kernel, dim3(a.P, n_img, a.slabs), dim3(GN_THREADS), 0, stream, a);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_img, a.slabs), dim3(GN_THREADS), 0, stream, a, scale, shift);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int layernorm_launch(const half_t* x, const half_t* gamma, const half_t* beta, half_t* y, int rows, int C, float eps,
                     hipStream_t stream) {
    if (x == nullptr || y == nullptr || gamma == nullptr || beta == nullptr) return LD_ERR_ARG;
    if (C % 8 || C > 2048 || rows <= 0) return LD_ERR_SHAPE;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, rows, C, eps);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int softmax_rows_launch(half_t* s, int rows, int cols, long long ld, hipStream_t stream, int valid) {
    if (s == nullptr || rows <= 0 || cols % 8 || ld % 8 || valid > cols) return LD_ERR_SHAPE;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, s, cols, valid > 0 ? valid : cols, ld);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}
