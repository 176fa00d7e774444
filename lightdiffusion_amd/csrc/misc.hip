// Small / bandwidth-bound kernels around the MFMA path: the 4-channel input convs, the <=4-channel output convs
// with their fused boundary math, the sigma -> timestep embedding, load-time weight repacks, sampler elementwise.
#include "kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------ conv_in
// 3x3 pad-1 conv, Cin <= 4, fp32 NCHW input (the sampler's latent) -> NHWC fp16.  Each thread: one pixel x 8 output
// channels; the [Cout][9*Cin] filter bank sits in LDS.  Fuses EPS.calculate_input (x / sqrt(sigma^2+1), rounded to
// fp16 like the reference's `.to(dtype)`, LD.py:5842) or the VAE's 1x1 post_quant_conv (LD.py:3470-3471).
template <int CIN>
__global__ __launch_bounds__(256) void small_conv_in_kernel(const SmallConvInArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    half_t* wl = reinterpret_cast<half_t*>(smem_raw);           // transposed filter bank [9*CIN][Cout]: 8 couts = one 16-byte read
    constexpr int KK = 9 * CIN;
    if ((KK & 3) == 0) {   // 8-byte reads of the row-major bank, scattered into the transposed copy (2-byte reads made this prologue most of the batch-1 launch)
        for (int i = threadIdx.x; i < a.Cout * KK / 4; i += blockDim.x) {
            const int o = (i * 4) / KK, k = i * 4 - o * KK;
            const half4 v = *reinterpret_cast<const half4*>(a.w + (long long)i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) wl[(k + e) * a.Cout + o] = v[e];
        }
    } else {
        for (int i = threadIdx.x; i < a.Cout * KK; i += blockDim.x) {
            const int o = i / KK, k = i - o * KK;
            wl[k * a.Cout + o] = a.w[i];
        }
    }
    float pw[CIN][CIN], pb[CIN];
    if (a.pre_w != nullptr) {
#pragma unroll
        for (int o = 0; o < CIN; ++o) {
            pb[o] = (float)a.pre_b[o];
#pragma unroll
            for (int c = 0; c < CIN; ++c) pw[o][c] = (float)a.pre_w[o * CIN + c];
        }
    }
    __syncthreads();
    const int cgroups = a.Cout >> 3;
    const long long total = (long long)a.N * a.H * a.W * cgroups;
    const long long plane = (long long)a.H * a.W;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(q % cgroups);          // channel group fastest: the lanes of a wave share the pixel -> broadcast loads
        long long pix = q / cgroups;
        const int x = (int)(pix % a.W);
        pix /= a.W;
        const int y = (int)(pix % a.H), n = (int)(pix / a.H);
        float inscale = 1.0f;
        if (a.scale_sigma != nullptr) {
            const float s = a.scale_sigma[n];
            inscale = 1.0f / sqrtf(s * s + 1.0f);
        }
        float acc[8];
        unpack8(ld16(a.b + cg * 8), acc);
        const float* xn = a.x + (long long)n * CIN * plane;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            if ((unsigned)iy >= (unsigned)a.H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x + kx - 1;
                if ((unsigned)ix >= (unsigned)a.W) continue;
                float v[CIN];
#pragma unroll
                for (int c = 0; c < CIN; ++c) v[c] = (float)(half_t)(xn[c * plane + (long long)iy * a.W + ix] * inscale);
                if (a.pre_w != nullptr) {
                    float t[CIN];
#pragma unroll
                    for (int o = 0; o < CIN; ++o) {
                        float s = pb[o];
#pragma unroll
                        for (int c = 0; c < CIN; ++c) s += pw[o][c] * v[c];
                        t[o] = (float)(half_t)s;
                    }
#pragma unroll
                    for (int c = 0; c < CIN; ++c) v[c] = t[c];
                }
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    float w8[8];
                    unpack8(ld16(wl + ((ky * 3 + kx) * CIN + c) * a.Cout + cg * 8), w8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[c] * w8[j];
                }
            }
        }
        const long long o = (((long long)n * a.H + y) * a.W + x) * a.Cout + cg * 8;
        const uint4 v8 = pack8(acc);
        st16(a.y + o, v8);
        if (a.dup_off != 0) st16(a.y + a.dup_off + o, v8);
    }
}

// ------------------------------------------------------------------------------------------------ conv_out
// 3x3 pad-1 conv to Cout <= 4 channels from NHWC fp16.  One wave per output pixel at a time; the 9*Cin reduction is split into
// 16-byte chunks and every lane OWNS the same chunks (tap, 8 channels) for all the pixels its wave visits, so the weights of
// those chunks live in registers for the whole kernel (the first version re-read 4 weight chunks per input chunk per pixel: 29 KB of
// cache traffic per pixel, 263 us for the 512x512 VAE output conv).  All input chunks of a pixel are loaded before the first
// FMA; wave-shuffle reduction; fused boundary math per `mode`.
// NCH = owned chunks per lane (9 * Cin / 8 / 64 rounded up), PX = pixels a wave works on at once (loads of all of them in flight).
template <int NCH, int PX>
__global__ __launch_bounds__(256) void small_conv_out_kernel(const SmallConvOutArgs a) {
    const int lane = threadIdx.x & 63;
    const long long npix = (long long)a.N * a.H * a.W;
    const int CH = a.Cin >> 3;           // chunks per tap
    const int KK = 9 * a.Cin;
    const int total = 9 * CH;
    int dy[NCH], dx[NCH], co[NCH];
    uint4 wv[NCH][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        const bool own = c < total;
        const int tap = own ? c / CH : 0, cc = own ? c - tap * CH : 0;
        dy[i] = tap / 3 - 1;
        dx[i] = tap % 3 - 1;
        co[i] = own ? cc * 8 : -1;
#pragma unroll
        for (int o = 0; o < 4; ++o)
            wv[i][o] = (own && o < a.Cout) ? ld16(a.w + (long long)o * KK + tap * a.Cin + cc * 8) : zero16();
    }
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long pix0 = wave * PX; pix0 < npix; pix0 += nwaves * PX) {
        uint4 xv[PX][NCH];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const long long pix = pix0 + q;
            const int x = (int)(pix % a.W);
            const int y = (int)((pix / a.W) % a.H);
            const int n = (int)(pix / ((long long)a.W * a.H));
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int iy = y + dy[i], ix = x + dx[i];
                const bool ok = pix < npix && co[i] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                xv[q][i] = ok ? ld16(a.x + (((long long)n * a.H + iy) * a.W + ix) * a.Cin + co[i]) : zero16();
            }
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const long long pix = pix0 + q;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NCH; ++i) {   // packed fp16 pairs, fp32 accumulate (v_dot2_f32_f16): weights stay packed in registers
                const unsigned xs[4] = {xv[q][i].x, xv[q][i].y, xv[q][i].z, xv[q][i].w};
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const unsigned ws[4] = {wv[i][o].x, wv[i][o].y, wv[i][o].z, wv[i][o].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[o] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, xs[j]), __builtin_bit_cast(half2v, ws[j]), acc[o], false);
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = wave_sum(acc[o]);
            if (lane < a.Cout && pix < npix) {
                const int x = (int)(pix % a.W);
                const int y = (int)((pix / a.W) % a.H);
                const int n = (int)(pix / ((long long)a.W * a.H));
                const int o = lane;
                float v = acc[0];
                if (o == 1) v = acc[1];
                if (o == 2) v = acc[2];
                if (o == 3) v = acc[3];
                v += (float)a.b[o];
                const long long nchw = (((long long)n * a.Cout + o) * a.H + y) * a.W + x;
                if (a.mode == 0) {
                    const float eps = (float)(half_t)v;   // the reference's UNet output is an fp16 tensor (.float() after)
                    const int ni = a.in_mod > 0 ? n % a.in_mod : n;
                    a.out[nchw] = a.x_in[(((long long)ni * a.Cout + o) * a.H + y) * a.W + x] - eps * a.sigma[ni];
                } else if (a.mode == 1) {
                    a.out[pix * a.Cout + o] = fminf(fmaxf((v + 1.0f) * 0.5f, 0.f), 1.f);
                } else {
                    a.out[nchw] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ tiny 1x1 conv
// out[n][o][pix] (fp32 NCHW) = bias[o] + sum_c w[o][c] * x[n][pix][c] (fp16 NHWC), C <= 8: the VAE's quant_conv (LD.py:3468, 3478)
__global__ void small_pointwise_kernel(const half_t* x, const half_t* w, const half_t* b, float* out, int N, int HW, int C) {
    const long long total = (long long)N * HW;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(q / HW);
        const long long pix = q - (long long)n * HW;
        float v[8];
        unpack8(ld16(x + q * 8), v);           // rows are padded to 8 channels by the caller (C == 8 here)
        for (int o = 0; o < C; ++o) {
            float s = (float)b[o];
            for (int c = 0; c < C; ++c) s += (float)w[o * C + c] * (float)(half_t)v[c];
            out[((long long)n * C + o) * HW + pix] = (float)(half_t)s;
        }
    }
}

__global__ void vae_out_finish_kernel(const half_t* __restrict__ t8, float* __restrict__ out, long long npix, int cout) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < npix; q += (long long)gridDim.x * blockDim.x) {
        float v[8];
        unpack8(ld16(t8 + q * 8), v);
        for (int o = 0; o < cout; ++o) out[q * cout + o] = fminf(fmaxf((v[o] + 1.0f) * 0.5f, 0.0f), 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------ timestep embedding
// ModelSamplingDiscrete.timestep (LD.py:1336-1339) + timestep_embedding (LD.py:803-812).  One block per sample.
__global__ __launch_bounds__(256) void timestep_embed_kernel(const float* sigma, const float* log_sigmas, int n_sig, int dim,
                                                             half_t* out, float* t_out, int sigma_mod) {
    __shared__ float bd[256];
    __shared__ int bi[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float ls = logf(sigma[sigma_mod > 0 ? n % sigma_mod : n]);
    float best = INFINITY;
    int besti = 0;
    for (int i = tid; i < n_sig; i += 256) {
        const float dd = fabsf(ls - log_sigmas[i]);
        if (dd < best) {
            best = dd;
            besti = i;
        }
    }
    bd[tid] = best;
    bi[tid] = besti;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            // ties resolve to the lowest index, like torch.argmin's first-occurrence rule
            if (bd[tid + s] < bd[tid] || (bd[tid + s] == bd[tid] && bi[tid + s] < bi[tid])) {
                bd[tid] = bd[tid + s];
                bi[tid] = bi[tid + s];
            }
        }
        __syncthreads();
    }
    const float t = (float)bi[0];
    if (tid == 0 && t_out != nullptr) t_out[n] = t;
    const int half_dim = dim / 2;
    for (int i = tid; i < half_dim; i += 256) {
        const float f = expf(-9.210340371976184f * (float)i / (float)half_dim);   // ln(10000)
        const float arg = t * f;
        out[(long long)n * dim + i] = (half_t)cosf(arg);
        out[(long long)n * dim + half_dim + i] = (half_t)sinf(arg);
    }
}

// ------------------------------------------------------------------------------------------------ CFG pair hand-over
// each range's first half -> its second half, 16 bytes per lane and four of them in flight (one launch instead of one copy node per range).
// Round 6: blockIdx.y = the range (no per-chunk range search), every load unconditional from a clamped index and only the stores predicated —
// the first form put each of a thread's four loads behind its own branch (DESIGN "hipcc traps" (a): a wait per load) and took 30 us for
// 8 MB as for 63 MB.
__global__ __launch_bounds__(256) void dup_halves_kernel(const DupArgs a) {
    const int r = blockIdx.y;
    const char* src = r == 0 ? a.base[0] : (r == 1 ? a.base[1] : a.base[2]);
    const unsigned long long nb = r == 0 ? a.bytes[0] : (r == 1 ? a.bytes[1] : a.bytes[2]);
    const long long n = (long long)(nb / 16);
    char* dst = const_cast<char*>(src) + nb;
    for (long long q = (long long)blockIdx.x * 1024 + threadIdx.x; q < n; q += (long long)gridDim.x * 1024) {
        uint4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long i = q + e * 256;
            v[e] = *reinterpret_cast<const uint4*>(src + (i < n ? i : n - 1) * 16);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long long i = q + e * 256;
            if (i < n) *reinterpret_cast<uint4*>(dst + i * 16) = v[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------ repack
template <typename T>
__global__ void repack_conv3x3_kernel(const T* src, int O, int I, half_t* dst) {
    const long long total = (long long)O * I * 9;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(q % I);
        const int tap = (int)((q / I) % 9);
        const int o = (int)(q / ((long long)I * 9));
        dst[q] = (half_t)(float)src[((long long)o * I + i) * 9 + tap];
    }
}

// copy/convert a [rows][cols] matrix; with geglu_bn > 0 the rows are tile-interleaved for the fused GEGLU epilogue:
// output tile t (bn rows) = [bn/2 value rows t*bn/2.. | bn/2 gate rows rows/2 + t*bn/2 ..]
template <typename T>
__global__ void repack_rows_kernel(const T* src, int rows, int cols, half_t* dst, int bn) {
    const long long total = (long long)rows * cols;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(q % cols);
        const int rdst = (int)(q / cols);
        int rsrc = rdst;
        if (bn > 0) {
            const int hb = bn / 2, tile = rdst / bn, within = rdst - tile * bn;
            rsrc = within < hb ? tile * hb + within : rows / 2 + tile * hb + (within - hb);
        }
        dst[q] = (half_t)(float)src[(long long)rsrc * cols + c];
    }
}

// context [n][T][D] (fp32 or fp16) -> fp16 [n][Tp][D], rows T..Tp-1 zero (keeps padded keys finite)
template <typename T>
__global__ void ctx_pad_kernel(const T* src, int n, int Tk, int Tp, int D, half_t* dst) {
    const long long total = (long long)n * Tp * D;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int dcol = (int)(q % D);
        const int t = (int)((q / D) % Tp);
        const int b = (int)(q / ((long long)D * Tp));
        dst[q] = t < Tk ? (half_t)(float)src[((long long)b * Tk + t) * D + dcol] : (half_t)0.f;
    }
}

__global__ void fill_half_kernel(half_t* dst, size_t n, float v) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) dst[q] = (half_t)v;
}

// ------------------------------------------------------------------------------------------------ sampler elementwise
__global__ void cfg_combine_kernel(const float* den2, float* out, float cfg, size_t n) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        const float u = den2[q], c = den2[n + q];
        out[q] = u + (c - u) * cfg;   // cfg_function, LD.py:2605
    }
}

__global__ void axpby_kernel(float* x, float a, const float* y, float b, const float* z, float c, size_t n) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        float v = a * x[q];
        if (y != nullptr) v += b * y[q];
        if (z != nullptr) v += c * z[q];
        x[q] = v;
    }
}

// The wrapper hook's device-side guards (unet.py: MI355XUNet.__call__): has the conditioning the resident cross-attention K / V^T were
// projected from changed, and are the two halves of the [uncond, cond] batch the same latents and sigmas (calc_cond_batch's cat([x, x]),
// LD.py:2515-2547)?  A mismatch stores `epoch` (the host's call counter) into flags[0] / flags[1]: every writer stores the same value, so the
// plain stores need no atomics and the flags are never reset.  The operands are compared as raw 32-bit words.
__global__ void hook_check_kernel(const unsigned* a, const unsigned* b, size_t words_ab, const unsigned* x, size_t half_words_x,
                                  const unsigned* sig, int half_sig, int* flags, int epoch) {
    bool d0 = false, d1 = false;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < words_ab; q += (size_t)gridDim.x * blockDim.x) d0 |= a[q] != b[q];
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < half_words_x; q += (size_t)gridDim.x * blockDim.x)
        d1 |= x[q] != x[half_words_x + q];
    if (blockIdx.x == 0 && (int)threadIdx.x < half_sig) d1 |= sig[threadIdx.x] != sig[half_sig + threadIdx.x];
    if (__any(d0) && (threadIdx.x & 63) == 0) flags[0] = epoch;
    if (__any(d1) && (threadIdx.x & 63) == 0) flags[1] = epoch;
}

inline int grid_for(long long total, int threads, int cap = 4096) {
    long long b = (total + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

int small_conv_in_launch(const SmallConvInArgs& a, hipStream_t stream) {
    if (a.x == nullptr || a.w == nullptr || a.b == nullptr || a.y == nullptr) return LD_ERR_ARG;
    if (a.Cin < 1 || a.Cin > 4 || (a.Cout & 7) || a.Cout <= 0) return LD_ERR_SHAPE;
    const size_t lds = (size_t)a.Cout * 9 * a.Cin * sizeof(half_t);
    if (lds > 64 * 1024) return LD_ERR_SHAPE;
    const long long total = (long long)a.N * a.H * a.W * (a.Cout >> 3);
    const dim3 grid(grid_for(total, 256, 1024));
    switch (a.Cin) {
        case 1: hipLaunchKernelGGL(small_conv_in_kernel<1>, grid, dim3(256), lds, stream, a); break;
        case 2: hipLaunchKernelGGL(small_conv_in_kernel<2>, grid, dim3(256), lds, stream, a); break;
        case 3: hipLaunchKernelGGL(small_conv_in_kernel<3>, grid, dim3(256), lds, stream, a); break;
        default: hipLaunchKernelGGL(small_conv_in_kernel<4>, grid, dim3(256), lds, stream, a); break;
    }
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int small_conv_out_launch(const SmallConvOutArgs& a, hipStream_t stream) {
    if (a.x == nullptr || a.w == nullptr || a.b == nullptr || a.out == nullptr) return LD_ERR_ARG;
    if (a.Cout < 1 || a.Cout > 4 || (a.Cin & 7)) return LD_ERR_SHAPE;
    if (a.mode == 0 && (a.x_in == nullptr || a.sigma == nullptr)) return LD_ERR_ARG;
    const int nch = (9 * (a.Cin >> 3) + 63) / 64;        // chunks a lane owns
    if (nch > 9) return LD_ERR_SHAPE;                    // Cin <= 512
    const long long npix = (long long)a.N * a.H * a.W;
    long long waves = npix / 16;                         // >= 16 pixels per wave amortise the weight preload ...
    if (waves < 2048) waves = npix / 4 < 2048 ? npix / 4 : 2048;   // ... but a batch-1 latent (8192 pixels) must still fill the chip's 1024 SIMDs
    if (waves > 8192) waves = 8192;
    if (waves < 4) waves = 4;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (nch <= 3) hipLaunchKernelGGL((small_conv_out_kernel<3, 4>), grid, block, 0, stream, a);
    else if (nch <= 6) hipLaunchKernelGGL((small_conv_out_kernel<6, 2>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((small_conv_out_kernel<9, 1>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int vae_out_finish_launch(const half_t* t8, float* out, long long npix, int cout, hipStream_t stream) {
    if (t8 == nullptr || out == nullptr || cout < 1 || cout > 8) return LD_ERR_ARG;
    hipLaunchKernelGGL(vae_out_finish_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, stream, t8, out, npix, cout);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int small_pointwise_launch(const half_t* x, const half_t* w, const half_t* b, float* out, int N, int HW, int C, hipStream_t stream) {
    if (x == nullptr || w == nullptr || b == nullptr || out == nullptr) return LD_ERR_ARG;
    if (C != 8) return LD_ERR_SHAPE;
    hipLaunchKernelGGL(small_pointwise_kernel, dim3(grid_for((long long)N * HW, 256)), dim3(256), 0, stream, x, w, b, out, N, HW, C);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int timestep_embed_launch(const float* sigma, const float* log_sigmas, int n_sig, int N, int dim, half_t* out, float* t_out,
                          hipStream_t stream, int sigma_mod) {
    if (sigma == nullptr || log_sigmas == nullptr || out == nullptr || N <= 0 || (dim & 1) || sigma_mod < 0) return LD_ERR_ARG;
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(N), dim3(256), 0, stream, sigma, log_sigmas, n_sig, dim, out, t_out, sigma_mod);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int dup_halves_launch(const DupArgs& a, hipStream_t stream) {
    if (a.count < 1 || a.count > 3) return LD_ERR_ARG;
    unsigned long long most = 0;
    for (int i = 0; i < a.count; ++i) {
        if (a.base[i] == nullptr || a.bytes[i] == 0 || (a.bytes[i] & 15) || (reinterpret_cast<uintptr_t>(a.base[i]) & 15)) return LD_ERR_ARG;
        if (a.bytes[i] / 16 > most) most = a.bytes[i] / 16;
    }
    // grid: x = 1024-chunk blocks of the longest range (shorter ranges' surplus blocks fall through the loop), y = the range
    hipLaunchKernelGGL(dup_halves_kernel, dim3(grid_for((long long)most, 1024, 4096), a.count), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int repack_conv3x3_launch(const void* src, int src_is_f32, int O, int I, half_t* dst, hipStream_t stream) {
    if (src == nullptr || dst == nullptr) return LD_ERR_ARG;
    const long long total = (long long)O * I * 9;
    if (src_is_f32)
        hipLaunchKernelGGL(repack_conv3x3_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const float*)src, O, I, dst);
    else
        hipLaunchKernelGGL(repack_conv3x3_kernel<half_t>, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const half_t*)src, O, I, dst);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int repack_rows_launch(const void* src, int src_is_f32, int rows, int cols, half_t* dst, int geglu_bn, hipStream_t stream) {
    if (src == nullptr || dst == nullptr) return LD_ERR_ARG;
    if (geglu_bn > 0 && (rows % geglu_bn)) return LD_ERR_SHAPE;
    const long long total = (long long)rows * cols;
    if (src_is_f32)
        hipLaunchKernelGGL(repack_rows_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const float*)src, rows, cols, dst, geglu_bn);
    else
        hipLaunchKernelGGL(repack_rows_kernel<half_t>, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const half_t*)src, rows, cols, dst, geglu_bn);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int ctx_pad_launch(const void* src, int src_is_f32, int n, int T, int Tp, int D, half_t* dst, hipStream_t stream) {
    if (src == nullptr || dst == nullptr || Tp < T) return LD_ERR_ARG;
    const long long total = (long long)n * Tp * D;
    if (src_is_f32)
        hipLaunchKernelGGL(ctx_pad_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const float*)src, n, T, Tp, D, dst);
    else
        hipLaunchKernelGGL(ctx_pad_kernel<half_t>, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const half_t*)src, n, T, Tp, D, dst);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int fill_half_launch(half_t* dst, size_t n, float v, hipStream_t stream) {
    if (dst == nullptr) return LD_ERR_ARG;
    hipLaunchKernelGGL(fill_half_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, stream, dst, n, v);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int cfg_combine_launch(const float* den2, float* out, float cfg, size_t n_half, hipStream_t stream) {
    if (den2 == nullptr || out == nullptr) return LD_ERR_ARG;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for((long long)n_half, 256)), dim3(256), 0, stream, den2, out, cfg, n_half);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int hook_check_launch(const void* a, const void* b, size_t words_ab, const void* x, size_t half_words_x, const void* sigma, int half_sigma,
                      int* flags, int epoch, hipStream_t stream) {
    if (flags == nullptr || (words_ab && (a == nullptr || b == nullptr)) || (half_words_x && x == nullptr) || (half_sigma && sigma == nullptr))
        return LD_ERR_ARG;
    if (half_sigma < 0 || half_sigma > 256) return LD_ERR_SHAPE;
    const long long total = (long long)(words_ab > half_words_x ? words_ab : half_words_x);
    hipLaunchKernelGGL(hook_check_kernel, dim3(grid_for(total > 0 ? total : 1, 256, 1024)), dim3(256), 0, stream, (const unsigned*)a,
                       (const unsigned*)b, words_ab, (const unsigned*)x, half_words_x, (const unsigned*)sigma, half_sigma, flags, epoch);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

int axpby_launch(float* x, float a, const float* y, float b, const float* z, float c, size_t n, hipStream_t stream) {
    if (x == nullptr) return LD_ERR_ARG;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, stream, x, a, y, b, z, c, n);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm fold (unet.hip): for a projection y = LN(x) W^T + b with LN(x) = (x - mu) * rstd * gamma + beta,
//   y = rstd * (x W'^T - mu * wsum) + b',   W' = W * diag(gamma),  wsum[n] = sum_k W'[n][k],  b'[n] = b[n] + sum_k W[n][k] * beta[k].
// One workgroup per weight row, fixed-order tree reduction (bitwise reproducible).  wsum is taken from the fp16-ROUNDED W' so that
// the subtraction in the consumer's epilogue cancels exactly what the matrix core accumulated.
namespace {
__global__ __launch_bounds__(256) void ln_fold_kernel(const half_t* __restrict__ W, int K, const half_t* __restrict__ gamma,
                                                      const half_t* __restrict__ beta, const half_t* __restrict__ bias, half_t* __restrict__ Wout,
                                                      half_t* __restrict__ bout, float* __restrict__ wsum) {
    __shared__ float red[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    float s = 0.f, sb = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float w = (float)W[(long long)n * K + k];
        const half_t wf = (half_t)(w * (float)gamma[k]);
        Wout[(long long)n * K + k] = wf;
        s += (float)wf;
        sb += w * (float)beta[k];
    }
    red[0][tid] = s;
    red[1][tid] = sb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        wsum[n] = red[0][0];
        bout[n] = (half_t)((bias != nullptr ? (float)bias[n] : 0.f) + red[1][0]);
    }
}
}  // namespace

int ln_fold_launch(const half_t* W, int N, int K, const half_t* gamma, const half_t* beta, const half_t* bias, half_t* Wout, half_t* bout,
                   float* wsum, hipStream_t stream) {
    if (W == nullptr || gamma == nullptr || beta == nullptr || Wout == nullptr || bout == nullptr || wsum == nullptr || N <= 0 || K <= 0)
        return LD_ERR_ARG;
    hipLaunchKernelGGL(ln_fold_kernel, dim3(N), dim3(256), 0, stream, W, K, gamma, beta, bias, Wout, bout, wsum);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// bislerp (LD.py:429-518, the hires-fix latent upscale): a 2-tap resize along ONE axis of an fp32 NCHW tensor whose blend
// of the two C-vectors slerps their direction and lerps their magnitude.  Called twice (width, then height), like the
// reference.  HBM-bound and tiny (a [4,4,128,128] latent); one thread per output pixel, channels looped (read twice).
// Tap positions follow F.interpolate(arange, mode="bilinear", align_corners=False) as generate_bilinear_data builds them.
namespace {
__global__ __launch_bounds__(256) void bislerp_axis_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int c, int h,
                                                           int w, int len_new, int axis_w) {
    const int ho = axis_w ? h : len_new, wo = axis_w ? len_new : w;
    const long long total = (long long)n * ho * wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(idx % wo), yo = (int)((idx / wo) % ho), b = (int)(idx / ((long long)wo * ho));
    const int len_old = axis_w ? w : h, o = axis_w ? xo : yo;
    float pos = ((float)o + 0.5f) * ((float)len_old / (float)len_new) - 0.5f;
    pos = fmaxf(pos, 0.f);
    int lo = (int)floorf(pos);
    if (lo > len_old - 1) lo = len_old - 1;
    const int hi = lo + 1 < len_old ? lo + 1 : len_old - 1;
    const float r = lo >= len_old - 1 ? 0.f : pos - (float)lo;
    const long long plane = (long long)h * w;
    const float* pa = x + (long long)b * c * plane + (axis_w ? (long long)yo * w + lo : (long long)lo * w + xo);
    const float* pb = x + (long long)b * c * plane + (axis_w ? (long long)yo * w + hi : (long long)hi * w + xo);
    float na = 0.f, nb = 0.f, dot = 0.f;
    for (int ch = 0; ch < c; ++ch) {
        const float a = pa[ch * plane], bb = pb[ch * plane];
        na += a * a;
        nb += bb * bb;
        dot += a * bb;
    }
    na = sqrtf(na);
    nb = sqrtf(nb);
    const float ia = na > 0.f ? 1.f / na : 0.f, ib = nb > 0.f ? 1.f / nb : 0.f;
    dot *= ia * ib;                                        // cosine of the angle between the two normalised vectors
    const float om = acosf(dot), so = sinf(om);
    const float wa = sinf((1.f - r) * om) / so * ia, wb = sinf(r * om) / so * ib;
    const float mag = na * (1.f - r) + nb * r;
    const long long oplane = (long long)ho * wo;
    float* py = y + (long long)b * c * oplane + (long long)yo * wo + xo;
    for (int ch = 0; ch < c; ++ch) {
        const float a = pa[ch * plane], bb = pb[ch * plane];
        float v = (wa * a + wb * bb) * mag;
        if (dot > 1.f - 1e-5f) v = a;                      // same direction
        if (dot < 1e-5f - 1.f) v = a * (1.f - r) + bb * r; // polar opposites
        py[ch * oplane] = v;
    }
    }
}
}  // namespace

int bislerp_launch(const float* x, float* tmp, float* y, int n, int c, int h, int w, int h_new, int w_new, hipStream_t stream) {
    if (x == nullptr || tmp == nullptr || y == nullptr || n <= 0 || c <= 0 || h <= 0 || w <= 0 || h_new <= 0 || w_new <= 0) return LD_ERR_ARG;
    hipLaunchKernelGGL(bislerp_axis_kernel, dim3(grid_for((long long)n * h * w_new, 256)), dim3(256), 0, stream, x, tmp, n, c, h, w, w_new, 1);
    hipLaunchKernelGGL(bislerp_axis_kernel, dim3(grid_for((long long)n * h_new * w_new, 256)), dim3(256), 0, stream, tmp, y, n, c, h, w_new,
                       h_new, 0);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// MLP-out fold (unet.hip): the feed-forward's second Linear and the SpatialTransformer's proj_out are back-to-back affine maps
//   t' = ff W2^T + b2 + t;   out = t' Wpo^T + bpo + x      (LD.py:3924-3928, 4259-4262)
// so   out = [ff | t] [Wpo W2 | Wpo]^T + (Wpo b2 + bpo) + x   is ONE contraction over K = 4C + C with a two-source A operand.
// This kernel derives W'[n][0:4C] = sum_c Wpo[n][c] W2[c][:], W'[n][4C:5C] = Wpo[n][:] and b'[n] once per weight load (fp32
// accumulation in a fixed order, one rounding to fp16).
namespace {
__global__ __launch_bounds__(256) void mlp_out_fold_kernel(const half_t* __restrict__ Wpo, const half_t* __restrict__ W2, const half_t* __restrict__ b2,
                                                           const half_t* __restrict__ bpo, int C, half_t* __restrict__ Wout, half_t* __restrict__ bout) {
    extern __shared__ float wrow[];          // Wpo[n][:] as fp32
    const int n = blockIdx.x, tid = threadIdx.x;
    const int H = 4 * C, K = 5 * C;
    for (int c = tid; c < C; c += 256) wrow[c] = (float)Wpo[(long long)n * C + c];
    __syncthreads();
    for (int k = tid; k < H; k += 256) {     // consecutive threads read consecutive k of every W2 row: coalesced
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += wrow[c] * (float)W2[(long long)c * H + k];
        Wout[(long long)n * K + k] = (half_t)s;
    }
    for (int c = tid; c < C; c += 256) Wout[(long long)n * K + H + c] = Wpo[(long long)n * C + c];
    if (tid == 0) {
        float s = (float)bpo[n];
        for (int c = 0; c < C; ++c) s += wrow[c] * (float)b2[c];
        bout[n] = (half_t)s;
    }
}
}  // namespace

int mlp_out_fold_launch(const half_t* Wpo, const half_t* W2, const half_t* b2, const half_t* bpo, int C, half_t* Wout, half_t* bout, hipStream_t stream) {
    if (Wpo == nullptr || W2 == nullptr || b2 == nullptr || bpo == nullptr || Wout == nullptr || bout == nullptr || C <= 0) return LD_ERR_ARG;
    hipLaunchKernelGGL(mlp_out_fold_kernel, dim3(C), dim3(256), (size_t)C * sizeof(float), stream, Wpo, W2, b2, bpo, C, Wout, bout);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// Skip fold (unet.hip): ResBlock1 with a 1x1 skip_connection computes  out = conv3x3(h; W2) + b2 + conv1x1(x; Wsk) + bsk  (LD.py:5267,
// 5273-5287) — one contraction over K = 9 Cout + Cin with the skip sources as a second K segment (gemm.h S1 / S2).  This kernel derives
// W'[n] = [W2[n][tap][c] | Wsk[n][:]] and b' = b2 + bsk (fp32 sum, one rounding) once per weight load.
namespace {
__global__ __launch_bounds__(256) void skip_fold_kernel(const uint4* __restrict__ W2, const uint4* __restrict__ Wsk, const half_t* __restrict__ b2,
                                                        const half_t* __restrict__ bsk, int N, int K9c, int SCc, uint4* __restrict__ Wout, half_t* __restrict__ bout) {
    const int Kc = K9c + SCc;                     // 16-byte chunks per output row
    const long long total = (long long)N * Kc;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(q / Kc), c = (int)(q - (long long)n * Kc);
        Wout[q] = c < K9c ? W2[(long long)n * K9c + c] : Wsk[(long long)n * SCc + (c - K9c)];
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N) bout[t] = (half_t)((float)b2[t] + (float)bsk[t]);
}
}  // namespace

int skip_fold_launch(const half_t* W2, const half_t* Wsk, const half_t* b2, const half_t* bsk, int N, int K9, int SC, half_t* Wout, half_t* bout,
                     hipStream_t stream) {
    if (W2 == nullptr || Wsk == nullptr || b2 == nullptr || bsk == nullptr || Wout == nullptr || bout == nullptr || N <= 0 || K9 <= 0 || SC <= 0 ||
        (K9 & 7) || (SC & 7))
        return LD_ERR_ARG;
    const long long total = (long long)N * ((K9 + SC) / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks * 256 < N) blocks = (N + 255) / 256;
    hipLaunchKernelGGL(skip_fold_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint4*>(W2), reinterpret_cast<const uint4*>(Wsk), b2, bsk, N,
                       K9 / 8, SC / 8, reinterpret_cast<uint4*>(Wout), bout);
    return hipGetLastError() == hipSuccess ? LD_OK : LD_ERR_HIP;
}
