// Dense contraction C[M,N] = epilogue(alpha * A[M,K] · W[N,K]^T) on MFMA (fp16 in, fp32 accumulate).
// One kernel family serves nn.Linear, 1x1 conv and — through an implicit-im2col A loader — the 3x3 convs
// (stride 1/2, fused nearest-upsample, fused channel concat of two NHWC sources).
#pragma once
#include "common.h"

struct GemmParams {
    // ---- A operand (activations, K contiguous)
    const half_t* A = nullptr;   // plain: [M][lda];  conv: NHWC source 1 [img][Hs][Ws][C1]
    const half_t* A2 = nullptr;  // conv only: NHWC source 2 [img][Hs][Ws][C2] (virtual channel concat), may be null
    int lda = 0;
    int conv = 0;                // 0 plain, 1 implicit im2col conv (ksize x ksize, pad ksize/2)
    int ksize = 3;               // 1 or 3
    int pad = -1;                // top/left zero padding; -1 = ksize/2 (bottom/right padding is implied by Ho, Wo)
    int Hs = 0, Ws = 0;          // source spatial size
    int Hv = 0, Wv = 0;          // size the conv sees (after nearest resize; == Hs,Ws without upsample)
    int Ho = 0, Wo = 0, stride = 1;
    int C1 = 0, C2 = 0;
    // ---- second K segment of a convolution (tap-major kernels): a 1x1 convolution over up to two RAW NHWC sources of the OUTPUT's spatial
    //   size, appended to the K axis — K = ksize^2 (C1 + C2) + SC1 + SC2, weight rows [conv taps | skip channels].  ResBlock1's
    //   skip_connection folded into out_layers' convolution (LD.py:5267, 5273-5287): out = W2 * gn(h) + Wskip x + (b2 + bskip) is ONE
    //   contraction; requires stride 1 and no resize.  The halo-tile and row-resident kernels decline it (gemm_conv_takes_skip_segment).
    const half_t* S1 = nullptr;
    const half_t* S2 = nullptr;
    int SC1 = 0, SC2 = 0;
    // ---- B operand (weights [N][K], K contiguous)
    const half_t* W = nullptr;
    int ldw = 0;
    int M = 0, N = 0, K = 0;
    int n_valid = 0;             // rows of W that exist (0 = N); rows n_valid..N-1 read as zeros (padded outputs)
    // ---- batching over blockIdx.z (element strides)
    int batch = 1;
    long long sA = 0, sW = 0, sC = 0, sR = 0;
    // ---- epilogue: v = alpha*acc; v += bias_n[n]; v += bias_m[m]; v += rowvec[m / rows_per_vec][n]; act; v += R[m][n]
    float alpha = 1.0f;
    const half_t* bias_n = nullptr;
    const half_t* bias_m = nullptr;
    const half_t* rowvec = nullptr;
    int rows_per_vec = 1, ldrv = 0;
    const half_t* R = nullptr;
    int ldr = 0;
    int act = 0;                 // 0 none, 1 SiLU, 3 quick-GELU, 2 GEGLU (weight rows tile-interleaved [BN/2 value | BN/2 gate], out width N/2)
    half_t* C = nullptr;
    int ldc = 0;
    // ---- tiling controls (0 = auto)
    int bm = 0, bn = 0;          // bn must match the repack-time choice for GEGLU
    int splitk = 0;
    int m_fastest = -1;          // tile order: -1 auto, 0 n fastest, 1 m fastest
    int xcd_gm = 0;              // (internal, 256 x 320 plain GEMM) XCD-blocked tile order: XCDs split gm x (8 / gm) over (M, N); 0 = off
    float* partial = nullptr;    // split-K workspace, >= splitk*M*N floats
    size_t partial_bytes = 0;
    // ---- LayerNorm folded into the contractions around it (v3 / v4 kernels, no split-K; unet.hip "ln fold"):
    //   producer (the GEMM that writes the residual stream): per output row and N tile, (sum, sum of squares) of the fp16 outputs
    float* stat_out = nullptr;       // [tiles_n][M][2] floats
    int* stat_parts_out = nullptr;   // host int: gemm_launch stores the number of N tiles (parts) it used
    //   consumer (a projection of LN(x), with gamma folded into W and beta into the bias at load time):
    //   out = rstd_row * (acc - mu_row * wsum[n]) + bias'[n]; with ln_swapped the LN rows are this GEMM's COLUMNS (V^T = Wv · x^T)
    const float* ln_stat = nullptr;  // the producer's stat_out
    int ln_parts = 0;                // its N tiles
    int ln_rows = 0;                 // its M (row stride of one part)
    int ln_zrows = 0;                // ln_swapped: LN rows per batch element z (row = z*ln_zrows + column)
    float ln_inv_c = 0.f, ln_eps = 0.f;
    const float* ln_wsum = nullptr;  // per output feature: sum_k of the folded weight row ([N], or [M] when ln_swapped)
    int ln_swapped = 0;
    // ---- GroupNorm(+SiLU) fused into the A operand of the halo-tile convolution (v6 only; see gemm_conv_fuses_groupnorm):
    //   the conv reads the RAW tensor and applies  y = x * scale[img][c] + shift[img][c]  (then SiLU) while the halo sits in LDS
    const float* gn_scale = nullptr;   // [n_img][C1 + C2] fp32: rstd * gamma
    const float* gn_shift = nullptr;   // [n_img][C1 + C2] fp32: beta - mean * rstd * gamma
    int gn_silu = 0;
    // ---- GroupNorm statistics of THIS contraction's output, emitted by its split-K second pass (the reduce kernel touches every output
    //   element anyway): per (image, pixel chunk, group) partial (sum, sum of squares) in exactly the layout, thread mapping and summation
    //   order of norm.hip's gn_stats_kernel, so the GroupNorm that follows skips its statistics launch and produces the same bits.
    //   Honoured only when the launch splits over K (otherwise *gn_part_done stays 0 and the caller runs the statistics pass).
    float* gn_part = nullptr;          // [n_img][gn_P][32][2] floats
    int gn_P = 0, gn_ppb = 0, gn_HW = 0;
    int* gn_part_done = nullptr;       // host int, set to the number of pixel chunks per image of the partials that were written (0: none)
    // ---- row-resident small-M convolution (conv8.hip)
    const half_t* W8 = nullptr;  // the weights in conv8's own layout (conv8_repack_launch): required for that kernel
    //   its in-launch reduction of the channel-slab partial sums needs >= 4 * (N / 80) zeroed ints that it leaves zeroed (self-resetting)
    int* sync = nullptr;
    int c8_S = 0;                // (internal) slab split chosen by conv8_plan
    int dbg = 0;                 // A/B build only (LD_AB_BUILD): ablation switches of the v5 kernel (timing runs, wrong results)
};

// conv8.hip: row-resident 3x3 convolution for the two-image (batch-1 CFG pair) 16x16 / 8x8 levels.  conv8_plan: does gemm_launch run this
// convolution there (p.partial and p.sync set)?  conv8_gn_chunks: pixel chunks per image of the GroupNorm partials it writes to gn_part
// (also stored to *gn_part_done).
bool conv8_plan(const GemmParams& p, int* S_out);
int conv8_gn_chunks(const GemmParams& p);
int conv8_launch(const GemmParams& p, hipStream_t stream);
// conv8's weight layout: [N / 80][Cin / 16][one 25 600-byte ring-stage image] from the general [O][ky][kx][I] layout
bool conv8_weight_eligible(int N, int Cin);
size_t conv8_weight_bytes(int N, int Cin);
int conv8_repack_launch(const half_t* w_okki, int N, int Cin, half_t* dst, hipStream_t stream);
#define LD_SYNC_INTS 1024        // ints a caller provides behind GemmParams::sync

bool gemm_ln_fold_available();   // the kernels that implement stat_out / ln_stat are the ones gemm_launch will pick
const char* gemm_last_kernel_name();   // kernel instantiation the calling thread's last gemm_launch dispatched

// BN the GEGLU weight interleave must use for a projection with N (=2*inner) output rows
static inline int gemm_pick_bn(int N) { return (N % 160 == 0) ? 160 : 128; }

int gemm_launch(const GemmParams& p, hipStream_t stream);
// true when gemm_launch would run this convolution on the halo-tile kernel, i.e. when it can take gn_scale / gn_shift
// (fill every other field first; gemm_launch rejects gn_scale on any other path)
bool gemm_conv_fuses_groupnorm(const GemmParams& p);
// does gemm_launch run this 3x3 convolution on a kernel that walks K tap-major (the 128 x 160 / 256 x 320 implicit-GEMM kernels), i.e. one
// that can take a second K segment (S1 / S2)?  Fill every field (partial, sync, W8 included) as for the launch itself, without the segment.
bool gemm_conv_takes_skip_segment(const GemmParams& p);
// does gemm_launch run this convolution on the halo-tile kernel?  (callers that need a property of it: the VAE's MFMA output conv)
bool gemm_conv_takes_halo_tile(const GemmParams& p);
