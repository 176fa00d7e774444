/* ld_mi355x.h — C ABI of the MI355X-native SD1.5 denoise hot path (libld_mi355x.so).
 *
 * Plain C: opaque handles, raw DEVICE pointers, sizes, a hipStream_t passed as void*.  No torch types.
 * Every function returns an LD_* status (0 = OK); nothing throws across the boundary.  After `*_reserve`
 * a handle performs no allocation: `ld_unet_forward` / `ld_vae_decode` only enqueue kernels on the given
 * stream (safe to capture into a hipGraph: fixed workspace addresses, no host sync, no malloc).
 *
 * What each entry point replaces in the reference (file:line into LightDiffusion.py = LD.py):
 *   ld_unet_forward      BaseModel.apply_model (LD.py:5828-5860) = EPS.calculate_input (1259-1261) →
 *                        ModelSamplingDiscrete.timestep (1336-1339) → UNetModel1.forward (5688-5767) →
 *                        EPS.calculate_denoised (1263-1265); i.e. what a `model_function_wrapper`
 *                        (LD.py:2558-2567, installed by ModelPatcher.set_model_unet_function_wrapper 3277)
 *                        must return: denoised x0, fp32, same shape as the input.
 *   ld_unet_set_context  the per-layer to_k / to_v projections of the cross-attention context
 *                        (CrossAttention.forward LD.py:4028-4036) — step-invariant, so hoisted out of the step.
 *   ld_vae_decode        VAE.decode (LD.py:6357-6381) = post_quant_conv + Decoder.forward (3470-3473, 3857-3882)
 *                        + process_output clamp + NCHW→NHWC.
 *   ld_op_*              single operators, for parity tests: the `operations=` classes of LD.py:2342-2429
 *                        (Linear / Conv2d / GroupNorm / LayerNorm) and optimized_attention (3966-3988).
 */
#ifndef LD_MI355X_H
#define LD_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LD_OK 0
#define LD_ERR_ARG 1    /* null / inconsistent argument */
#define LD_ERR_SHAPE 2  /* shape or alignment not supported by the kernels */
#define LD_ERR_HIP 3    /* a HIP call or launch failed */
#define LD_ERR_STATE 4  /* call order (missing weights / reserve / context) */

#define LD_F16 0
#define LD_F32 1

const char* ld_version(void);
const char* ld_status_string(int status);

/* ------------------------------------------------------------------ UNet (UNetModel1 ctor arguments, LD.py:5294-5340) */
typedef struct {
    int in_channels, out_channels, model_channels;
    int num_levels;
    int channel_mult[8];
    int num_res_blocks[8];
    int transformer_depth[16];        /* one per input ResBlock */
    int transformer_depth_output[24]; /* one per output ResBlock, in the reference's (popped-from-the-end) list order */
    int transformer_depth_middle;
    int context_dim, num_heads;
} ld_unet_config;

typedef struct ld_unet ld_unet;

int ld_unet_create(const ld_unet_config* cfg, ld_unet** out);
void ld_unet_destroy(ld_unet* u);
/* parameter table: checkpoint key names (minus "model.diffusion_model."), shapes as stored in the checkpoint */
int ld_unet_param_count(const ld_unet* u);
int ld_unet_param_info(const ld_unet* u, int index, const char** name, int* ndim, int64_t shape[4]);
/* copy + repack one checkpoint tensor (device pointer, LD_F16 or LD_F32, checkpoint layout: conv OIHW, linear [out,in]) */
int ld_unet_load_param(ld_unet* u, const char* name, const void* dev_src, int dtype, void* stream);
/* size the activation workspace for up to max_n UNet samples (= 2 x image batch under CFG) of max_h x max_w latents */
int ld_unet_reserve(ld_unet* u, int max_n, int max_h, int max_w, int max_ctx_tokens);
size_t ld_unet_workspace_bytes(const ld_unet* u);
size_t ld_unet_weight_bytes(const ld_unet* u);
/* ctx: [n][tokens][context_dim] (LD_F16 / LD_F32), batch order as the reference builds it: [uncond..., cond...] */
int ld_unet_set_context(ld_unet* u, const void* ctx, int dtype, int n, int tokens, void* stream);
/* x, out: [n][in_channels][h][w] fp32 NCHW; sigma: [n] fp32 (sigma, not t).  out = denoised = x - eps * sigma.
 * eps_only != 0 writes the raw UNet output (fp32 of the fp16 eps) instead. */
int ld_unet_forward(ld_unet* u, const float* x, const float* sigma, float* out, int n, int h, int w, int eps_only, void* stream);
/* Classifier-free-guidance pair — what the reference's calc_cond_batch feeds the model every step: cat([x, x]) against cat([uncond, cond]) (LD.py:2515-2547).
 * x: [nb][in_channels][h][w], sigma: [nb]; the resident context has 2 nb rows ([uncond x nb ; cond x nb]); out: [2 nb][..] denoised, same order.
 * Same result as ld_unet_forward on the duplicated inputs; the layers in front of the first cross-attention (conv_in, the first ResBlock, the first
 * transformer's GroupNorm / proj_in / self-attention) see identical inputs in both halves and are evaluated ONCE, their outputs copied (exact; per sample
 * the bits can differ from ld_unet_forward where a contraction's tile / split choice follows the row count). */
int ld_unet_forward_pair(ld_unet* u, const float* x, const float* sigma, float* out, int nb, int h, int w, void* stream);
/* one forward with a HIP-event pair around every launch (recorded on `stream`), summed per kernel class:
 * 0 conv3x3 (implicit GEMM)  1 linear / 1x1 GEMM  2 attention  3 GroupNorm  4 LayerNorm  5 misc.  Synchronises the stream. */
int ld_unet_profile(ld_unet* u, const float* x, const float* sigma, float* out, int n, int h, int w, void* stream, double ms[6],
                    double flops[6], int launches[6]);
/* the same for ld_unet_forward_pair (x, sigma: nb samples; out: 2 nb) */
int ld_unet_profile_pair(ld_unet* u, const float* x, const float* sigma, float* out, int nb, int h, int w, void* stream, double ms[6],
                         double flops[6], int launches[6]);
/* per kernel INSTANTIATION (the names rocprofv3 --kernel-trace lists, abbreviated) of the last ld_unet_profile call:
 * one text line "name<TAB>launches<TAB>total ms<TAB>algorithmic FLOPs" each, NUL-terminated.  LD_ERR_ARG if buf is too small. */
int ld_unet_profile_kernels(const ld_unet* u, char* buf, size_t buf_bytes);
/* per LAUNCH of the last ld_unet_profile call, in launch order: "what<TAB>M<TAB>N<TAB>K<TAB>batch<TAB>algorithmic FLOPs<TAB>
 * microseconds<TAB>kernel" (contractions; GroupNorm rows carry images / pixels / channels / silu, attention rows batch*heads / Lq / Lk / d) */
int ld_unet_profile_launches(const ld_unet* u, char* buf, size_t buf_bytes);
/* number of kernel launches of the last forward, and algorithmic FLOPs of it (2*M*N*K over every contraction) */
int ld_unet_last_launches(const ld_unet* u);
double ld_unet_last_flops(const ld_unet* u);

/* ------------------------------------------------------------------ VAE decoder (Decoder ctor arguments, LD.py:6312-6323) */
typedef struct {
    int z_channels, ch, num_levels;
    int ch_mult[8];
    int num_res_blocks, out_ch;
    int with_encoder;   /* != 0: also hold encoder.* and quant_conv.* (VAE.encode path) */
} ld_vae_config;

typedef struct ld_vae ld_vae;

int ld_vae_create(const ld_vae_config* cfg, ld_vae** out);
void ld_vae_destroy(ld_vae* v);
int ld_vae_param_count(const ld_vae* v);
int ld_vae_param_info(const ld_vae* v, int index, const char** name, int* ndim, int64_t shape[4]);
int ld_vae_load_param(ld_vae* v, const char* name, const void* dev_src, int dtype, void* stream);
int ld_vae_reserve(ld_vae* v, int max_b, int max_h, int max_w);   /* latent size */
size_t ld_vae_workspace_bytes(const ld_vae* v);
/* workspace bytes ld_vae_reserve(b, h, w) would allocate (host-only dry run, nothing is allocated; 0 on an invalid shape): lets the host
 * split a batch by free device memory the way VAE.decode does (LD.py:6357-6362) */
size_t ld_vae_plan_bytes(ld_vae* v, int b, int h, int w);
/* z: [b][z_channels][h][w] fp32 (already divided by 0.18215); out: [b][8h][8w][3] fp32 in [0,1] */
int ld_vae_decode(ld_vae* v, const float* z, float* out, int b, int h, int w, void* stream);
/* VAE.encode's device part (LD.py:6383-6410): pixels fp32 NCHW [b][3][8h][8w] in [-1,1] -> moments fp32 NCHW [b][2z][h][w]
 * (mean | logvar, before DiagonalGaussianRegularizer's host-RNG sample, LD.py:3446-3458); (h, w) = latent size */
int ld_vae_encode(ld_vae* v, const float* pixels_nchw, float* moments, int b, int h, int w, void* stream);
/* ld_vae_decode with a HIP-event pair around every launch (recorded on `stream`; synchronises it), and the per-launch table of that
 * run in the format of ld_unet_profile_launches: the per-layer VAE table of profiles/ (TFLOP/s and bytes per stage) is built from it */
int ld_vae_profile(ld_vae* v, const float* z, float* out, int b, int h, int w, void* stream);
int ld_vae_profile_launches(const ld_vae* v, char* buf, size_t buf_bytes);
int ld_vae_last_launches(const ld_vae* v);
double ld_vae_last_flops(const ld_vae* v);

/* ------------------------------------------------------------------ single operators (fp16 device tensors unless noted) */
/* y[M][N] = act(alpha * x[M][K] · w[N][K]^T + bias[N]) + residual[M][N];  act: 0 none, 1 SiLU, 3 quick-GELU, 2 GEGLU (w, bias in
 * checkpoint row order [value | gate]; y is [M][N/2]).  ws/ws_bytes: optional split-K scratch. */
int ld_op_linear(const void* x, const void* w, const void* bias, const void* residual, void* y, int M, int N, int K,
                 float alpha, int act, void* ws, size_t ws_bytes, void* stream);
/* NHWC conv, ksize 1 or 3 (pad ksize/2), w in [Cout][ky][kx][Cin] order (see ld_op_repack_conv).  Two optional
 * NHWC sources are concatenated along channels; (hv, wv) != (h, w) resizes the input nearest-neighbour first. */
int ld_op_conv(const void* x1, int c1, const void* x2, int c2, int n, int h, int w, int hv, int wv, int stride, int ksize,
               const void* wt, const void* bias, const void* rowvec, const void* residual, void* y, int cout,
               void* ws, size_t ws_bytes, void* stream);
/* GroupNorm(32) + SiLU + 3x3 conv (stride 1, pad 1) over the channel concat of two NHWC sources — ResBlock1.in_layers /
 * out_layers (LD.py:5224-5262).  Where the conv runs on the halo-tile kernel the normalisation is fused into its A operand
 * (statistics pass only, no normalised tensor in HBM).  ws >= ld_op_groupnorm_conv_ws_bytes(...). */
size_t ld_op_groupnorm_conv_ws_bytes(int c1, int c2, int n, int h, int w, int cout);
int ld_op_groupnorm_conv(const void* x1, int c1, const void* x2, int c2, int n, int h, int w, const void* gamma, const void* beta, float eps,
                         const void* wt, const void* bias, const void* rowvec, const void* residual, void* y, int cout, void* ws,
                         size_t ws_bytes, void* stream);
int ld_op_repack_conv(const void* src_oihw, int dtype, int cout, int cin, void* dst, void* stream);
/* ResBlock1's out_layers convolution and its 1x1 skip_connection as ONE contraction (LD.py:5267, 5273-5287; the UNet executor's "skip
 * fold"):  y = conv3x3(x; wt) + bias + conv1x1(cat(s1, s2); wskip) + bskip (+ rowvec per image), stride 1, pad 1.  x [n][h][w][c];
 * s1 / s2 raw NHWC sources of the same spatial size with sc1 / sc2 channels (s2 may be NULL with sc2 = 0); wt [cout][9c] as ld_op_repack_conv
 * writes it; wskip [cout][sc1 + sc2].  The skip channels are a second segment of the K axis: the weights are concatenated to
 * [cout][9c + sc1 + sc2] in `ws` first (the executor keeps that copy resident).  ws >= ld_op_conv_skip_ws_bytes(...). */
size_t ld_op_conv_skip_ws_bytes(int c, int sc1, int sc2, int cout);
int ld_op_conv_skip(const void* x, int c, int n, int h, int w, const void* wt, const void* bias, const void* s1, int sc1, const void* s2, int sc2,
                    const void* wskip, const void* bskip, const void* rowvec, void* y, int cout, void* ws, size_t ws_bytes, void* stream);
/* 3x3 stride-1 convolution (hv = 2h: behind a nearest-2x upsampling) that also returns the GroupNorm(32) partial statistics of its OUTPUT
 * where the kernel that runs the shape writes them (the halo convolution's generic epilogue, the row-resident kernel, the split-K second
 * pass) — what lets the GroupNorm that follows (ResBlock1 out_layers / the next block's in_layers, LD.py:5224-5262; the VAE's ResnetBlock,
 * LD.py:3560-3576) skip its statistics pass.  part: [n][*chunks][32][2] floats (sum, sum of squares per image, pixel chunk, group), sized
 * ld_op_conv_gn_partials_floats(n, hv*wv); *chunks = 0 when this shape's kernel does not write them.  For parity tests. */
size_t ld_op_conv_gn_partials_floats(int n, int hw);
int ld_op_conv_gn_partials(const void* x, int c, int n, int h, int w, int hv, int wv, const void* wt, const void* bias,
                           const void* residual, void* y, int cout, float* part, int* chunks, void* ws, size_t ws_bytes, void* stream);
/* GroupNorm(32) over the channel concat of two NHWC sources (+ optional SiLU); ws >= ld_op_groupnorm_ws_bytes */
size_t ld_op_groupnorm_ws_bytes(int n, int hw);
int ld_op_groupnorm(const void* x1, int c1, const void* x2, int c2, int n, int hw, const void* gamma, const void* beta,
                    float eps, int silu, void* y, void* ws, void* stream);
int ld_op_layernorm(const void* x, const void* gamma, const void* beta, void* y, int rows, int c, float eps, void* stream);
/* q [b][lq][heads*d], k [b][lk][heads*d], vt [b][heads*d][lk_pad] (V transposed, lk_pad = ldvt >= lk, multiple of 8);
 * causal != 0 masks keys after the query position (the CLIP text model's mask, LD.py:4440-4446) */
int ld_op_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo, int b,
                    int heads, int lq, int lk, int d, float scale, int causal, void* stream);
/* the same with V row-major, v [b][lk][ldv] like k — the form the UNet executor runs on the output of its fused q|k|v projection
 * (q, k, v may be column blocks of one [b][l][3*heads*d] tensor: ldq = ldk = ldv = 3*heads*d; requires lq == lk then for the batch strides);
 * the kernel transposes V while it reads it from LDS (ds_read_b64_tr_b16) */
int ld_op_attention_rowv(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int b,
                         int heads, int lq, int lk, int d, float scale, int causal, void* stream);
int ld_op_softmax_rows(void* s, int rows, int cols, void* stream);
int ld_op_timestep_embed(const float* sigma, const float* log_sigmas, int n_sigmas, int n, int dim, void* out_f16, float* t_out,
                         void* stream);
/* guidance / sampler elementwise on fp32 latents: out = u + (c-u)*cfg with den2 = [u ; c];  x = a*x + b*y + c*z */
int ld_op_cfg_combine(const float* den2, float* out, float cfg, size_t n_half, void* stream);
int ld_op_axpby(float* x, float a, const float* y, float b, const float* z, float c, size_t n, void* stream);
/* Device-side guards of the model_function_wrapper hook (LD.py:2558-2567; the object on that seam replays a captured hipGraph the way the
 * reference's stable-fast patch does with enable_cuda_graph, LD.py:9896-9933): flags[0] = epoch when the 32-bit words of a and b differ
 * (the step's c_crossattn against the conditioning the resident cross-attention K / V^T were projected from), flags[1] = epoch when the two
 * halves of x (2 * half_words_x words) or of sigma (2 * half_sigma floats, half_sigma <= 256) differ (calc_cond_batch's cat([x_in, x_in]),
 * LD.py:2515-2547).  flags: two ints the device can write (device or pinned host memory); never reset — the host compares with its epoch. */
int ld_op_hook_check(const void* a, const void* b, size_t words_ab, const void* x, size_t half_words_x, const void* sigma, int half_sigma,
                     int* flags, int epoch, void* stream);

/* The LayerNorm fold of the UNet's transformer blocks (BasicTransformerBlock, LD.py:4117-4162: every attention / GEGLU
 * projection reads LayerNorm(x)) as an operator pair, for parity tests: t[M][C] = x · w_prod^T + b_prod (the GEMM that writes the
 * residual stream, emitting per-row statistics) and y[M][N] = LayerNorm(t; gamma, beta, eps) · w^T + bias, finished on the fp32
 * accumulators of t · (w diag(gamma))^T.  ws: >= 2*N*C + 8*N + 8*((C+63)/64)*M + 1024 bytes. */
int ld_op_linear_ln(const void* x, const void* w_prod, const void* b_prod, const void* gamma, const void* beta, const void* w,
                    const void* bias, void* t_out, void* y, int M, int C, int N, float eps, void* ws, size_t ws_bytes, void* stream);
/* The same pair with the GEGLU of FeedForward.net[0] as the consumer (LD.py:4513-4515, 4524-4540): y[M][N/2] = a * gelu(g) with
 * [a | g] = LayerNorm(t) · w^T + bias — the transformer block's MLP input exactly as the executor runs it (row-panel kernel at C = 320).
 * ws: >= 4*N*C + 12*N + 8*((C+63)/64)*M + 2048 bytes. */
int ld_op_linear_ln_geglu(const void* x, const void* w_prod, const void* b_prod, const void* gamma, const void* beta, const void* w,
                          const void* bias, void* t_out, void* y, int M, int C, int N, float eps, void* ws, size_t ws_bytes, void* stream);
/* bislerp (LD.py:429-518, LatentUpscale.upscale 6639-6654): fp32 NCHW latents [n][c][h][w] -> [n][c][h_new][w_new];
 * tmp: n*c*h*w_new floats of scratch (the width pass runs first, as in the reference) */
int ld_op_bislerp(const float* x, float* tmp, float* y, int n, int c, int h, int w, int h_new, int w_new, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LD_MI355X_H */
