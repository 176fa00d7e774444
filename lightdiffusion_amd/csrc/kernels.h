// Launcher declarations for the gfx950 kernels (internal to the shared library; the public C ABI is
// include/ld_mi355x.h).  Every launcher validates shapes on the host and returns an LD_* status.
#pragma once
#include <cstdlib>
#include "common.h"
#include "gemm.h"

// ---- norm.hip
// pixel chunks per image: aim at ~2048 blocks (x 4 channel slabs x n images), at least 8 pixels per chunk, at most 256 chunks
static inline int gn_num_chunks(int n_img, int HW) {
    // workgroups per launch, measured (profiles/README.md): 512 is best at UNet batch 2 (GroupNorm 0.92 -> 0.74 ms per forward),
    // 1024 at UNet batch 16 (1.51 -> 1.40 ms), 2048 for the 512x512 VAE tensors
    const long long px = (long long)(n_img < 1 ? 1 : n_img) * HW;   // (sized by pixels: a batch-1 VAE image is one huge tensor)
    const int target = px <= 8192 ? 512 : (px <= 131072 ? 1024 : 2048);
    int p = target / (4 * (n_img < 1 ? 1 : n_img));
    if (p > HW / 8) p = HW / 8;
    if (p > 256) p = 256;
    return p < 1 ? 1 : p;
}
size_t groupnorm_workspace_bytes(int n_img, int HW);
// stats_ready: 0 = run the statistics pass; > 0 = `partial` already holds a producer's partial statistics with that many pixel chunks per
// image ([n_img][stats_ready][32][2]: gemm.h gn_part / conv8) and only the apply pass runs
int groupnorm_launch(const half_t* x1, int C1, const half_t* x2, int C2, int n_img, int HW, const half_t* gamma,
                     const half_t* beta, float eps, int silu, half_t* y, float* partial, hipStream_t stream, int stats_ready = 0);
// the statistics pass alone: partial [n_img][gn_num_chunks(n_img, HW)][32][2] (for a consumer that finishes the normalisation itself: conv8)
int groupnorm_stats_launch(const half_t* x1, int C1, const half_t* x2, int C2, int n_img, int HW, float* partial, hipStream_t stream);
// statistics pass + finalize only: scale / shift [n_img][C1 + C2] fp32 for a consumer that normalises on the fly (gemm.h gn_scale)
int groupnorm_scale_shift_launch(const half_t* x1, int C1, const half_t* x2, int C2, int n_img, int HW, const half_t* gamma, const half_t* beta,
                                 float eps, float* partial, float* scale, float* shift, hipStream_t stream, int stats_ready = 0);
int layernorm_launch(const half_t* x, const half_t* gamma, const half_t* beta, half_t* y, int rows, int C, float eps,
                     hipStream_t stream);
// in-place row softmax over fp16 scores; cols and ld multiples of 8; valid (0 = cols): columns >= valid are padding (written as 0)
int softmax_rows_launch(half_t* s, int rows, int cols, long long ld, hipStream_t stream, int valid = 0);

// ---- attention.hip
struct AttnParams {
    const half_t* Q = nullptr;   // element (b, l, h, dd) at Q + b*sQ + l*ldq + h*d + dd
    const half_t* K = nullptr;   // element (b, key, h, dd) at K + b*sK + key*ldk + h*d + dd
    const half_t* Vt = nullptr;  // element (b, h, dd, key) at Vt + b*sV + (h*d + dd)*ldvt + key   (V transposed)
    const half_t* V = nullptr;   // ... or row-major: element (b, key, h, dd) at V + b*sV + key*ldv + h*d + dd  (exactly one of Vt / V)
    int ldv = 0;
    half_t* O = nullptr;         // element (b, l, h, dd) at O + b*sO + l*ldo + h*d + dd
    int ldq = 0, ldk = 0, ldvt = 0, ldo = 0;
    long long sQ = 0, sK = 0, sV = 0, sO = 0;
    int B = 0, H = 0, Lq = 0, Lk = 0, d = 0;
    float scale = 1.0f;
    int causal = 0;              // key index <= query index only (CLIP text model, LD.py:4440-4446)
};
int attention_launch(const AttnParams& p, hipStream_t stream);
const char* attention_last_kernel_name();   // instantiation the calling thread's last attention_launch dispatched

// ---- misc.hip
int ln_fold_launch(const half_t* W, int N, int K, const half_t* gamma, const half_t* beta, const half_t* bias, half_t* Wout, half_t* bout,
                   float* wsum, hipStream_t stream);
// W'[C][5C] = [Wpo W2 | Wpo], b' = Wpo b2 + bpo (ff.net.2 followed by proj_out as one contraction; see misc.hip)
int mlp_out_fold_launch(const half_t* Wpo, const half_t* W2, const half_t* b2, const half_t* bpo, int C, half_t* Wout, half_t* bout, hipStream_t stream);
// W'[N][K9 + SC] = [W2 | Wsk], b' = b2 + bsk (a ResBlock's skip_connection as a second K segment of its out_layers convolution; see misc.hip)
int skip_fold_launch(const half_t* W2, const half_t* Wsk, const half_t* b2, const half_t* bsk, int N, int K9, int SC, half_t* Wout, half_t* bout,
                     hipStream_t stream);
struct SmallConvInArgs {        // 3x3 pad-1 conv with <= 4 input channels from an fp32 NCHW tensor (conv_in of UNet / VAE)
    const float* x = nullptr;   // [N][Cin][H][W] fp32
    const float* scale_sigma = nullptr;  // optional [N]: input scaled by 1/sqrt(sigma^2+1) (EPS.calculate_input, LD.py:1259-1261)
    const half_t* pre_w = nullptr;       // optional 1x1 pre-conv [Cin][Cin] + bias (VAE post_quant_conv, LD.py:3467-3471)
    const half_t* pre_b = nullptr;
    const half_t* w = nullptr;  // [Cout][9*Cin] (tap-major, channel-minor)
    const half_t* b = nullptr;
    half_t* y = nullptr;        // NHWC [N][H][W][Cout]
    int N = 0, Cin = 0, H = 0, W = 0, Cout = 0;
    long long dup_off = 0;      // != 0: every output chunk is stored a second time at y + dup_off (elements): the second half of a CFG pair (unet.hip)
};
int small_conv_in_launch(const SmallConvInArgs& a, hipStream_t stream);

struct SmallConvOutArgs {       // 3x3 pad-1 conv to <= 4 output channels from an NHWC fp16 tensor
    const half_t* x = nullptr;  // [N][H][W][Cin]
    const half_t* w = nullptr;  // [Cout][9*Cin]
    const half_t* b = nullptr;
    int N = 0, H = 0, W = 0, Cin = 0, Cout = 0;
    int mode = 0;               // 0: UNet out -> denoised NCHW fp32 = x_in - fp16(eps)*sigma (EPS.calculate_denoised, LD.py:1263-1265)
                                // 1: VAE out  -> NHWC fp32 clamp((v+1)/2, 0, 1) (VAE.process_output, LD.py:6296-6298)
                                // 2: raw eps NCHW fp32 (tests)
    const float* x_in = nullptr;   // mode 0: [N][Cout][H][W] fp32
    const float* sigma = nullptr;  // mode 0: [N]
    float* out = nullptr;
    int in_mod = 0;                // > 0: x_in / sigma hold in_mod samples and sample n reads n % in_mod (CFG pair: both halves see the same latents)
};
int small_conv_out_launch(const SmallConvOutArgs& a, hipStream_t stream);

// VAE output tail behind the MFMA output convolution: t8 [npix][8] fp16 (first `cout` columns valid) -> out [npix][cout] fp32 =
// clamp((v + 1) / 2, 0, 1)  (VAE.process_output, LD.py:6296-6298)
int vae_out_finish_launch(const half_t* t8, float* out, long long npix, int cout, hipStream_t stream);

int small_pointwise_launch(const half_t* x, const half_t* w, const half_t* b, float* out, int N, int HW, int C, hipStream_t stream);

// timestep lookup + sinusoidal embedding: sigma[N] -> t = argmin |log sigma - log_sigmas| -> [N][dim] fp16 (cos | sin)
// sigma_mod > 0: sigma holds sigma_mod samples and sample n reads sigma[n % sigma_mod] (CFG pair)
int timestep_embed_launch(const float* sigma, const float* log_sigmas, int n_sig, int N, int dim, half_t* out, float* t_out,
                          hipStream_t stream, int sigma_mod = 0);

// up to three ranges [base, base + bytes) copied to [base + bytes, base + 2 bytes) in one launch (the hand-over of a CFG pair's shared prefix, unet.hip)
struct DupArgs {
    char* base[3] = {nullptr, nullptr, nullptr};
    unsigned long long bytes[3] = {0, 0, 0};   // multiples of 16
    int count = 0;
};
int dup_halves_launch(const DupArgs& a, hipStream_t stream);

// weight repack (device side, run once at load)
int repack_conv3x3_launch(const void* src, int src_is_f32, int O, int I, half_t* dst, hipStream_t stream);  // OIHW -> [O][ky][kx][I]
int repack_rows_launch(const void* src, int src_is_f32, int rows, int cols, half_t* dst, int geglu_bn, hipStream_t stream);
int ctx_pad_launch(const void* src, int src_is_f32, int n, int T, int Tp, int D, half_t* dst, hipStream_t stream);
int fill_half_launch(half_t* dst, size_t n, float v, hipStream_t stream);

// sampler / guidance elementwise (fp32 latents)
int cfg_combine_launch(const float* den2, float* out, float cfg, size_t n_half, hipStream_t stream);   // out = u + (c-u)*cfg, den2=[u;c]
int axpby_launch(float* x, float a, const float* y, float b, const float* z, float c, size_t n, hipStream_t stream);  // x = a*x + b*y + c*z
// wrapper-hook guards: flags[0] = epoch when a != b (words_ab 32-bit words), flags[1] = epoch when the halves of x / sigma differ
int hook_check_launch(const void* a, const void* b, size_t words_ab, const void* x, size_t half_words_x, const void* sigma, int half_sigma,
                      int* flags, int epoch, hipStream_t stream);
// bislerp (LD.py:429-518): fp32 NCHW [n][c][h][w] -> [n][c][h_new][w_new]; tmp holds n*c*h*w_new floats (width pass first)
int bislerp_launch(const float* x, float* tmp, float* y, int n, int c, int h, int w, int h_new, int w_new, hipStream_t stream);
