// Single-operator entry points of the C ABI (include/ld_mi355x.h) — thin, argument-checked wrappers over the
// kernel launchers; used by the parity tests and by hosts that want to compose their own graphs.
#include "kernels.h"
#include "../../include/ld_mi355x.h"

namespace {
// scratch for ld_op_linear's GEGLU path: the op takes weights in checkpoint row order, the kernel wants them
// tile-interleaved, so the op repacks into caller-provided workspace
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
}  // namespace

extern "C" {

const char* ld_version(void) { return "ld_mi355x 0.1 (gfx950)"; }

const char* ld_status_string(int s) {
    switch (s) {
        case LD_OK: return "ok";
        case LD_ERR_ARG: return "invalid argument";
        case LD_ERR_SHAPE: return "unsupported shape or alignment";
        case LD_ERR_HIP: return "HIP runtime error";
        case LD_ERR_STATE: return "invalid call order (weights / reserve / context missing)";
        default: return "unknown status";
    }
}

int ld_op_linear(const void* x, const void* w, const void* bias, const void* residual, void* y, int M, int N, int K, float alpha,
                 int act, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GemmParams p;
    p.A = (const half_t*)x; p.lda = K;
    p.W = (const half_t*)w; p.ldw = K;
    p.M = M; p.N = N; p.K = K;
    p.alpha = alpha;
    p.bias_n = (const half_t*)bias;
    p.R = (const half_t*)residual;
    p.act = act;
    p.C = (half_t*)y;
    p.ldc = p.ldr = (act == 2 ? N / 2 : N);
    char* wsp = (char*)ws;
    if (act == 2) {   // repack [value | gate] rows into the tile-interleaved order the fused epilogue expects
        if (bias == nullptr || (N & 15)) return LD_ERR_ARG;
        const int bn = gemm_pick_bn(N);
        if (N % bn) return LD_ERR_SHAPE;
        const size_t wb = align256((size_t)N * K * sizeof(half_t)), bb = align256((size_t)N * sizeof(half_t));
        if (ws == nullptr || ws_bytes < wb + bb) return LD_ERR_ARG;
        half_t* w2 = (half_t*)wsp;
        half_t* b2 = (half_t*)(wsp + wb);
        int st = repack_rows_launch(w, 0, N, K, w2, bn, stream);
        if (st != LD_OK) return st;
        st = repack_rows_launch(bias, 0, N, 1, b2, bn, stream);
        if (st != LD_OK) return st;
        p.W = w2;
        p.bias_n = b2;
        p.bn = bn;
        wsp += wb + bb;
        ws_bytes -= wb + bb;
    }
    p.partial = (float*)wsp;
    p.partial_bytes = ws ? ws_bytes : 0;
    if (ws == nullptr) p.partial = nullptr;
    return gemm_launch(p, stream);
}

static int op_conv(const void* x1, int c1, const void* x2, int c2, int n, int h, int w, int hv, int wv, int stride, int ksize,
                   const void* wt, const void* bias, const void* rowvec, const void* residual, void* y, int cout, void* ws,
                   size_t ws_bytes, void* stream, float* gn_part, int* gn_chunks) {
    if (stride < 1 || (ksize != 1 && ksize != 3)) return LD_ERR_ARG;
    GemmParams p;
    if (gn_part != nullptr) {   // GroupNorm partial statistics of the output, where the kernel that runs this shape writes them (as the executors ask: unet.hip want_stats)
        const int HW = (ksize == 3 ? (hv - 1) / stride + 1 : hv) * (ksize == 3 ? (wv - 1) / stride + 1 : wv);
        p.gn_part = gn_part;
        p.gn_P = gn_num_chunks(n, HW);
        p.gn_HW = HW;
        p.gn_ppb = (HW + p.gn_P - 1) / p.gn_P;
        p.gn_part_done = gn_chunks;
    }
    p.conv = 1;
    p.ksize = ksize;
    p.A = (const half_t*)x1; p.A2 = (const half_t*)x2; p.C1 = c1; p.C2 = c2;
    p.Hs = h; p.Ws = w; p.Hv = hv; p.Wv = wv; p.stride = stride;
    p.Ho = ksize == 3 ? (hv - 1) / stride + 1 : hv;
    p.Wo = ksize == 3 ? (wv - 1) / stride + 1 : wv;
    p.K = ksize * ksize * (c1 + c2);
    p.W = (const half_t*)wt; p.ldw = p.K;
    p.M = n * p.Ho * p.Wo; p.N = cout;
    p.bias_n = (const half_t*)bias;
    p.rowvec = (const half_t*)rowvec; p.rows_per_vec = p.Ho * p.Wo; p.ldrv = cout;
    p.R = (const half_t*)residual; p.ldr = cout;
    p.C = (half_t*)y; p.ldc = cout;
    p.partial = (float*)ws;
    p.partial_bytes = ws ? ws_bytes : 0;
    constexpr size_t sync_b = LD_SYNC_INTS * sizeof(int);
    if (ws != nullptr && ksize == 3 && conv8_weight_eligible(cout, c1 + c2) && ws_bytes > sync_b + align256(conv8_weight_bytes(cout, c1 + c2)) + ((size_t)8 << 20)) {
        // row-resident kernel (conv8.hip): the head of a roomy scratch buffer holds the (zeroed) counters of its in-launch reduction and a
        // copy of the weights in its layout, made per call here (the UNet executor keeps that copy resident)
        GemmParams c8 = p;
        const size_t w8b = align256(conv8_weight_bytes(cout, c1 + c2));
        c8.sync = (int*)ws;
        c8.W8 = (const half_t*)((char*)ws + sync_b);
        c8.partial = (float*)((char*)ws + sync_b + w8b);
        c8.partial_bytes = ws_bytes - sync_b - w8b;
        if (conv8_plan(c8, nullptr)) {
            if (hipMemsetAsync(ws, 0, sync_b, (hipStream_t)stream) != hipSuccess) return LD_ERR_HIP;
            const int st = conv8_repack_launch((const half_t*)wt, cout, c1 + c2, (half_t*)((char*)ws + sync_b), (hipStream_t)stream);
            if (st != LD_OK) return st;
            return gemm_launch(c8, (hipStream_t)stream);
        }
    }
    return gemm_launch(p, (hipStream_t)stream);
}

int ld_op_conv(const void* x1, int c1, const void* x2, int c2, int n, int h, int w, int hv, int wv, int stride, int ksize,
               const void* wt, const void* bias, const void* rowvec, const void* residual, void* y, int cout, void* ws,
               size_t ws_bytes, void* stream) {
    return op_conv(x1, c1, x2, c2, n, h, w, hv, wv, stride, ksize, wt, bias, rowvec, residual, y, cout, ws, ws_bytes, stream, nullptr, nullptr);
}

size_t ld_op_conv_gn_partials_floats(int n, int hw) { return groupnorm_workspace_bytes(n, hw) / sizeof(float); }

int ld_op_conv_gn_partials(const void* x, int c, int n, int h, int w, int hv, int wv, const void* wt, const void* bias, const void* residual,
                           void* y, int cout, float* part, int* chunks, void* ws, size_t ws_bytes, void* stream) {
    if (part == nullptr || chunks == nullptr) return LD_ERR_ARG;
    *chunks = 0;
    return op_conv(x, c, nullptr, 0, n, h, w, hv, wv, 1, 3, wt, bias, nullptr, residual, y, cout, ws, ws_bytes, stream, part, chunks);
}

size_t ld_op_groupnorm_conv_ws_bytes(int c1, int c2, int n, int h, int w, int cout) {
    const size_t C = (size_t)c1 + c2, HW = (size_t)h * w;
    return align256(groupnorm_workspace_bytes(n, (int)HW)) + 2 * align256((size_t)n * C * sizeof(float)) + align256((size_t)n * HW * C * sizeof(half_t)) +
           align256(LD_SYNC_INTS * sizeof(int)) + (conv8_weight_eligible(cout, c1 + c2) ? align256(conv8_weight_bytes(cout, c1 + c2)) : 0) + ((size_t)96 << 20);
}

int ld_op_groupnorm_conv(const void* x1, int c1, const void* x2, int c2, int n, int h, int w, const void* gamma, const void* beta, float eps,
                         const void* wt, const void* bias, const void* rowvec, const void* residual, void* y, int cout, void* ws, size_t ws_bytes,
                         void* stream_) {
    // GroupNorm(32) + SiLU + 3x3 convolution (stride 1, pad 1): the reference's ResBlock1.in_layers / out_layers (LD.py:5224-5262).
    // On the halo-tile kernel with one N tile the normalisation is fused into the convolution's A operand; otherwise two-pass GroupNorm, then the conv.
    if (x1 == nullptr || gamma == nullptr || beta == nullptr || wt == nullptr || y == nullptr || ws == nullptr) return LD_ERR_ARG;
    if (ws_bytes < ld_op_groupnorm_conv_ws_bytes(c1, c2, n, h, w, cout)) return LD_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int C = c1 + c2, HW = h * w;
    char* q = (char*)ws;
    float* part = (float*)q; q += align256(groupnorm_workspace_bytes(n, HW));
    float* scale = (float*)q; q += align256((size_t)n * C * sizeof(float));
    float* shift = (float*)q; q += align256((size_t)n * C * sizeof(float));
    half_t* g = (half_t*)q; q += align256((size_t)n * HW * C * sizeof(half_t));
    int* sync = (int*)q; q += align256(LD_SYNC_INTS * sizeof(int));
    half_t* w8 = nullptr;
    if (conv8_weight_eligible(cout, C)) {
        w8 = (half_t*)q;
        q += align256(conv8_weight_bytes(cout, C));
    }
    GemmParams p;
    p.conv = 1; p.ksize = 3;
    p.A = (const half_t*)x1; p.A2 = (const half_t*)x2; p.C1 = c1; p.C2 = c2;
    p.Hs = p.Hv = p.Ho = h; p.Ws = p.Wv = p.Wo = w; p.stride = 1;
    p.K = 9 * C; p.W = (const half_t*)wt; p.ldw = p.K;
    p.M = n * HW; p.N = cout;
    p.bias_n = (const half_t*)bias;
    p.rowvec = (const half_t*)rowvec; p.rows_per_vec = HW; p.ldrv = cout;
    p.R = (const half_t*)residual; p.ldr = cout;
    p.C = (half_t*)y; p.ldc = cout;
    p.partial = (float*)q; p.partial_bytes = (size_t)96 << 20;
    if (gemm_conv_fuses_groupnorm(p)) {
        int st = groupnorm_scale_shift_launch((const half_t*)x1, c1, (const half_t*)x2, c2, n, HW, (const half_t*)gamma, (const half_t*)beta, eps, part,
                                              scale, shift, stream);
        if (st != LD_OK) return st;
        p.gn_scale = scale; p.gn_shift = shift; p.gn_silu = 1;
        return gemm_launch(p, stream);
    }
    int st = groupnorm_launch((const half_t*)x1, c1, (const half_t*)x2, c2, n, HW, (const half_t*)gamma, (const half_t*)beta, eps, 1, g, part, stream);
    if (st != LD_OK) return st;
    p.A = g; p.A2 = nullptr; p.C1 = C; p.C2 = 0;
    p.sync = sync; p.W8 = w8;
    if (conv8_plan(p, nullptr)) {   // row-resident kernel (conv8.hip) on the normalised tensor: zeroed counters and its own weight layout
        if (hipMemsetAsync(sync, 0, LD_SYNC_INTS * sizeof(int), stream) != hipSuccess) return LD_ERR_HIP;   // (a caller's scratch: not known to be zero)
        st = conv8_repack_launch((const half_t*)wt, cout, C, w8, stream);   // (per call here; the UNet executor keeps the copy resident)
        if (st != LD_OK) return st;
    }
    return gemm_launch(p, stream);
}

size_t ld_op_conv_skip_ws_bytes(int c, int sc1, int sc2, int cout) {
    return align256((size_t)cout * (9 * (size_t)c + sc1 + sc2) * sizeof(half_t)) + align256((size_t)cout * sizeof(half_t)) + ((size_t)96 << 20);
}

int ld_op_conv_skip(const void* x, int c, int n, int h, int w, const void* wt, const void* bias, const void* s1, int sc1, const void* s2, int sc2,
                    const void* wskip, const void* bskip, const void* rowvec, void* y, int cout, void* ws, size_t ws_bytes, void* stream_) {
    if (x == nullptr || wt == nullptr || bias == nullptr || s1 == nullptr || wskip == nullptr || bskip == nullptr || y == nullptr || ws == nullptr) return LD_ERR_ARG;
    if (c <= 0 || sc1 <= 0 || sc2 < 0 || (sc2 > 0 && s2 == nullptr) || cout <= 0 || n <= 0 || h <= 0 || w <= 0) return LD_ERR_ARG;
    if (ws_bytes < ld_op_conv_skip_ws_bytes(c, sc1, sc2, cout)) return LD_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int K9 = 9 * c, SC = sc1 + sc2;
    char* q = (char*)ws;
    half_t* wf = (half_t*)q; q += align256((size_t)cout * (K9 + SC) * sizeof(half_t));
    half_t* bf = (half_t*)q; q += align256((size_t)cout * sizeof(half_t));
    int st = skip_fold_launch((const half_t*)wt, (const half_t*)wskip, (const half_t*)bias, (const half_t*)bskip, cout, K9, SC, wf, bf, stream);
    if (st != LD_OK) return st;
    GemmParams p;
    p.conv = 1; p.ksize = 3;
    p.A = (const half_t*)x; p.C1 = c;
    p.Hs = p.Hv = p.Ho = h; p.Ws = p.Wv = p.Wo = w; p.stride = 1;
    p.S1 = (const half_t*)s1; p.SC1 = sc1; p.S2 = (const half_t*)s2; p.SC2 = sc2;
    p.K = K9 + SC; p.W = wf; p.ldw = p.K;
    p.M = n * h * w; p.N = cout;
    p.bias_n = bf;
    p.rowvec = (const half_t*)rowvec; p.rows_per_vec = h * w; p.ldrv = cout;
    p.C = (half_t*)y; p.ldc = cout;
    p.partial = (float*)q; p.partial_bytes = (size_t)96 << 20;
    return gemm_launch(p, stream);
}

int ld_op_repack_conv(const void* src, int dtype, int cout, int cin, void* dst, void* stream) {
    return repack_conv3x3_launch(src, dtype == LD_F32, cout, cin, (half_t*)dst, (hipStream_t)stream);
}

size_t ld_op_groupnorm_ws_bytes(int n, int hw) { return groupnorm_workspace_bytes(n, hw); }

int ld_op_groupnorm(const void* x1, int c1, const void* x2, int c2, int n, int hw, const void* gamma, const void* beta, float eps,
                    int silu, void* y, void* ws, void* stream) {
    return groupnorm_launch((const half_t*)x1, c1, (const half_t*)x2, c2, n, hw, (const half_t*)gamma, (const half_t*)beta, eps,
                            silu, (half_t*)y, (float*)ws, (hipStream_t)stream);
}

int ld_op_layernorm(const void* x, const void* gamma, const void* beta, void* y, int rows, int c, float eps, void* stream) {
    return layernorm_launch((const half_t*)x, (const half_t*)gamma, (const half_t*)beta, (half_t*)y, rows, c, eps, (hipStream_t)stream);
}

int ld_op_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo, int b, int heads,
                    int lq, int lk, int d, float scale, int causal, void* stream) {
    AttnParams a;
    a.Q = (const half_t*)q; a.ldq = ldq; a.sQ = (long long)lq * ldq;
    a.K = (const half_t*)k; a.ldk = ldk; a.sK = (long long)lk * ldk;
    a.Vt = (const half_t*)vt; a.ldvt = ldvt; a.sV = (long long)heads * d * ldvt;
    a.O = (half_t*)o; a.ldo = ldo; a.sO = (long long)lq * ldo;
    a.B = b; a.H = heads; a.Lq = lq; a.Lk = lk; a.d = d; a.scale = scale; a.causal = causal;
    return attention_launch(a, (hipStream_t)stream);
}

int ld_op_attention_rowv(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, int b, int heads,
                         int lq, int lk, int d, float scale, int causal, void* stream) {
    // q, k, v as column blocks of one fused [b][l][ldq] tensor (ldq == ldk == ldv with overlapping row ranges) share ONE batch stride,
    // which the per-operand strides below only reproduce when lq == lk
    const char *qb = (const char*)q, *kb = (const char*)k, *vb = (const char*)v;
    const long long rowb = (long long)ldq * 2;
    const bool fused = ldq == ldk && ldk == ldv && rowb > 0 && (kb - qb >= 0 ? kb - qb : qb - kb) < rowb && (vb - qb >= 0 ? vb - qb : qb - vb) < rowb;
    if (fused && lq != lk) return LD_ERR_SHAPE;
    AttnParams a;
    a.Q = (const half_t*)q; a.ldq = ldq; a.sQ = (long long)lq * ldq;
    a.K = (const half_t*)k; a.ldk = ldk; a.sK = (long long)lk * ldk;
    a.V = (const half_t*)v; a.ldv = ldv; a.sV = (long long)lk * ldv;
    a.O = (half_t*)o; a.ldo = ldo; a.sO = (long long)lq * ldo;
    a.B = b; a.H = heads; a.Lq = lq; a.Lk = lk; a.d = d; a.scale = scale; a.causal = causal;
    return attention_launch(a, (hipStream_t)stream);
}

int ld_op_softmax_rows(void* s, int rows, int cols, void* stream) {
    return softmax_rows_launch((half_t*)s, rows, cols, cols, (hipStream_t)stream);
}

int ld_op_timestep_embed(const float* sigma, const float* log_sigmas, int n_sigmas, int n, int dim, void* out, float* t_out, void* stream) {
    return timestep_embed_launch(sigma, log_sigmas, n_sigmas, n, dim, (half_t*)out, t_out, (hipStream_t)stream);
}

int ld_op_cfg_combine(const float* den2, float* out, float cfg, size_t n_half, void* stream) {
    return cfg_combine_launch(den2, out, cfg, n_half, (hipStream_t)stream);
}

int ld_op_hook_check(const void* a, const void* b, size_t words_ab, const void* x, size_t half_words_x, const void* sigma, int half_sigma,
                     int* flags, int epoch, void* stream) {
    return hook_check_launch(a, b, words_ab, x, half_words_x, sigma, half_sigma, flags, epoch, (hipStream_t)stream);
}

int ld_op_axpby(float* x, float a, const float* y, float b, const float* z, float c, size_t n, void* stream) {
    return axpby_launch(x, a, y, b, z, c, n, (hipStream_t)stream);
}

int ld_op_linear_ln(const void* x, const void* w_prod, const void* b_prod, const void* gamma, const void* beta, const void* w,
                    const void* bias, void* t_out, void* y, int M, int C, int N, float eps, void* ws, size_t ws_bytes, void* stream_) {
    // the UNet's LayerNorm fold (unet.hip, gemm.h) as a stand-alone operator pair, for parity tests:
    //   t = x · w_prod^T + b_prod         (producer: also emits per-row (sum, sum of squares) partials of the fp16 t)
    //   y = LayerNorm(t; gamma, beta, eps) · w^T + bias   computed as rstd * (t · W'^T - mu * wsum) + b' on the accumulators
    if (x == nullptr || w_prod == nullptr || gamma == nullptr || beta == nullptr || w == nullptr || t_out == nullptr || y == nullptr ||
        ws == nullptr)
        return LD_ERR_ARG;
    if (!gemm_ln_fold_available()) return LD_ERR_STATE;
    hipStream_t stream = (hipStream_t)stream_;
    const size_t wb = align256((size_t)N * C * sizeof(half_t)), bb = align256((size_t)N * sizeof(half_t)), sb = align256((size_t)N * sizeof(float));
    const size_t stb = align256((size_t)((C + 63) / 64) * M * 2 * sizeof(float));
    if (ws_bytes < wb + bb + sb + stb) return LD_ERR_ARG;
    char* wsp = (char*)ws;
    half_t* w2 = (half_t*)wsp;
    half_t* b2 = (half_t*)(wsp + wb);
    float* wsum = (float*)(wsp + wb + bb);
    float* stat = (float*)(wsp + wb + bb + sb);
    int st = ln_fold_launch((const half_t*)w, N, C, (const half_t*)gamma, (const half_t*)beta, (const half_t*)bias, w2, b2, wsum, stream);
    if (st != LD_OK) return st;
    int parts = 0;
    GemmParams a;
    a.A = (const half_t*)x; a.lda = C;
    a.W = (const half_t*)w_prod; a.ldw = C;
    a.M = M; a.N = C; a.K = C;
    a.bias_n = (const half_t*)b_prod;
    a.C = (half_t*)t_out; a.ldc = C;
    a.stat_out = stat; a.stat_parts_out = &parts;
    st = gemm_launch(a, stream);
    if (st != LD_OK) return st;
    GemmParams b;
    b.A = (const half_t*)t_out; b.lda = C;
    b.W = w2; b.ldw = C;
    b.M = M; b.N = N; b.K = C;
    b.bias_n = b2;
    b.C = (half_t*)y; b.ldc = N;
    b.ln_stat = stat; b.ln_parts = parts; b.ln_rows = M;
    b.ln_inv_c = 1.0f / (float)C; b.ln_eps = eps;
    b.ln_wsum = wsum;
    return gemm_launch(b, stream);
}

int ld_op_linear_ln_geglu(const void* x, const void* w_prod, const void* b_prod, const void* gamma, const void* beta, const void* w,
                          const void* bias, void* t_out, void* y, int M, int C, int N, float eps, void* ws, size_t ws_bytes, void* stream_) {
    // the transformer block's MLP input as the executor runs it (unet.hip): t = x · w_prod^T + b_prod with row statistics, then
    //   y[M][N/2] = a * gelu(g),  [a | g] = LayerNorm(t) · w^T + bias     (GEGLU, LD.py:4513-4515, on the LayerNorm-folded weights)
    // w rows are repacked into the tile-interleaved [value | gate] order first, then folded (the fold is row-wise).
    if (x == nullptr || w_prod == nullptr || gamma == nullptr || beta == nullptr || w == nullptr || bias == nullptr || t_out == nullptr ||
        y == nullptr || ws == nullptr)
        return LD_ERR_ARG;
    if (!gemm_ln_fold_available()) return LD_ERR_STATE;
    hipStream_t stream = (hipStream_t)stream_;
    const int bn = gemm_pick_bn(N);
    if ((N & 15) || (N % bn)) return LD_ERR_SHAPE;
    const size_t wb = align256((size_t)N * C * sizeof(half_t)), bb = align256((size_t)N * sizeof(half_t)), sb = align256((size_t)N * sizeof(float));
    const size_t stb = align256((size_t)((C + 63) / 64) * M * 2 * sizeof(float));
    if (ws_bytes < 2 * wb + 2 * bb + sb + stb) return LD_ERR_ARG;
    char* wsp = (char*)ws;
    half_t* wr = (half_t*)wsp;                         // repacked rows
    half_t* br = (half_t*)(wsp + wb);
    half_t* w2 = (half_t*)(wsp + wb + bb);             // folded
    half_t* b2 = (half_t*)(wsp + 2 * wb + bb);
    float* wsum = (float*)(wsp + 2 * wb + 2 * bb);
    float* stat = (float*)(wsp + 2 * wb + 2 * bb + sb);
    int st = repack_rows_launch(w, 0, N, C, wr, bn, stream);
    if (st == LD_OK) st = repack_rows_launch(bias, 0, N, 1, br, bn, stream);
    if (st == LD_OK) st = ln_fold_launch(wr, N, C, (const half_t*)gamma, (const half_t*)beta, br, w2, b2, wsum, stream);
    if (st != LD_OK) return st;
    int parts = 0;
    GemmParams a;
    a.A = (const half_t*)x; a.lda = C;
    a.W = (const half_t*)w_prod; a.ldw = C;
    a.M = M; a.N = C; a.K = C;
    a.bias_n = (const half_t*)b_prod;
    a.C = (half_t*)t_out; a.ldc = C;
    a.stat_out = stat; a.stat_parts_out = &parts;
    st = gemm_launch(a, stream);
    if (st != LD_OK) return st;
    GemmParams b;
    b.A = (const half_t*)t_out; b.lda = C;
    b.W = w2; b.ldw = C;
    b.M = M; b.N = N; b.K = C;
    b.bias_n = b2;
    b.act = 2; b.bn = bn;
    b.C = (half_t*)y; b.ldc = N / 2; b.ldr = N / 2;
    b.ln_stat = stat; b.ln_parts = parts; b.ln_rows = M;
    b.ln_inv_c = 1.0f / (float)C; b.ln_eps = eps;
    b.ln_wsum = wsum;
    return gemm_launch(b, stream);
}

int ld_op_bislerp(const float* x, float* tmp, float* y, int n, int c, int h, int w, int h_new, int w_new, void* stream) {
    return bislerp_launch(x, tmp, y, n, c, h, w, h_new, w_new, (hipStream_t)stream);
}

}  // extern "C"
