// KL-VAE decoder executor: post_quant_conv + Decoder.forward (LD.py:3470-3473, 3857-3882) + VAE.decode's clamp and
// NHWC output (LD.py:6357-6381), on NHWC fp16 activations with the same kernels as the UNet.
// The single-head mid-block attention (AttnBlock, LD.py:3605-3642) runs as one fused [q | k | v] projection + flash_attn512_kernel
// (attention.hip: the 512 — or 256 — channels split over a pair of waves) since round 6; other widths keep the three MFMA GEMMs around a
// row softmax (S = alpha·Q·K^T, P = softmax(S), O = P·V with V^T produced directly by a swapped projection GEMM).
#include <cstring>

#include "runtime.h"
#include "../../include/ld_mi355x.h"

namespace {
struct VResW {
    int cin, cout;
    int n1_g, n1_b, c1_w, c1_b, n2_g, n2_b, c2_w, c2_b, nin_w = -1, nin_b = -1;
};
struct VAttnW {
    int c = 0;
    int n_g, n_b, q_w, q_b, k_w, k_b, v_w, v_b, o_w, o_b;
};
}  // namespace

struct ld_vae {
    ld_vae_config cfg;
    ParamTable pt;
    int pq_w, pq_b, cin_w, cin_b;
    VResW mid1, mid2;
    VAttnW mid_attn;
    std::vector<std::vector<VResW>> up;   // [level][block]
    // encoder (Encoder, LD.py:3649-3758 + quant_conv 3468), present when cfg.with_encoder
    int e_cin_w = -1, e_cin_b = -1, e_no_g = -1, e_no_b = -1, e_co_w = -1, e_co_b = -1, e_q_w = -1, e_q_b = -1;
    std::vector<std::vector<VResW>> down;
    std::vector<int> dwn_w, dwn_b;
    VResW e_mid1, e_mid2;
    VAttnW e_attn;
    std::vector<int> ups_w, ups_b;        // [level] (-1 at level 0)
    int no_g, no_b, co_w, co_b;
    int block_in0 = 0;
    Arena arena;
    char* ws_base = nullptr;
    size_t ws_bytes = 0;
    float* splitk_ws = nullptr;
    size_t splitk_bytes = 0;
    int plan_b = 0, plan_h = 0, plan_w = 0;
    int last_launches = 0;
    double last_flops = 0.0;
    Timing timing;
    bool want_timing = false;
    half_t* co_pad = nullptr;             // conv_out weights zero-padded to 32 output rows + 32 biases (MFMA output conv)
};

namespace {

VResW add_vres(ld_vae* v, const std::string& p, int cin, int cout) {
    ParamTable& t = v->pt;
    VResW r;
    r.cin = cin;
    r.cout = cout;
    r.n1_g = t.add(p + ".norm1.weight", PK_VEC, {cin});
    r.n1_b = t.add(p + ".norm1.bias", PK_VEC, {cin});
    r.c1_w = t.add(p + ".conv1.weight", PK_CONV3, {cout, cin, 3, 3});
    r.c1_b = t.add(p + ".conv1.bias", PK_VEC, {cout});
    r.n2_g = t.add(p + ".norm2.weight", PK_VEC, {cout});
    r.n2_b = t.add(p + ".norm2.bias", PK_VEC, {cout});
    r.c2_w = t.add(p + ".conv2.weight", PK_CONV3, {cout, cout, 3, 3});
    r.c2_b = t.add(p + ".conv2.bias", PK_VEC, {cout});
    if (cin != cout) {
        r.nin_w = t.add(p + ".nin_shortcut.weight", PK_MAT, {cout, cin, 1, 1});
        r.nin_b = t.add(p + ".nin_shortcut.bias", PK_VEC, {cout});
    }
    return r;
}

VAttnW add_vattn(ld_vae* v, const std::string& p, int c) {
    ParamTable& t = v->pt;
    VAttnW a;
    a.c = c;
    a.n_g = t.add(p + ".norm.weight", PK_VEC, {c});
    a.n_b = t.add(p + ".norm.bias", PK_VEC, {c});
    a.q_w = t.add(p + ".q.weight", PK_MAT, {c, c, 1, 1});
    a.k_w = t.add(p + ".k.weight", PK_MAT, {c, c, 1, 1}, 16);   // [q;k;v] as one [3C][C] projection
    a.v_w = t.add(p + ".v.weight", PK_MAT, {c, c, 1, 1}, 16);
    a.q_b = t.add(p + ".q.bias", PK_VEC, {c});
    a.k_b = t.add(p + ".k.bias", PK_VEC, {c}, 16);
    a.v_b = t.add(p + ".v.bias", PK_VEC, {c}, 16);
    a.o_w = t.add(p + ".proj_out.weight", PK_MAT, {c, c, 1, 1});
    a.o_b = t.add(p + ".proj_out.bias", PK_VEC, {c});
    return a;
}

int build(ld_vae* v) {
    const ld_vae_config& c = v->cfg;
    if (c.num_levels < 1 || c.num_levels > 8 || c.z_channels < 1 || c.z_channels > 4 || c.out_ch < 1 || c.out_ch > 4) return LD_ERR_ARG;
    ParamTable& t = v->pt;
    const int z = c.z_channels;
    v->pq_w = t.add("post_quant_conv.weight", PK_MAT, {z, z, 1, 1});
    v->pq_b = t.add("post_quant_conv.bias", PK_VEC, {z});
    int bi = c.ch * c.ch_mult[c.num_levels - 1];
    if (bi % 64) return LD_ERR_SHAPE;
    v->block_in0 = bi;
    v->cin_w = t.add("decoder.conv_in.weight", PK_CONV3, {bi, z, 3, 3});
    v->cin_b = t.add("decoder.conv_in.bias", PK_VEC, {bi});
    v->mid1 = add_vres(v, "decoder.mid.block_1", bi, bi);
    v->mid_attn = add_vattn(v, "decoder.mid.attn_1", bi);
    v->mid2 = add_vres(v, "decoder.mid.block_2", bi, bi);
    v->up.assign(c.num_levels, {});
    v->ups_w.assign(c.num_levels, -1);
    v->ups_b.assign(c.num_levels, -1);
    char buf[96];
    for (int lvl = c.num_levels - 1; lvl >= 0; --lvl) {
        const int bo = c.ch * c.ch_mult[lvl];
        if (bo % 64) return LD_ERR_SHAPE;
        for (int b = 0; b <= c.num_res_blocks; ++b) {
            snprintf(buf, sizeof buf, "decoder.up.%d.block.%d", lvl, b);
            v->up[lvl].push_back(add_vres(v, buf, bi, bo));
            bi = bo;
        }
        if (lvl != 0) {
            snprintf(buf, sizeof buf, "decoder.up.%d.upsample.conv", lvl);
            v->ups_w[lvl] = t.add(std::string(buf) + ".weight", PK_CONV3, {bi, bi, 3, 3});
            v->ups_b[lvl] = t.add(std::string(buf) + ".bias", PK_VEC, {bi});
        }
    }
    v->no_g = t.add("decoder.norm_out.weight", PK_VEC, {bi});
    v->no_b = t.add("decoder.norm_out.bias", PK_VEC, {bi});
    v->co_w = t.add("decoder.conv_out.weight", PK_CONV3, {c.out_ch, bi, 3, 3});
    v->co_b = t.add("decoder.conv_out.bias", PK_VEC, {c.out_ch});
    if (c.with_encoder) {
        if (2 * z != 8 || c.out_ch > 4) return LD_ERR_SHAPE;          // moments are 8 channels (double_z of z_channels 4)
        v->e_cin_w = t.add("encoder.conv_in.weight", PK_CONV3, {c.ch, c.out_ch, 3, 3});
        v->e_cin_b = t.add("encoder.conv_in.bias", PK_VEC, {c.ch});
        v->down.assign(c.num_levels, {});
        v->dwn_w.assign(c.num_levels, -1);
        v->dwn_b.assign(c.num_levels, -1);
        int ei = c.ch;
        for (int lvl = 0; lvl < c.num_levels; ++lvl) {
            const int eo = c.ch * c.ch_mult[lvl];
            for (int b = 0; b < c.num_res_blocks; ++b) {
                snprintf(buf, sizeof buf, "encoder.down.%d.block.%d", lvl, b);
                v->down[lvl].push_back(add_vres(v, buf, ei, eo));
                ei = eo;
            }
            if (lvl != c.num_levels - 1) {
                snprintf(buf, sizeof buf, "encoder.down.%d.downsample.conv", lvl);
                v->dwn_w[lvl] = t.add(std::string(buf) + ".weight", PK_CONV3, {ei, ei, 3, 3});
                v->dwn_b[lvl] = t.add(std::string(buf) + ".bias", PK_VEC, {ei});
            }
        }
        v->e_mid1 = add_vres(v, "encoder.mid.block_1", ei, ei);
        v->e_attn = add_vattn(v, "encoder.mid.attn_1", ei);
        v->e_mid2 = add_vres(v, "encoder.mid.block_2", ei, ei);
        v->e_no_g = t.add("encoder.norm_out.weight", PK_VEC, {ei});
        v->e_no_b = t.add("encoder.norm_out.bias", PK_VEC, {ei});
        v->e_co_w = t.add("encoder.conv_out.weight", PK_CONV3, {2 * z, ei, 3, 3});
        v->e_co_b = t.add("encoder.conv_out.bias", PK_VEC, {2 * z});
        v->e_q_w = t.add("quant_conv.weight", PK_MAT, {2 * z, 2 * z, 1, 1});
        v->e_q_b = t.add("quant_conv.bias", PK_VEC, {2 * z});
    }
    return t.finalize();
}

struct VRun {
    ld_vae* v;
    Exec ex;
    int n;
    half_t* P(int s) const { return v->pt.ptr(s); }

    // GroupNorm partial statistics of the tensor a convolution just wrote (its epilogue sums what it stores: gemm.h gn_part), for the
    // GroupNorm that reads that tensor next: the statistics pass over it is skipped
    const half_t* st_of = nullptr;
    float* st_buf = nullptr;
    int st_P = 0;
    void gn(const half_t* x, int C, int HW, int gslot, int bslot, int silu, half_t* y) {
        const bool ready = x == st_of && st_P > 0;
        ex.groupnorm(x, C, nullptr, 0, n, HW, P(gslot), P(bslot), 1e-6f, silu, y, ready ? st_buf : nullptr, ready ? st_P : 0);
    }

    void conv(const half_t* x, int cin, int Hs, int Ws, int Hv, int Wv, int ksize, int wslot, int bslot, int cout, const half_t* R, half_t* out,
              int stride = 1, int pad = -1, int Ho = 0, int Wo = 0, float* stats = nullptr) {
        GemmParams p;
        p.conv = 1;
        p.ksize = ksize;
        p.pad = pad;
        p.A = x; p.C1 = cin;
        p.Hs = Hs; p.Ws = Ws; p.Hv = Hv; p.Wv = Wv; p.Ho = Ho ? Ho : Hv; p.Wo = Wo ? Wo : Wv; p.stride = stride;
        p.W = P(wslot); p.ldw = ksize * ksize * cin;
        p.M = n * p.Ho * p.Wo; p.N = cout; p.K = ksize * ksize * cin;
        p.bias_n = P(bslot);
        p.R = R; p.ldr = cout;
        p.C = out; p.ldc = cout;
        int done = 0;
        if (stats != nullptr) {
            p.gn_part = stats;
            p.gn_part_done = &done;
        }
        ex.gemm(p);
        if (stats != nullptr) {
            st_of = out;
            st_buf = stats;
            st_P = done;
        } else if (out == st_of) {
            st_of = nullptr;      // (the tensor the statistics described was overwritten)
        }
    }

    // GroupNorm (eps 1e-6) + swish + 3x3 convolution (norm1 -> conv1, norm2 -> conv2 of a ResnetBlock): where the convolution's output is one
    // halo tile wide (N = 256 at 256-pixel rows, N = 128 at 512-pixel rows) the normalisation is applied inside its halo loader and only the
    // statistics are finalised here (Exec::gn_silu_conv); otherwise the two-pass GroupNorm into `g`, then the convolution, as before.
    void gn_conv(const half_t* x, int cin, int H, int W, int gslot, int bslot, half_t* g, int wslot, int cbslot, int cout, const half_t* R, half_t* out,
                 float* stats) {
        GemmParams p;
        p.conv = 1;
        p.ksize = 3;
        p.pad = -1;
        p.A = x; p.C1 = cin;
        p.Hs = p.Hv = p.Ho = H; p.Ws = p.Wv = p.Wo = W; p.stride = 1;
        p.W = P(wslot); p.ldw = 9 * cin;
        p.M = n * H * W; p.N = cout; p.K = 9 * cin;
        p.bias_n = P(cbslot);
        p.R = R; p.ldr = cout;
        p.C = out; p.ldc = cout;
        int done = 0;
        if (stats != nullptr) {
            p.gn_part = stats;
            p.gn_part_done = &done;
        }
        const bool ready = x == st_of && st_P > 0;
        ex.gn_silu_conv(p, n, H * W, P(gslot), P(bslot), 1e-6f, g, ready ? st_buf : nullptr, ready ? st_P : 0);
        if (stats != nullptr) {
            st_of = out;
            st_buf = stats;
            st_P = done;
        } else if (out == st_of) {
            st_of = nullptr;
        }
    }

    // ResnetBlock.forward, LD.py:3560-3576 (GroupNorm eps 1e-6, swish)
    half_t* resblock(const VResW& r, const half_t* x, int H, int W) {
        Arena& ar = *ex.arena;
        const size_t M = (size_t)n * H * W;
        half_t* out = ar.halfs(M * r.cout);
        float* so = reinterpret_cast<float*>(ar.alloc(groupnorm_workspace_bytes(n, H * W)));   // statistics of `out` (for the next GroupNorm)
        const size_t mk = ar.mark();
        half_t* g1 = ar.halfs(M * r.cin);
        half_t* h1 = ar.halfs(M * r.cout);
        float* s1 = reinterpret_cast<float*>(ar.alloc(groupnorm_workspace_bytes(n, H * W)));
        gn_conv(x, r.cin, H, W, r.n1_g, r.n1_b, g1, r.c1_w, r.c1_b, r.cout, nullptr, h1, s1);
        half_t* g2 = r.cout <= r.cin ? g1 : ar.halfs(M * r.cout);
        const half_t* skip = x;
        if (r.nin_w >= 0) {
            // (h1 stays live until conv2 has read it — the fused GroupNorm reads the raw tensor — so the shortcut gets its own buffer)
            half_t* sk = ar.halfs(M * r.cout);
            conv(x, r.cin, H, W, H, W, 1, r.nin_w, r.nin_b, r.cout, nullptr, sk);   // (writes no statistics; h1's stay valid)
            skip = sk;
        }
        gn_conv(h1, r.cout, H, W, r.n2_g, r.n2_b, g2, r.c2_w, r.c2_b, r.cout, skip, out, so);
        ar.release(mk);
        return out;
    }

    // AttnBlock.forward, LD.py:3630-3642 (one head of C channels, pytorch_attention LD.py:3591-3602)
    half_t* attn(const VAttnW& aw, const half_t* x, int H, int W) {
        Arena& ar = *ex.arena;
        const int C = aw.c, L = H * W;
        const size_t M = (size_t)n * L;
        half_t* out = ar.halfs(M * C);
        const size_t mk = ar.mark();
        half_t* g = ar.halfs(M * C);
        ex.groupnorm(x, C, nullptr, 0, n, L, P(aw.n_g), P(aw.n_b), 1e-6f, 0, g);
        half_t* o = g;   // the attention output reuses the normalised tensor's buffer
        if (C == 512 || C == 256) {
            // round 6: [q | k | v] as ONE projection, then flash attention over the single head (flash_attn512_kernel: V row-major, the
            // score matrix never exists — it was 2.1 GB of fp16 at 1024^2, b = 4 — and the V^T GEMM, the softmax pass and a GEMM are gone)
            half_t* qkv = ar.halfs(M * 3 * C);
            {
                GemmParams p;
                p.A = g; p.lda = C; p.W = P(aw.q_w); p.ldw = C;
                p.M = (int)M; p.N = 3 * C; p.K = C; p.bias_n = P(aw.q_b);
                p.C = qkv; p.ldc = 3 * C;
                ex.gemm(p);
            }
            AttnParams a;
            a.Q = qkv; a.K = qkv + C; a.V = qkv + 2 * C;
            a.ldq = a.ldk = a.ldv = 3 * C;
            a.sQ = a.sK = a.sV = (long long)L * 3 * C;
            a.O = o; a.ldo = C; a.sO = (long long)L * C;
            a.B = n; a.H = 1; a.Lq = a.Lk = L; a.d = C;
            a.scale = 1.0f / sqrtf((float)C);
            ex.attention(a);
        } else {
            // any other width: three GEMMs around a row softmax over the materialised L x L scores
            half_t* qk = ar.halfs(M * 2 * C);
            {
                GemmParams p;
                p.A = g; p.lda = C; p.W = P(aw.q_w); p.ldw = C;
                p.M = (int)M; p.N = 2 * C; p.K = C; p.bias_n = P(aw.q_b);
                p.C = qk; p.ldc = 2 * C;
                ex.gemm(p);
            }
            // key axis padded to a multiple of 8 (16-byte rows): pad keys get zero scores, zero probabilities and a finite V^T
            // column (bias only), so any latent size works (63x63, 65x65 ...), as in the reference
            const int Lp = (L + 7) & ~7;
            half_t* vt = ar.halfs((size_t)n * C * Lp);
            {   // V^T[b] = Wv · g_b^T + bv (bias along rows)
                GemmParams p;
                p.A = P(aw.v_w); p.lda = C; p.sA = 0;
                p.W = g; p.ldw = C; p.sW = (long long)L * C;
                p.M = C; p.N = Lp; p.n_valid = L; p.K = C; p.batch = n;
                p.bias_m = P(aw.v_b);
                p.C = vt; p.ldc = Lp; p.sC = (long long)C * Lp;
                ex.gemm(p);
            }
            half_t* s = ar.halfs((size_t)n * L * Lp);
            {   // S_b = Q_b K_b^T / sqrt(C)
                GemmParams p;
                p.A = qk; p.lda = 2 * C; p.sA = (long long)L * 2 * C;
                p.W = qk + C; p.ldw = 2 * C; p.sW = (long long)L * 2 * C;
                p.M = L; p.N = Lp; p.n_valid = L; p.K = C; p.batch = n;
                p.alpha = 1.0f / sqrtf((float)C);
                p.C = s; p.ldc = Lp; p.sC = (long long)L * Lp;
                ex.gemm(p);
            }
            ex.launches += 1;
            ex.t_begin(KC_MISC, 0.0, 1, "softmax", (long long)n * L, Lp, 0, 1);
            if (!ex.dry && ex.status == LD_OK) ex.note(softmax_rows_launch(s, n * L, Lp, Lp, ex.stream, L));
            ex.t_end("softmax_rows_kernel");
            {   // O_b = P_b V_b  (W operand = V^T [C][Lp]; the pad keys carry zero probability)
                GemmParams p;
                p.A = s; p.lda = Lp; p.sA = (long long)L * Lp;
                p.W = vt; p.ldw = Lp; p.sW = (long long)C * Lp;
                p.M = L; p.N = C; p.K = Lp; p.batch = n;
                p.C = o; p.ldc = C; p.sC = (long long)L * C;
                ex.gemm(p);
            }
        }
        {
            GemmParams p;
            p.A = o; p.lda = C; p.W = P(aw.o_w); p.ldw = C;
            p.M = (int)M; p.N = C; p.K = C; p.bias_n = P(aw.o_b);
            p.R = x; p.ldr = C;
            p.C = out; p.ldc = C;
            ex.gemm(p);
        }
        ar.release(mk);
        return out;
    }
};

int run_decode(ld_vae* v, bool dry, const float* z, float* out, int b, int h, int w, hipStream_t stream, size_t* dry_peak = nullptr) {
    VRun R;
    R.v = v;
    R.n = b;
    Exec& ex = R.ex;
    ex.stream = stream;
    ex.dry = dry;
    Arena plan;
    ex.arena = dry ? &plan : &v->arena;
    ex.splitk_ws = v->splitk_ws;
    ex.splitk_bytes = v->splitk_bytes;
    if (v->want_timing && !dry) {
        v->timing.reset();
        ex.timing = &v->timing;
    }
    Arena& ar = *ex.arena;
    ar.release(0);
    const ld_vae_config& c = v->cfg;
    int H = h, W = w;
    int C = v->block_in0;
    half_t* f = ar.halfs((size_t)b * H * W * C);
    {
        SmallConvInArgs a;
        a.x = z; a.pre_w = v->pt.ptr(v->pq_w); a.pre_b = v->pt.ptr(v->pq_b);
        a.w = v->pt.ptr(v->cin_w); a.b = v->pt.ptr(v->cin_b); a.y = f;
        a.N = b; a.Cin = c.z_channels; a.H = H; a.W = W; a.Cout = C;
        ex.launches += 1;
        ex.flops += 2.0 * b * H * W * C * 9.0 * c.z_channels;
        ex.t_begin(KC_MISC, 2.0 * b * H * W * C * 9.0 * c.z_channels, 1, "conv_in", (long long)b * H * W, C, 9 * c.z_channels, 1);
        if (!dry) ex.note(small_conv_in_launch(a, stream));
        ex.t_end("small_conv_in_kernel");
    }
    f = R.resblock(v->mid1, f, H, W);
    f = R.attn(v->mid_attn, f, H, W);
    f = R.resblock(v->mid2, f, H, W);
    for (int lvl = c.num_levels - 1; lvl >= 0; --lvl) {
        for (const VResW& r : v->up[lvl]) {
            f = R.resblock(r, f, H, W);
            C = r.cout;
        }
        if (lvl != 0) {   // Upsample: nearest 2x then conv (LD.py:3498-3511), fused into the conv's loader
            half_t* o = ar.halfs((size_t)b * 4 * H * W * C);
            float* so = reinterpret_cast<float*>(ar.alloc(groupnorm_workspace_bytes(b, 4 * H * W)));
            R.conv(f, C, H, W, 2 * H, 2 * W, 3, v->ups_w[lvl], v->ups_b[lvl], C, nullptr, o, 1, -1, 0, 0, so);
            f = o;
            H *= 2;
            W *= 2;
        }
    }
    {
        half_t* g = ar.halfs((size_t)b * H * W * C);
        R.gn(f, C, H * W, v->no_g, v->no_b, 1, g);
        // conv_out (C -> 3): on the halo-tile MFMA kernel when the image is cut into its 4-row x 128-pixel tiles (weights zero-padded to
        // 32 rows, 8 stored columns, then one elementwise pass for clamp((v + 1) / 2) -> fp32 NHWC): 1427 -> ~200 us at 512x512 x 8;
        // other sizes keep the vector-ALU kernel
        GemmParams q;
        q.conv = 1; q.ksize = 3; q.pad = -1; q.stride = 1;
        q.A = g; q.C1 = C;
        q.Hs = H; q.Ws = W; q.Hv = H; q.Wv = W; q.Ho = H; q.Wo = W;
        q.W = v->co_pad; q.ldw = 9 * C;
        q.M = b * H * W; q.N = 32; q.n_valid = 8; q.K = 9 * C;
        q.bias_n = v->co_pad != nullptr ? v->co_pad + (size_t)32 * 9 * C : nullptr;
        q.partial = ex.splitk_ws; q.partial_bytes = ex.splitk_bytes;
        if (v->co_pad != nullptr && c.out_ch <= 8 && gemm_conv_takes_halo_tile(q)) {
            half_t* t8 = ar.halfs((size_t)b * H * W * 8);
            q.C = t8; q.ldc = 8;
            if (!dry) {   // the three real rows of the weight matrix and the biases into the padded copies (stream-ordered, 7 KB)
                const size_t wb = (size_t)c.out_ch * 9 * C * sizeof(half_t);
                if (hipMemcpyAsync(v->co_pad, v->pt.ptr(v->co_w), wb, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                    hipMemcpyAsync(v->co_pad + (size_t)32 * 9 * C, v->pt.ptr(v->co_b), c.out_ch * sizeof(half_t), hipMemcpyDeviceToDevice, stream) != hipSuccess)
                    ex.note(LD_ERR_HIP);
            }
            const double fl = 2.0 * b * H * W * C * 9.0 * c.out_ch;
            ex.launches += 2;
            ex.flops += fl;
            ex.t_begin(KC_CONV3, fl, 1, "conv_out", (long long)b * H * W, c.out_ch, 9 * C, 1);
            if (!dry && ex.status == LD_OK) ex.note(gemm_launch(q, stream));
            ex.t_end(gemm_last_kernel_name());
            ex.t_begin(KC_MISC, 0.0, 1, "out_finish", (long long)b * H * W, c.out_ch, 0, 1);
            if (!dry && ex.status == LD_OK) ex.note(vae_out_finish_launch(t8, out, (long long)b * H * W, c.out_ch, stream));
            ex.t_end("vae_out_finish_kernel");
        } else {
            SmallConvOutArgs a;
            a.x = g; a.w = v->pt.ptr(v->co_w); a.b = v->pt.ptr(v->co_b);
            a.N = b; a.H = H; a.W = W; a.Cin = C; a.Cout = c.out_ch; a.mode = 1; a.out = out;
            ex.launches += 1;
            ex.flops += 2.0 * b * H * W * C * 9.0 * c.out_ch;
            ex.t_begin(KC_MISC, 2.0 * b * H * W * C * 9.0 * c.out_ch, 1, "conv_out", (long long)b * H * W, c.out_ch, 9 * C, 1);
            if (!dry) ex.note(small_conv_out_launch(a, stream));
            ex.t_end("small_conv_out_kernel");
        }
    }
    v->last_launches = ex.launches;
    v->last_flops = ex.flops;
    if (dry_peak) *dry_peak = ar.peak;
    return ex.status;
}

// VAE.encode's device part (LD.py:6383-6410 → AutoencodingEngine.encode 3475-3481 → Encoder.forward 3731-3758 → quant_conv):
// pixels fp32 NCHW [b][3][8h][8w] already mapped to [-1,1]  ->  moments fp32 NCHW [b][2z][h][w] (mean | logvar)
int run_encode(ld_vae* v, bool dry, const float* px, float* moments, int b, int h, int w, hipStream_t stream, size_t* dry_peak = nullptr) {
    VRun R;
    R.v = v;
    R.n = b;
    Exec& ex = R.ex;
    ex.stream = stream;
    ex.dry = dry;
    Arena plan;
    ex.arena = dry ? &plan : &v->arena;
    ex.splitk_ws = v->splitk_ws;
    ex.splitk_bytes = v->splitk_bytes;
    Arena& ar = *ex.arena;
    ar.release(0);
    const ld_vae_config& c = v->cfg;
    int H = h, W = w;
    for (int l = 1; l < c.num_levels; ++l) {
        H *= 2;
        W *= 2;
    }
    int C = c.ch;
    half_t* f = ar.halfs((size_t)b * H * W * C);
    {
        SmallConvInArgs a;
        a.x = px; a.w = v->pt.ptr(v->e_cin_w); a.b = v->pt.ptr(v->e_cin_b); a.y = f;
        a.N = b; a.Cin = c.out_ch; a.H = H; a.W = W; a.Cout = C;
        ex.launches += 1;
        ex.flops += 2.0 * b * H * W * C * 9.0 * c.out_ch;
        if (!dry) ex.note(small_conv_in_launch(a, stream));
    }
    for (int lvl = 0; lvl < c.num_levels; ++lvl) {
        for (const VResW& r : v->down[lvl]) {
            f = R.resblock(r, f, H, W);
            C = r.cout;
        }
        if (lvl != c.num_levels - 1) {
            // Downsample (LD.py:3514-3528): F.pad(0,1,0,1) then 3x3 stride-2 conv without padding = top/left pad 0,
            // bottom/right zeros supplied by the loader's bounds check
            const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
            half_t* o = ar.halfs((size_t)b * Ho * Wo * C);
            R.conv(f, C, H, W, H, W, 3, v->dwn_w[lvl], v->dwn_b[lvl], C, nullptr, o, 2, 0, Ho, Wo);
            f = o;
            H = Ho;
            W = Wo;
        }
    }
    f = R.resblock(v->e_mid1, f, H, W);
    f = R.attn(v->e_attn, f, H, W);
    f = R.resblock(v->e_mid2, f, H, W);
    {
        half_t* g = ar.halfs((size_t)b * H * W * C);
        R.gn(f, C, H * W, v->e_no_g, v->e_no_b, 1, g);
        const int Z2 = 2 * c.z_channels;
        half_t* m = ar.halfs((size_t)b * H * W * Z2);
        R.conv(g, C, H, W, H, W, 3, v->e_co_w, v->e_co_b, Z2, nullptr, m);
        ex.launches += 1;
        if (!dry && ex.status == LD_OK) ex.note(small_pointwise_launch(m, v->pt.ptr(v->e_q_w), v->pt.ptr(v->e_q_b), moments, b, H * W, Z2, stream));
    }
    if (H != h || W != w) ex.note(LD_ERR_SHAPE);
    v->last_launches = ex.launches;
    v->last_flops = ex.flops;
    if (dry_peak) *dry_peak = ar.peak;
    return ex.status;
}

}  // namespace

extern "C" {

int ld_vae_create(const ld_vae_config* cfg, ld_vae** out) {
    if (cfg == nullptr || out == nullptr) return LD_ERR_ARG;
    ld_vae* v = new ld_vae();
    v->cfg = *cfg;
    int st = build(v);
    if (st == LD_OK) {   // zero-padded [32][9 C] conv_out weights + 32 biases for the MFMA output convolution (rows >= out_ch stay zero)
        const int C0 = v->cfg.ch * v->cfg.ch_mult[0];
        const size_t bytes = ((size_t)32 * 9 * C0 + 32) * sizeof(half_t);
        if (hipMalloc((void**)&v->co_pad, bytes) != hipSuccess || hipMemset(v->co_pad, 0, bytes) != hipSuccess) st = LD_ERR_HIP;
    }
    if (st != LD_OK) {
        v->pt.destroy();
        if (v->co_pad) (void)hipFree(v->co_pad);
        delete v;
        return st;
    }
    *out = v;
    return LD_OK;
}

void ld_vae_destroy(ld_vae* v) {
    if (v == nullptr) return;
    v->pt.destroy();
    v->timing.destroy();
    if (v->co_pad) (void)hipFree(v->co_pad);
    if (v->ws_base) (void)hipFree(v->ws_base);
    delete v;
}

int ld_vae_param_count(const ld_vae* v) { return v ? (int)v->pt.slots.size() : 0; }

int ld_vae_param_info(const ld_vae* v, int i, const char** name, int* ndim, int64_t shape[4]) {
    if (v == nullptr || i < 0 || i >= (int)v->pt.slots.size()) return LD_ERR_ARG;
    const ParamSlot& s = v->pt.slots[i];
    if (name) *name = s.name.c_str();
    if (ndim) *ndim = s.ndim;
    if (shape)
        for (int k = 0; k < 4; ++k) shape[k] = s.shape[k];
    return LD_OK;
}

int ld_vae_load_param(ld_vae* v, const char* name, const void* src, int dtype, void* stream) {
    if (v == nullptr || name == nullptr) return LD_ERR_ARG;
    return v->pt.load(name, src, dtype, (hipStream_t)stream);
}

size_t ld_vae_workspace_bytes(const ld_vae* v) { return v ? v->ws_bytes : 0; }

int ld_vae_reserve(ld_vae* v, int max_b, int max_h, int max_w) {
    if (v == nullptr || max_b < 1 || max_h < 1 || max_w < 1) return LD_ERR_ARG;
    if (v->ws_base) {
        (void)hipFree(v->ws_base);
        v->ws_base = nullptr;
    }
    v->arena = Arena();
    v->plan_b = v->plan_h = v->plan_w = 0;
    size_t peak = 0;
    int st = run_decode(v, true, nullptr, nullptr, max_b, max_h, max_w, nullptr, &peak);
    if (st != LD_OK) return st;
    if (v->cfg.with_encoder) {
        size_t pe = 0;
        st = run_encode(v, true, nullptr, nullptr, max_b, max_h, max_w, nullptr, &pe);
        if (st != LD_OK) return st;
        if (pe > peak) peak = pe;
    }
    const size_t act = (peak + 4095) / 4096 * 4096;
    v->splitk_bytes = (size_t)96 << 20;
    v->ws_bytes = act + v->splitk_bytes + 4096;
    if (hipMalloc((void**)&v->ws_base, v->ws_bytes) != hipSuccess) {
        v->ws_base = nullptr;
        return LD_ERR_HIP;
    }
    v->arena.base = v->ws_base;
    v->arena.cap = act;
    v->splitk_ws = reinterpret_cast<float*>(v->ws_base + act);
    return LD_OK;
}

size_t ld_vae_plan_bytes(ld_vae* v, int b, int h, int w) {
    // host-only dry run of the executor: the workspace a (b, h, w) decode (and encode, when the encoder is loaded) needs, without allocating
    if (v == nullptr || b < 1 || h < 1 || w < 1) return 0;
    size_t peak = 0;
    if (run_decode(v, true, nullptr, nullptr, b, h, w, nullptr, &peak) != LD_OK) return 0;
    if (v->cfg.with_encoder) {
        size_t pe = 0;
        if (run_encode(v, true, nullptr, nullptr, b, h, w, nullptr, &pe) != LD_OK) return 0;
        if (pe > peak) peak = pe;
    }
    v->plan_b = v->plan_h = v->plan_w = 0;          // (the dry run re-planned the arena marks: the next real call plans its own shape)
    return (peak + 4095) / 4096 * 4096 + ((size_t)96 << 20) + 4096;
}

int ld_vae_decode(ld_vae* v, const float* z, float* out, int b, int h, int w, void* stream) {
    if (v == nullptr || z == nullptr || out == nullptr) return LD_ERR_ARG;
    if (v->ws_base == nullptr || !v->pt.all_loaded()) return LD_ERR_STATE;
    if (b < 1 || h < 1 || w < 1) return LD_ERR_SHAPE;
    if (b != v->plan_b || h != v->plan_h || w != v->plan_w) {
        size_t peak = 0;
        const int st = run_decode(v, true, nullptr, nullptr, b, h, w, nullptr, &peak);
        if (st != LD_OK) return st;
        if (peak > v->arena.cap) return LD_ERR_SHAPE;
        v->plan_b = b;
        v->plan_h = h;
        v->plan_w = w;
    }
    return run_decode(v, false, z, out, b, h, w, (hipStream_t)stream);
}

int ld_vae_encode(ld_vae* v, const float* pixels_nchw, float* moments, int b, int h, int w, void* stream) {
    if (v == nullptr || pixels_nchw == nullptr || moments == nullptr) return LD_ERR_ARG;
    if (!v->cfg.with_encoder) return LD_ERR_STATE;
    if (v->ws_base == nullptr || !v->pt.all_loaded()) return LD_ERR_STATE;
    if (b < 1 || h < 1 || w < 1) return LD_ERR_SHAPE;
    size_t peak = 0;
    int st = run_encode(v, true, nullptr, nullptr, b, h, w, nullptr, &peak);
    if (st != LD_OK) return st;
    if (peak > v->arena.cap) return LD_ERR_SHAPE;
    v->plan_b = v->plan_h = v->plan_w = 0;   // the decode plan cache does not cover encode shapes
    return run_encode(v, false, pixels_nchw, moments, b, h, w, (hipStream_t)stream);
}

int ld_vae_profile(ld_vae* v, const float* z, float* out, int b, int h, int w, void* stream) {
    if (v == nullptr) return LD_ERR_ARG;
    v->want_timing = true;
    const int st = ld_vae_decode(v, z, out, b, h, w, stream);
    v->want_timing = false;
    if (st != LD_OK) return st;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return LD_ERR_HIP;
    v->timing.collect();
    return LD_OK;
}

int ld_vae_profile_launches(const ld_vae* v, char* buf, size_t buf_bytes) {
    if (v == nullptr) return LD_ERR_ARG;
    return v->timing.format_launches(buf, buf_bytes);
}

int ld_vae_last_launches(const ld_vae* v) { return v ? v->last_launches : 0; }
double ld_vae_last_flops(const ld_vae* v) { return v ? v->last_flops : 0.0; }

}  // extern "C"
