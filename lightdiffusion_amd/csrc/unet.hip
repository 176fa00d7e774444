// SD1.x UNet executor (host side).  Walks the same block structure UNetModel1.__init__ builds (LD.py:5379-5686)
// and UNetModel1.forward runs (LD.py:5688-5767), but on NHWC fp16 activations — in that layout a feature map IS the
// [tokens, channels] matrix of the SpatialTransformer, so the reference's 32 NCHW<->NLC transposing copies
// (LD.py:4251, 4259) disappear, the skip concats are virtual (two-source A loaders), nearest-upsample is an
// index map inside the conv loader, and every bias / time-embedding add / residual / SiLU / GEGLU is an epilogue.
#include <cstring>

#include "runtime.h"
#include "../../include/ld_mi355x.h"

#ifdef LD_AB_BUILD
// A/B build only: executor-level switches for same-process A/B timing (tools/ab_unet.py); bit 0: no MLP-out fold
static int g_unet_dbg = 0;
extern "C" void ld_debug_unet_flags(int bits) { g_unet_dbg = bits; }
#endif

namespace {

struct ResW {
    std::string prefix;
    int cin = 0, cout = 0, emb_off = 0;
    int gn1_g, gn1_b, c1_w, c1_b, emb_w, emb_b, gn2_g, gn2_b, c2_w, c2_b, sk_w = -1, sk_b = -1;
    // skip fold (derived with the LayerNorm folds): [W2 | Wskip] ([cout][9 cout + cin]) and b2 + bskip — the 1x1 skip_connection runs as a
    // second K segment of out_layers' convolution where that convolution runs on a tap-major kernel (gemm.h S1 / S2); byte offsets into fold_base
    size_t f_c2_w = 0, f_c2_b = 0;
};
struct StW {
    int c = 0, bn = 0, ctx_slot = 0;
    // LN fold (derived at the first forward after a weight load): gamma-folded weights / beta-folded biases (byte offsets into
    // ld_unet::fold_base) and their fp32 row sums, for the projections that consume LN1 (q|k, v), LN2 (q of attn2), LN3 (GEGLU)
    size_t f_qk_w = 0, f_q2_w = 0, f_ff1_w = 0, f_qk_b = 0, f_q2_b = 0, f_ff1_b = 0, f_qk_s = 0, f_q2_s = 0,
           f_ff1_s = 0;
    // MLP-out fold: [Wpo W2 | Wpo] ([C][5C]) and Wpo b2 + bpo — ff.net.2 and proj_out run as ONE two-source contraction
    size_t f_mo_w = 0, f_mo_b = 0;
    int gn_g, gn_b, pin_w, pin_b, ln1_g, ln1_b, q1_w, k1_w, v1_w, o1_w, o1_b, ln2_g, ln2_b, q2_w, k2_w, v2_w, o2_w, o2_b, ln3_g,
        ln3_b, ff1_w, ff1_b, ff2_w, ff2_b, pout_w, pout_b;
};
struct ConvW {
    int cin = 0, cout = 0, w = -1, b = -1;
};
enum LayerKind { L_CONV_IN, L_RES, L_ST, L_DOWN, L_UP };
struct Layer {
    int kind, idx;
};
struct Feat {
    half_t* p;
    int C, H, W;
    float* gn = nullptr;   // GroupNorm partial statistics of this tensor, written by its producer's epilogue ([n][gnP][32][2]; gemm.h gn_part), or null
    int gnP = 0;
};

}  // namespace

struct ld_unet {
    ld_unet_config cfg;
    ParamTable pt;
    std::vector<ResW> res;
    std::vector<StW> st;
    std::vector<ConvW> convs;
    std::vector<std::vector<Layer>> in_blocks, out_blocks;
    std::vector<Layer> mid_block;
    int te0_w, te0_b, te2_w, te2_b, outn_g, outn_b, outc_w, outc_b;
    int emb_total = 0, ted = 0;
    // workspace
    Arena arena;
    char* ws_base = nullptr;
    size_t ws_bytes = 0;
    float* splitk_ws = nullptr;
    size_t splitk_bytes = 0;
    float* log_sigmas = nullptr;   // 1000-entry table, device
    // context (cross-attention K / V^T per transformer, hoisted out of the step)
    int max_n = 0, max_h = 0, max_w = 0, max_tok = 0;
    half_t* ctx16 = nullptr;
    std::vector<half_t*> ctx_k, ctx_vt;
    int ctx_n = 0, ctx_tok = 0, ctx_tpad = 0;
    int plan_n = 0, plan_h = 0, plan_w = 0, plan_pair = 0;   // last shape (and route) validated against the reserved arena
    int last_launches = 0;
    double last_flops = 0.0;
    int* sync_ws = nullptr;        // LD_SYNC_INTS zeroed ints for the in-launch reductions (gemm.h GemmParams::sync)
    // derived copies of the 3x3 convolution weights in the row-resident kernel's layout (conv8.hip), re-derived with the LayerNorm folds
    struct W8 { int slot, N, Cin; size_t off; };
    std::vector<W8> w8_list;
    std::vector<long long> w8_of_slot;   // byte offset into w8_base per parameter slot, -1: none
    char* w8_base = nullptr;
    size_t w8_bytes = 0;
    const half_t* w8(int slot) const {
        return (w8_base != nullptr && slot >= 0 && slot < (int)w8_of_slot.size() && w8_of_slot[slot] >= 0) ? reinterpret_cast<const half_t*>(w8_base + w8_of_slot[slot]) : nullptr;
    }
    void want_w8(int slot, int N, int Cin) {
        if (!conv8_weight_eligible(N, Cin)) return;
        w8_list.push_back({slot, N, Cin, w8_bytes});
        w8_bytes += (conv8_weight_bytes(N, Cin) + 255) / 256 * 256;
    }
    Timing timing;
    bool want_timing = false;
    char* fold_base = nullptr;     // LN-folded copies of the LN-consuming projections (see StW)
    size_t fold_bytes = 0;
    bool fold_dirty = true;        // set by every ld_unet_load_param; cleared when the fold kernels have run
    bool ln_fold = false;          // the GEMM kernels in use implement it (and LD_UNET_NO_LN_FOLD is not set)
};

namespace {

std::string S(const char* fmt, int a = 0, int b = 0) {
    char buf[160];
    snprintf(buf, sizeof buf, fmt, a, b);
    return buf;
}

int add_res(ld_unet* u, const std::string& p, int cin, int cout) {
    ParamTable& t = u->pt;
    ResW r;
    r.prefix = p;
    r.cin = cin;
    r.cout = cout;
    r.gn1_g = t.add(p + ".in_layers.0.weight", PK_VEC, {cin});
    r.gn1_b = t.add(p + ".in_layers.0.bias", PK_VEC, {cin});
    r.c1_w = t.add(p + ".in_layers.2.weight", PK_CONV3, {cout, cin, 3, 3});
    r.c1_b = t.add(p + ".in_layers.2.bias", PK_VEC, {cout});
    u->want_w8(r.c1_w, cout, cin);
    r.gn2_g = t.add(p + ".out_layers.0.weight", PK_VEC, {cout});
    r.gn2_b = t.add(p + ".out_layers.0.bias", PK_VEC, {cout});
    r.c2_w = t.add(p + ".out_layers.3.weight", PK_CONV3, {cout, cout, 3, 3});
    r.c2_b = t.add(p + ".out_layers.3.bias", PK_VEC, {cout});
    u->want_w8(r.c2_w, cout, cout);
    if (cin != cout) {
        r.sk_w = t.add(p + ".skip_connection.weight", PK_MAT, {cout, cin, 1, 1});
        r.sk_b = t.add(p + ".skip_connection.bias", PK_VEC, {cout});
        size_t& fb = u->fold_bytes;
        r.f_c2_w = fb;
        fb += ((size_t)cout * (9 * cout + cin) * sizeof(half_t) + 255) / 256 * 256;
        r.f_c2_b = fb;
        fb += ((size_t)cout * sizeof(half_t) + 255) / 256 * 256;
    }
    r.emb_off = u->emb_total;
    u->emb_total += cout;
    u->res.push_back(r);
    return (int)u->res.size() - 1;
}

int add_st(ld_unet* u, const std::string& p, int c) {
    ParamTable& t = u->pt;
    const int ctx = u->cfg.context_dim;
    StW s;
    s.c = c;
    s.bn = gemm_pick_bn(8 * c);
    s.ctx_slot = (int)u->st.size();
    s.gn_g = t.add(p + ".norm.weight", PK_VEC, {c});
    s.gn_b = t.add(p + ".norm.bias", PK_VEC, {c});
    s.pin_w = t.add(p + ".proj_in.weight", PK_MAT, {c, c, 1, 1});
    s.pin_b = t.add(p + ".proj_in.bias", PK_VEC, {c});
    const std::string b = p + ".transformer_blocks.0";
    s.ln1_g = t.add(b + ".norm1.weight", PK_VEC, {c});
    s.ln1_b = t.add(b + ".norm1.bias", PK_VEC, {c});
    s.q1_w = t.add(b + ".attn1.to_q.weight", PK_MAT, {c, c});
    s.k1_w = t.add(b + ".attn1.to_k.weight", PK_MAT, {c, c}, 16);   // contiguous with to_q: one [2C][C] projection
    s.v1_w = t.add(b + ".attn1.to_v.weight", PK_MAT, {c, c}, 16);   // ... and to_v: one [3C][C] projection (the attention kernel reads V row-major)
    s.o1_w = t.add(b + ".attn1.to_out.0.weight", PK_MAT, {c, c});
    s.o1_b = t.add(b + ".attn1.to_out.0.bias", PK_VEC, {c});
    s.ln2_g = t.add(b + ".norm2.weight", PK_VEC, {c});
    s.ln2_b = t.add(b + ".norm2.bias", PK_VEC, {c});
    s.q2_w = t.add(b + ".attn2.to_q.weight", PK_MAT, {c, c});
    s.k2_w = t.add(b + ".attn2.to_k.weight", PK_MAT, {c, ctx});
    s.v2_w = t.add(b + ".attn2.to_v.weight", PK_MAT, {c, ctx});
    s.o2_w = t.add(b + ".attn2.to_out.0.weight", PK_MAT, {c, c});
    s.o2_b = t.add(b + ".attn2.to_out.0.bias", PK_VEC, {c});
    s.ln3_g = t.add(b + ".norm3.weight", PK_VEC, {c});
    s.ln3_b = t.add(b + ".norm3.bias", PK_VEC, {c});
    s.ff1_w = t.add(b + ".ff.net.0.proj.weight", PK_GEGLU_W, {8 * c, c}, 256, s.bn);
    s.ff1_b = t.add(b + ".ff.net.0.proj.bias", PK_GEGLU_B, {8 * c}, 256, s.bn);
    s.ff2_w = t.add(b + ".ff.net.2.weight", PK_MAT, {c, 4 * c});
    s.ff2_b = t.add(b + ".ff.net.2.bias", PK_VEC, {c});
    s.pout_w = t.add(p + ".proj_out.weight", PK_MAT, {c, c, 1, 1});
    s.pout_b = t.add(p + ".proj_out.bias", PK_VEC, {c});
    {   // LN-fold buffer layout (256-byte aligned pieces)
        size_t& fb = u->fold_bytes;
        auto take = [&](size_t bytes) {
            const size_t o = fb;
            fb += (bytes + 255) / 256 * 256;
            return o;
        };
        const size_t C = (size_t)c;
        s.f_qk_w = take(3 * C * C * sizeof(half_t));   // [Wq ; Wk ; Wv] of attn1
        s.f_q2_w = take(C * C * sizeof(half_t));
        s.f_ff1_w = take(8 * C * C * sizeof(half_t));
        s.f_qk_b = take(3 * C * sizeof(half_t));
        s.f_q2_b = take(C * sizeof(half_t));
        s.f_ff1_b = take(8 * C * sizeof(half_t));
        s.f_qk_s = take(3 * C * sizeof(float));
        s.f_q2_s = take(C * sizeof(float));
        s.f_ff1_s = take(8 * C * sizeof(float));
        s.f_mo_w = take(5 * C * C * sizeof(half_t));
        s.f_mo_b = take(C * sizeof(half_t));
    }
    u->st.push_back(s);
    return (int)u->st.size() - 1;
}

int add_conv(ld_unet* u, const std::string& p, int cin, int cout) {
    ConvW c;
    c.cin = cin;
    c.cout = cout;
    c.w = u->pt.add(p + ".weight", PK_CONV3, {cout, cin, 3, 3});
    c.b = u->pt.add(p + ".bias", PK_VEC, {cout});
    u->convs.push_back(c);
    return (int)u->convs.size() - 1;
}

int build(ld_unet* u) {
    const ld_unet_config& c = u->cfg;
    if (c.num_levels < 1 || c.num_levels > 8 || c.model_channels % 32 || c.num_heads < 1) return LD_ERR_ARG;
    const int mc = c.model_channels;
    u->ted = mc * 4;
    ParamTable& t = u->pt;
    u->te0_w = t.add("time_embed.0.weight", PK_MAT, {u->ted, mc});
    u->te0_b = t.add("time_embed.0.bias", PK_VEC, {u->ted});
    u->te2_w = t.add("time_embed.2.weight", PK_MAT, {u->ted, u->ted});
    u->te2_b = t.add("time_embed.2.bias", PK_VEC, {u->ted});
    u->in_blocks.push_back({{L_CONV_IN, add_conv(u, "input_blocks.0.0", c.in_channels, mc)}});
    int ch = mc, idx = 1, td = 0;
    std::vector<int> chans{mc};
    for (int lvl = 0; lvl < c.num_levels; ++lvl) {
        for (int r = 0; r < c.num_res_blocks[lvl]; ++r) {
            std::vector<Layer> L;
            L.push_back({L_RES, add_res(u, S("input_blocks.%d.0", idx), ch, c.channel_mult[lvl] * mc)});
            ch = c.channel_mult[lvl] * mc;
            if (c.transformer_depth[td++] > 0) L.push_back({L_ST, add_st(u, S("input_blocks.%d.1", idx), ch)});
            u->in_blocks.push_back(L);
            chans.push_back(ch);
            ++idx;
        }
        if (lvl != c.num_levels - 1) {
            u->in_blocks.push_back({{L_DOWN, add_conv(u, S("input_blocks.%d.0.op", idx), ch, ch)}});
            chans.push_back(ch);
            ++idx;
        }
    }
    u->mid_block.push_back({L_RES, add_res(u, "middle_block.0", ch, ch)});
    if (c.transformer_depth_middle > 0) u->mid_block.push_back({L_ST, add_st(u, "middle_block.1", ch)});
    u->mid_block.push_back({L_RES, add_res(u, "middle_block.2", ch, ch)});
    // the reference pops transformer_depth_output from the END of the list (LD.py:5621)
    int n_out = 0;
    for (int lvl = 0; lvl < c.num_levels; ++lvl) n_out += c.num_res_blocks[lvl] + 1;
    int tdo = n_out;
    idx = 0;
    for (int lvl = c.num_levels - 1; lvl >= 0; --lvl) {
        for (int i = 0; i <= c.num_res_blocks[lvl]; ++i) {
            const int ich = chans.back();
            chans.pop_back();
            std::vector<Layer> L;
            L.push_back({L_RES, add_res(u, S("output_blocks.%d.0", idx), ch + ich, mc * c.channel_mult[lvl])});
            ch = mc * c.channel_mult[lvl];
            int j = 1;
            if (c.transformer_depth_output[--tdo] > 0) {
                L.push_back({L_ST, add_st(u, S("output_blocks.%d.%d", idx, j), ch)});
                ++j;
            }
            if (lvl && i == c.num_res_blocks[lvl]) {
                L.push_back({L_UP, add_conv(u, S("output_blocks.%d.%d.conv", idx, j), ch, ch)});
                u->want_w8(u->convs.back().w, ch, ch);
            }
            u->out_blocks.push_back(L);
            ++idx;
        }
    }
    u->outn_g = t.add("out.0.weight", PK_VEC, {ch});
    u->outn_b = t.add("out.0.bias", PK_VEC, {ch});
    u->outc_w = t.add("out.2.weight", PK_CONV3, {c.out_channels, mc, 3, 3});
    u->outc_b = t.add("out.2.bias", PK_VEC, {c.out_channels});
    // the 22 emb_layers Linears as ONE [sum(Cout)][4*mc] matrix (tightly packed) -> one GEMM per step
    for (size_t i = 0; i < u->res.size(); ++i)
        u->res[i].emb_w = t.add(u->res[i].prefix + ".emb_layers.1.weight", PK_MAT, {u->res[i].cout, u->ted}, i == 0 ? 256 : 16);
    for (size_t i = 0; i < u->res.size(); ++i)
        u->res[i].emb_b = t.add(u->res[i].prefix + ".emb_layers.1.bias", PK_VEC, {u->res[i].cout}, i == 0 ? 256 : 16);
    return t.finalize();
}

// ---------------------------------------------------------------------------------------------------- forward pieces
struct Run {
    ld_unet* u;
    Exec ex;
    int n;                   // samples the kernels run on
    int n_alloc = 0;         // samples every tensor is SIZED for (= n, except in front of the first cross-attention of a CFG pair)
    bool pair_pending = false;   // CFG pair (ld_unet_forward_pair): the layers in front of the first cross-attention see the same input in the uncond and
                                 // the cond half of the batch, so they run ONCE on n = n_alloc / 2 samples and their results are copied into the second half
    const half_t* emb_all;   // [n_alloc][emb_total]
    half_t* P(int slot) const { return u->pt.ptr(slot); }
    int na() const { return n_alloc > 0 ? n_alloc : n; }
    // first half of a [2 * half_bytes] tensor -> its second half (a memcpy node of the step's hipGraph; layouts other than SD1.5's, whose hand-over is one dup_halves launch)
    void dup(void* base, size_t half_bytes) {
        ex.launches += 1;
        ex.t_begin(KC_MISC, 0.0, 1, "dup", (long long)half_bytes, 0, 0, 1);
        if (!ex.dry && ex.status == LD_OK &&
            hipMemcpyAsync(static_cast<char*>(base) + half_bytes, base, half_bytes, hipMemcpyDeviceToDevice, ex.stream) != hipSuccess)
            ex.note(LD_ERR_HIP);
        ex.t_end("hipMemcpyAsync(pair)");
    }

    // GroupNorm partial statistics of a contraction's output (gemm.h gn_part): where the launch can emit them (its split-K second pass, or
    // the row-resident kernel) *done receives the chunk count and the GroupNorm that follows skips its statistics launch
    void want_stats(GemmParams& p, float* buf, int HW, int* done) {
        p.gn_part = buf;
        p.gn_P = gn_num_chunks(n, HW);
        p.gn_HW = HW;
        p.gn_ppb = (HW + p.gn_P - 1) / p.gn_P;
        p.gn_part_done = done;
    }

    half_t* conv3(const half_t* x1, int C1, const half_t* x2, int C2, int Hs, int Ws, int Hv, int Wv, int stride, int wslot, int bslot,
                  int cout, const half_t* rowvec, int ldrv, const half_t* R, half_t* out, int* Ho_, int* Wo_, int ksize = 3, float* stats = nullptr,
                  int* stats_done = nullptr) {
        const int Ho = ksize == 3 ? (Hv - 1) / stride + 1 : Hv, Wo = ksize == 3 ? (Wv - 1) / stride + 1 : Wv;
        GemmParams p;
        p.conv = 1;
        p.ksize = ksize;
        p.A = x1; p.A2 = x2; p.C1 = C1; p.C2 = C2;
        p.Hs = Hs; p.Ws = Ws; p.Hv = Hv; p.Wv = Wv; p.Ho = Ho; p.Wo = Wo; p.stride = stride;
        p.W = P(wslot); p.ldw = ksize * ksize * (C1 + C2);
        p.W8 = ksize == 3 ? u->w8(wslot) : nullptr;
        p.M = n * Ho * Wo; p.N = cout; p.K = ksize * ksize * (C1 + C2);
        p.bias_n = P(bslot);
        p.rowvec = rowvec; p.rows_per_vec = Ho * Wo; p.ldrv = ldrv;
        p.R = R; p.ldr = cout;
        p.C = out; p.ldc = cout;
        if (stats != nullptr) want_stats(p, stats, Ho * Wo, stats_done);
        ex.gemm(p);
        if (Ho_) *Ho_ = Ho;
        if (Wo_) *Wo_ = Wo;
        return out;
    }

    void linear(const half_t* x, int lda, int wslot, int bslot, const half_t* R, half_t* y, int M, int N, int K, int act = 0, int bn = 0) {
        GemmParams p;
        p.A = x; p.lda = lda;
        p.W = P(wslot); p.ldw = K;
        p.M = M; p.N = N; p.K = K;
        p.bias_n = bslot >= 0 ? P(bslot) : nullptr;
        p.R = R; p.ldr = (act == 2 ? N / 2 : N);
        p.act = act; p.bn = bn;
        p.C = y; p.ldc = (act == 2 ? N / 2 : N);
        ex.gemm(p);
    }

    // ResBlock1._forward, LD.py:5273-5287.  Input = channel concat of (x1,C1) and (x2,C2).
    // in_stats / in_P: GroupNorm partial statistics of x1 from its producer (only usable when there is no second source)
    Feat resblock(const ResW& r, const half_t* x1, int C1, const half_t* x2, int C2, int H, int W, float* in_stats = nullptr, int in_P = 0) {
        Arena& ar = *ex.arena;
        const size_t M = (size_t)na() * H * W;                   // rows the tensors are sized for (the kernels run on n * H * W)
        half_t* out = ar.halfs(M * r.cout);
        float* gno = reinterpret_cast<float*>(ar.alloc(groupnorm_workspace_bytes(na(), H * W)));   // statistics of `out` (for the next GroupNorm)
        int gno_done = 0;
        if (C2 != 0) in_stats = nullptr;
        const size_t mk = ar.mark();
        // in_layers: GroupNorm + SiLU + conv3x3 (+ the time-embedding row vector); out_layers: GroupNorm + SiLU + conv3x3 (+ skip).
        // Each GroupNorm is fused into its convolution's A operand where that convolution runs on the halo-tile kernel
        // (Exec::gn_silu_conv); g1 / g2 are only written on the two-pass route.
        auto conv_params = [&](const half_t* a1, int c1, const half_t* a2, int c2, int wslot, int bslot, const half_t* rowvec, int ldrv, const half_t* R,
                               half_t* dst) {
            GemmParams p;
            p.conv = 1; p.ksize = 3;
            p.A = a1; p.A2 = a2; p.C1 = c1; p.C2 = c2;
            p.Hs = H; p.Ws = W; p.Hv = H; p.Wv = W; p.Ho = H; p.Wo = W; p.stride = 1;
            p.W = P(wslot); p.ldw = 9 * (c1 + c2);
            p.W8 = u->w8(wslot);
            p.M = n * H * W; p.N = r.cout; p.K = 9 * (c1 + c2);
            p.bias_n = P(bslot);
            p.rowvec = rowvec; p.rows_per_vec = H * W; p.ldrv = ldrv;
            p.R = R; p.ldr = r.cout;
            p.C = dst; p.ldc = r.cout;
            return p;
        };
        half_t* g1 = ar.halfs(M * r.cin);
        half_t* h1 = ar.halfs(M * r.cout);
        // The 1x1 skip_connection (LD.py:5267): folded into out_layers' convolution as a second K segment where that convolution runs on a
        // tap-major kernel (same FLOPs at the 3x3 kernels' rate, one launch and one [M][cout] round trip less); a launch of its own in front
        // of the halo-tile and row-resident kernels.  (On a second stream beside in_layers in a batch-1 step it measured 2.9 % SLOWER than in
        // order: tools/experiments/fork_join_skip_conv_r05.patch.txt.)
        const half_t* skip = x1;
        bool fold_skip = false;
        if (r.sk_w >= 0) {
            GemmParams probe = conv_params(h1, r.cout, nullptr, 0, r.c2_w, r.c2_b, nullptr, 0, nullptr, out);
            fold_skip = u->ln_fold && u->fold_base != nullptr && ex.conv_takes_skip_segment(probe);
#ifdef LD_AB_BUILD
            if (g_unet_dbg & 8) fold_skip = false;
#endif
            half_t* sk = (!fold_skip || ex.dry) ? ar.halfs(M * r.cout) : nullptr;   // (a planning run sizes for either route: ld_unet_reserve plans before the split-K scratch exists)
            if (!fold_skip) {
                conv3(x1, C1, x2, C2, H, W, H, W, 1, r.sk_w, r.sk_b, r.cout, nullptr, 0, nullptr, sk, nullptr, nullptr, 1);
                skip = sk;
            }
        }
        // the GroupNorm statistics of h1 (out_layers' norm) come from conv1's split-K second pass where it has one (gemm.h gn_part)
        float* gnp = reinterpret_cast<float*>(ar.alloc(groupnorm_workspace_bytes(na(), H * W)));
        int gnp_done = 0;
        {
            GemmParams c1 = conv_params(x1, C1, x2, C2, r.c1_w, r.c1_b, emb_all + r.emb_off, u->emb_total, nullptr, h1);
            want_stats(c1, gnp, H * W, &gnp_done);
            ex.gn_silu_conv(c1, n, H * W, P(r.gn1_g), P(r.gn1_b), 1e-5f, g1, in_stats, in_P);
        }
        half_t* g2 = g1;   // g1 is dead once conv1 has consumed it (stream order); reuse when it is large enough
        if (r.cout > r.cin) g2 = ar.halfs(M * r.cout);
        {
            GemmParams c2 = conv_params(h1, r.cout, nullptr, 0, r.c2_w, r.c2_b, nullptr, 0, skip, out);
            if (fold_skip) {
                c2.S1 = x1; c2.SC1 = C1; c2.S2 = x2; c2.SC2 = C2;
                c2.W = reinterpret_cast<const half_t*>(u->fold_base + r.f_c2_w);
                c2.K = 9 * r.cout + r.cin; c2.ldw = c2.K;
                c2.bias_n = reinterpret_cast<const half_t*>(u->fold_base + r.f_c2_b);
                c2.R = nullptr;
                c2.W8 = nullptr;
            }
            want_stats(c2, gno, H * W, &gno_done);
            ex.gn_silu_conv(c2, n, H * W, P(r.gn2_g), P(r.gn2_b), 1e-5f, g2, gnp_done ? gnp : nullptr, gnp_done);
        }
        ar.release(mk);
        return {out, r.cout, H, W, gno_done ? gno : nullptr, gno_done};
    }

    // SpatialTransformer.forward (LD.py:4239-4262) around BasicTransformerBlock._forward (LD.py:4117-4162)
    Feat transformer(const StW& s, const half_t* x, int H, int W, float* in_stats = nullptr, int in_P = 0) {
        Arena& ar = *ex.arena;
        const int C = s.c, L = H * W, heads = u->cfg.num_heads, d = C / heads;
        int M = n * L;                                           // rows the kernels run on (doubles at the split of a CFG pair, below)
        const size_t Ma = (size_t)na() * L;                      // rows the tensors are sized for
        half_t* out = ar.halfs(Ma * C);
        float* gno = reinterpret_cast<float*>(ar.alloc(groupnorm_workspace_bytes(na(), L)));   // statistics of `out` (for the next GroupNorm)
        int gno_done = 0;
        const size_t mk = ar.mark();
        half_t* g = ar.halfs(Ma * C);
        ex.groupnorm(x, C, nullptr, 0, n, L, P(s.gn_g), P(s.gn_b), 1e-6f, 0, g, in_stats, in_P);
        half_t* t = ar.halfs(Ma * C);
        half_t* nrm = g;               // reuse (only the un-folded path materialises LN(x))
        half_t* qkv = ar.halfs(Ma * 3 * C);
        half_t* ao = ar.halfs(Ma * C);
        half_t* ff = nullptr;
        const bool fold = u->ln_fold;
        // LN fold: the three LayerNorms disappear into the GEMMs around them.  The GEMM that WRITES the residual stream t also
        // writes per-row (sum, sum of squares) partials of t (one pair per N tile); the projections that read LN(t) run on t itself
        // with gamma folded into their weights and finish  rstd * (acc - mu * wsum) + b'  in the epilogue (gemm.h).
        float* stat = fold ? reinterpret_cast<float*>(ar.alloc((size_t)((C + 63) / 64) * Ma * 2 * sizeof(float))) : nullptr;
        int parts = 0;
        char* fb = u->fold_base;
        auto producer = [&](const half_t* x_, int lda, int wslot, int bslot, const half_t* R, int K) {   // t = x_ W^T + b (+ R), with row stats
            GemmParams p;
            p.A = x_; p.lda = lda;
            p.W = P(wslot); p.ldw = K;
            p.M = M; p.N = C; p.K = K;
            p.bias_n = P(bslot);
            p.R = R; p.ldr = C;
            p.C = t; p.ldc = C;
            if (fold) {
                p.stat_out = stat;
                p.stat_parts_out = &parts;
            }
            ex.gemm(p);
        };
        auto ln_args = [&](GemmParams& p, size_t wsum_off) {
            p.ln_stat = stat; p.ln_parts = parts; p.ln_rows = M;
            p.ln_inv_c = 1.0f / (float)C; p.ln_eps = 1e-5f;
            p.ln_wsum = reinterpret_cast<const float*>(fb + wsum_off);
        };
        producer(g, C, s.pin_w, s.pin_b, nullptr, C);
        // ---- self attention: x += to_out(attn(LN1(x)))
        if (!fold) ex.layernorm(t, P(s.ln1_g), P(s.ln1_b), nrm, M, C);
        {   // [q | k | v] = LN1(t) [Wq ; Wk ; Wv]^T — one launch; the attention kernel reads V row-major (transposing LDS reads), so there
            // is no V^T projection
            GemmParams p;
            p.A = fold ? t : nrm; p.lda = C;
            p.W = fold ? reinterpret_cast<const half_t*>(fb + s.f_qk_w) : P(s.q1_w); p.ldw = C;
            p.M = M; p.N = 3 * C; p.K = C;
            p.C = qkv; p.ldc = 3 * C;
            if (fold) {
                p.bias_n = reinterpret_cast<const half_t*>(fb + s.f_qk_b);
                ln_args(p, s.f_qk_s);
            }
            ex.gemm(p);
        }
        {
            AttnParams a;
            a.Q = qkv; a.ldq = 3 * C; a.sQ = (long long)L * 3 * C;
            a.K = qkv + C; a.ldk = 3 * C; a.sK = (long long)L * 3 * C;
            a.V = qkv + 2 * C; a.ldv = 3 * C; a.sV = (long long)L * 3 * C;
            a.O = ao; a.ldo = C; a.sO = (long long)L * C;
            a.B = n; a.H = heads; a.Lq = L; a.Lk = L; a.d = d;
            a.scale = 1.0f / sqrtf((float)d);
            ex.attention(a);
        }
        // CFG pair: everything up to the cross-attention's K / V sees the same input in both halves of the batch and runs on the first half only
        // — also the out-projection of attn1, the LayerNorm-2 statistics it emits and the q projection of attn2 (only K / V of attn2 read the
        // conditioning).  Then the residual stream t, q2 and the block's input (its residual at the end) are copied into the second half and
        // everything else runs on all n_alloc samples (the statistics of the first half are not needed again: the out-projection of attn2
        // rewrites them for all rows).
        const bool split_here = pair_pending;
        producer(ao, C, s.o1_w, s.o1_b, t, C);
        // ---- cross attention against the hoisted context K / V^T
        if (!fold) ex.layernorm(t, P(s.ln2_g), P(s.ln2_b), nrm, M, C);
        half_t* q2 = qkv;   // reuse
        {
            GemmParams p;
            p.A = fold ? t : nrm; p.lda = C;
            p.W = fold ? reinterpret_cast<const half_t*>(fb + s.f_q2_w) : P(s.q2_w); p.ldw = C;
            p.M = M; p.N = C; p.K = C;
            p.C = q2; p.ldc = C;
            if (fold) {
                p.bias_n = reinterpret_cast<const half_t*>(fb + s.f_q2_b);
                ln_args(p, s.f_q2_s);
            }
            ex.gemm(p);
        }
        if (split_here) {
            const size_t hb = (size_t)M * C * sizeof(half_t);
            DupArgs da;
            da.count = 3;
            da.base[0] = reinterpret_cast<char*>(t); da.base[1] = reinterpret_cast<char*>(q2); da.base[2] = reinterpret_cast<char*>(const_cast<half_t*>(x));
            da.bytes[0] = da.bytes[1] = da.bytes[2] = hb;
            ex.launches += 1;
            ex.t_begin(KC_MISC, 0.0, 1, "dup3", (long long)hb, 0, 0, 1);
            if (!ex.dry && ex.status == LD_OK) ex.note(dup_halves_launch(da, ex.stream));
            ex.t_end("dup_halves_kernel");
            pair_pending = false;
            n = n_alloc;
            M = n * L;
        }
        {
            const int Tp = u->ctx_tpad;
            AttnParams a;
            a.Q = q2; a.ldq = C; a.sQ = (long long)L * C;
            a.K = u->ctx_k[s.ctx_slot]; a.ldk = C; a.sK = (long long)Tp * C;
            a.Vt = u->ctx_vt[s.ctx_slot]; a.ldvt = Tp; a.sV = (long long)C * Tp;
            a.O = ao; a.ldo = C; a.sO = (long long)L * C;
            a.B = n; a.H = heads; a.Lq = L; a.Lk = u->ctx_tok; a.d = d;
            a.scale = 1.0f / sqrtf((float)d);
            ex.attention(a);
        }
        producer(ao, C, s.o2_w, s.o2_b, t, C);
        // ---- GEGLU feed-forward: x = ff2(a * gelu(gate)) + x
        if (!fold) ex.layernorm(t, P(s.ln3_g), P(s.ln3_b), nrm, M, C);
        ff = ar.halfs(Ma * 4 * C);
        {
            GemmParams p;
            p.A = fold ? t : nrm; p.lda = C;
            p.W = fold ? reinterpret_cast<const half_t*>(fb + s.f_ff1_w) : P(s.ff1_w); p.ldw = C;
            p.M = M; p.N = 8 * C; p.K = C;
            p.bias_n = fold ? reinterpret_cast<const half_t*>(fb + s.f_ff1_b) : P(s.ff1_b);
            p.act = 2; p.bn = s.bn;
            p.C = ff; p.ldc = 4 * C; p.ldr = 4 * C;
            if (fold) ln_args(p, s.f_ff1_s);
            ex.gemm(p);
        }
#ifdef LD_AB_BUILD
        if (fold && !(g_unet_dbg & 1)) {
#else
        if (fold) {
#endif
            // out = x + proj_out(t + ff2(ff)) as one contraction over [ff | t] (K = 5C) against the folded [Wpo W2 | Wpo] (misc.hip)
            GemmParams p;
            p.conv = 1; p.ksize = 1;
            p.A = ff; p.A2 = t; p.C1 = 4 * C; p.C2 = C;
            p.Hs = p.Hv = p.Ho = H; p.Ws = p.Wv = p.Wo = W; p.stride = 1;
            p.W = reinterpret_cast<const half_t*>(fb + s.f_mo_w); p.ldw = 5 * C;
            p.M = M; p.N = C; p.K = 5 * C;
            p.bias_n = reinterpret_cast<const half_t*>(fb + s.f_mo_b);
            p.R = x; p.ldr = C;
            p.C = out; p.ldc = C;
            want_stats(p, gno, L, &gno_done);
            ex.gemm(p);
        } else {
            linear(ff, 4 * C, s.ff2_w, s.ff2_b, t, t, M, C, 4 * C);
            linear(t, C, s.pout_w, s.pout_b, x, out, M, C, C);
        }
        ar.release(mk);
        return {out, C, H, W, gno_done ? gno : nullptr, gno_done};
    }
};

// LN fold: derive W' = W diag(gamma), b' = b + W beta and the fp32 row sums of W' for the projections that read LN1 / LN2 / LN3
// (64 small launches, once per weight load; stream-ordered before the forward that needs them).
int fold_layernorms(ld_unet* u, hipStream_t stream) {
    for (const StW& s : u->st) {
        const int C = s.c;
        char* fb = u->fold_base;
        auto H = [&](size_t off) { return reinterpret_cast<half_t*>(fb + off); };
        auto F = [&](size_t off) { return reinterpret_cast<float*>(fb + off); };
        const half_t* P_q1 = u->pt.ptr(s.q1_w);   // to_q and to_k are contiguous: one [2C][C] matrix
        int st = ln_fold_launch(P_q1, 3 * C, C, u->pt.ptr(s.ln1_g), u->pt.ptr(s.ln1_b), nullptr, H(s.f_qk_w), H(s.f_qk_b), F(s.f_qk_s), stream);
        if (st == LD_OK) st = ln_fold_launch(u->pt.ptr(s.q2_w), C, C, u->pt.ptr(s.ln2_g), u->pt.ptr(s.ln2_b), nullptr, H(s.f_q2_w), H(s.f_q2_b), F(s.f_q2_s), stream);
        if (st == LD_OK) st = ln_fold_launch(u->pt.ptr(s.ff1_w), 8 * C, C, u->pt.ptr(s.ln3_g), u->pt.ptr(s.ln3_b), u->pt.ptr(s.ff1_b), H(s.f_ff1_w), H(s.f_ff1_b), F(s.f_ff1_s), stream);
        if (st == LD_OK) st = mlp_out_fold_launch(u->pt.ptr(s.pout_w), u->pt.ptr(s.ff2_w), u->pt.ptr(s.ff2_b), u->pt.ptr(s.pout_b), C, H(s.f_mo_w), H(s.f_mo_b), stream);
        if (st != LD_OK) return st;
    }
    for (const ResW& r : u->res) {
        if (r.sk_w < 0) continue;
        const int st = skip_fold_launch(u->pt.ptr(r.c2_w), u->pt.ptr(r.sk_w), u->pt.ptr(r.c2_b), u->pt.ptr(r.sk_b), r.cout, 9 * r.cout, r.cin,
                                        reinterpret_cast<half_t*>(u->fold_base + r.f_c2_w), reinterpret_cast<half_t*>(u->fold_base + r.f_c2_b), stream);
        if (st != LD_OK) return st;
    }
    u->fold_dirty = false;
    return LD_OK;
}

// the row-resident kernel's copies of the 3x3 convolution weights (conv8.hip), once per weight load
int derive_conv8_weights(ld_unet* u, hipStream_t stream) {
    for (const ld_unet::W8& w : u->w8_list) {
        const int st = conv8_repack_launch(u->pt.ptr(w.slot), w.N, w.Cin, reinterpret_cast<half_t*>(u->w8_base + w.off), stream);
        if (st != LD_OK) return st;
    }
    return LD_OK;
}

// `pair`: classifier-free-guidance pair (ld_unet_forward_pair).  x / sigma hold n / 2 samples; the batch is [uncond x n/2 ; cond x n/2] of the SAME latents
// (what calc_cond_batch feeds the model: cat([x, x]), LD.py:2515-2547); the resident context has n rows.  The layers in front of the first
// cross-attention run once on n / 2 samples (Run::pair_pending), everything else on n.  out: n samples.
int run_forward(ld_unet* u, bool dry, const float* x, const float* sigma, float* out, int n, int h, int w, int eps_only, hipStream_t stream,
                size_t* dry_peak = nullptr, bool pair = false) {
    if (!dry && u->fold_dirty) {
        if (u->w8_base != nullptr) {
            const int st = derive_conv8_weights(u, stream);
            if (st != LD_OK) return st;
        }
        if (u->ln_fold) {
            const int st = fold_layernorms(u, stream);
            if (st != LD_OK) return st;
        }
        u->fold_dirty = false;
    }
    if (pair && (n & 1)) return LD_ERR_SHAPE;
    Run R;
    R.u = u;
    R.n = n;
    R.n_alloc = n;
    Exec& ex = R.ex;
    ex.stream = stream;
    ex.dry = dry;
    Arena plan;   // dry runs bump a private arena (no base): sizes only
    ex.arena = dry ? &plan : &u->arena;
    ex.splitk_ws = u->splitk_ws;
    ex.splitk_bytes = u->splitk_bytes;
    ex.sync_ws = u->sync_ws;
#ifdef LD_AB_BUILD
    ex.ab_flags = g_unet_dbg;
#endif
    if (u->want_timing && !dry) {
        u->timing.reset();
        ex.timing = &u->timing;
    }
    Arena& ar = *ex.arena;
    ar.release(0);
    const ld_unet_config& c = u->cfg;
    const int mc = c.model_channels, ted = u->ted;
    const int in_mod = pair ? n / 2 : 0;   // CFG pair: x / sigma hold n / 2 samples; the kernels that read them per sample index n % in_mod
    if (pair) R.pair_pending = true;

    // timestep embedding -> time_embed MLP -> all ResBlock emb_layers at once (every consumer applies SiLU first)
    half_t* temb = ar.halfs((size_t)n * mc);
    ex.launches += 1;
    ex.t_begin(KC_MISC, 0.0, 1);
    if (!dry) ex.note(timestep_embed_launch(sigma, u->log_sigmas, 1000, n, mc, temb, nullptr, stream, in_mod));
    ex.t_end("timestep_embed_kernel");
    half_t* e1 = ar.halfs((size_t)n * ted);
    R.linear(temb, mc, u->te0_w, u->te0_b, nullptr, e1, n, ted, mc, 1);
    half_t* semb = ar.halfs((size_t)n * ted);
    R.linear(e1, ted, u->te2_w, u->te2_b, nullptr, semb, n, ted, ted, 1);
    half_t* emb_all = ar.halfs((size_t)n * u->emb_total);
    R.linear(semb, ted, u->res[0].emb_w, u->res[0].emb_b, nullptr, emb_all, n, u->emb_total, ted);
    R.emb_all = emb_all;

    auto run_layers = [&](const std::vector<Layer>& layers, Feat f, const half_t* x2, int C2, int outH, int outW) -> Feat {
        for (const Layer& L : layers) {
            switch (L.kind) {
                case L_RES: {
                    f = R.resblock(u->res[L.idx], f.p, f.C, x2, C2, f.H, f.W, f.gn, f.gnP);
                    x2 = nullptr;
                    C2 = 0;
                    break;
                }
                case L_ST: f = R.transformer(u->st[L.idx], f.p, f.H, f.W, f.gn, f.gnP); break;
                case L_DOWN: {
                    const ConvW& cw = u->convs[L.idx];
                    const int Ho = (f.H - 1) / 2 + 1, Wo = (f.W - 1) / 2 + 1;
                    if (R.pair_pending) {                         // (not in front of the first transformer of any supported layout; be safe)
                        R.dup(const_cast<half_t*>(f.p), (size_t)R.n * f.H * f.W * f.C * sizeof(half_t));
                        R.pair_pending = false;
                        R.n = n;
                    }
                    half_t* o = ar.halfs((size_t)n * Ho * Wo * cw.cout);
                    float* gno = reinterpret_cast<float*>(ar.alloc(groupnorm_workspace_bytes(n, Ho * Wo)));
                    int gno_done = 0;
                    R.conv3(f.p, f.C, nullptr, 0, f.H, f.W, f.H, f.W, 2, cw.w, cw.b, cw.cout, nullptr, 0, nullptr, o, nullptr, nullptr, 3, gno, &gno_done);
                    f = {o, cw.cout, Ho, Wo, gno_done ? gno : nullptr, gno_done};
                    break;
                }
                case L_UP: {   // Upsample1: nearest resize to the next skip's H x W, then conv (LD.py:5141-5152)
                    const ConvW& cw = u->convs[L.idx];
                    const int Hv = outH > 0 ? outH : f.H * 2, Wv = outW > 0 ? outW : f.W * 2;
                    half_t* o = ar.halfs((size_t)n * Hv * Wv * cw.cout);
                    R.conv3(f.p, f.C, nullptr, 0, f.H, f.W, Hv, Wv, 1, cw.w, cw.b, cw.cout, nullptr, 0, nullptr, o, nullptr, nullptr);
                    f = {o, cw.cout, Hv, Wv};
                    break;
                }
                default: break;
            }
        }
        return f;
    };

    std::vector<Feat> hs;
    Feat f{nullptr, 0, h, w};
    {   // input_blocks[0]: conv_in fused with EPS.calculate_input and the fp32 NCHW -> fp16 NHWC layout change
        const ConvW& cw = u->convs[u->in_blocks[0][0].idx];
        half_t* o = ar.halfs((size_t)n * h * w * mc);
        if (R.pair_pending) R.n = n / 2;                          // the shared prefix runs on the first half (see Run::pair_pending)
        SmallConvInArgs a;
        a.x = x; a.scale_sigma = sigma; a.w = u->pt.ptr(cw.w); a.b = u->pt.ptr(cw.b); a.y = o;
        a.N = R.n; a.Cin = c.in_channels; a.H = h; a.W = w; a.Cout = mc;
        if (R.pair_pending) a.dup_off = (long long)R.n * h * w * mc;   // a skip connection: read by the last output block on all n samples
        ex.launches += 1;
        ex.flops += 2.0 * R.n * h * w * mc * 9.0 * c.in_channels;
        ex.t_begin(KC_MISC, 2.0 * R.n * h * w * mc * 9.0 * c.in_channels, 1);
        if (!dry) ex.note(small_conv_in_launch(a, stream));
        ex.t_end("small_conv_in_kernel");
        f = {o, mc, h, w};
        hs.push_back(f);
    }
    for (size_t b = 1; b < u->in_blocks.size(); ++b) {
        f = run_layers(u->in_blocks[b], f, nullptr, 0, 0, 0);
        if (R.pair_pending && b == 1) {                          // no transformer in the first block: its output is the last shared tensor
            R.dup(const_cast<half_t*>(f.p), (size_t)R.n * f.H * f.W * f.C * sizeof(half_t));
            R.pair_pending = false;
            R.n = n;
            f.gn = nullptr;                                      // (its statistics cover the first half of the batch only)
            f.gnP = 0;
        }
        hs.push_back(f);
    }
    if (R.pair_pending) {                                        // (a UNet without input blocks: nothing left to share)
        R.pair_pending = false;
        R.n = n;
    }
    f = run_layers(u->mid_block, f, nullptr, 0, 0, 0);
    for (size_t b = 0; b < u->out_blocks.size(); ++b) {
        const Feat skip = hs.back();
        hs.pop_back();
        if (skip.H != f.H || skip.W != f.W) ex.note(LD_ERR_SHAPE);
        const int oh = hs.empty() ? 0 : hs.back().H, ow = hs.empty() ? 0 : hs.back().W;
        f = run_layers(u->out_blocks[b], f, skip.p, skip.C, oh, ow);
    }
    {   // out: GroupNorm + SiLU, 3x3 conv to 4 channels fused with EPS.calculate_denoised, back to fp32 NCHW
        half_t* g = ar.halfs((size_t)n * f.H * f.W * f.C);
        ex.groupnorm(f.p, f.C, nullptr, 0, n, f.H * f.W, u->pt.ptr(u->outn_g), u->pt.ptr(u->outn_b), 1e-5f, 1, g, f.gn, f.gnP);
        SmallConvOutArgs a;
        a.x = g; a.w = u->pt.ptr(u->outc_w); a.b = u->pt.ptr(u->outc_b);
        a.N = n; a.H = f.H; a.W = f.W; a.Cin = f.C; a.Cout = c.out_channels;
        a.mode = eps_only ? 2 : 0;
        a.x_in = x; a.sigma = sigma; a.out = out; a.in_mod = in_mod;
        ex.launches += 1;
        ex.flops += 2.0 * n * f.H * f.W * f.C * 9.0 * c.out_channels;
        ex.t_begin(KC_MISC, 2.0 * n * f.H * f.W * f.C * 9.0 * c.out_channels, 1);
        if (!dry) ex.note(small_conv_out_launch(a, stream));
        ex.t_end("small_conv_out_kernel");
    }
    u->last_launches = ex.launches;
    u->last_flops = ex.flops;
    if (dry_peak) *dry_peak = ar.peak;
    return ex.status;
}

}  // namespace

// ==================================================================================================== C ABI
extern "C" {

int ld_unet_create(const ld_unet_config* cfg, ld_unet** out) {
    if (cfg == nullptr || out == nullptr) return LD_ERR_ARG;
    ld_unet* u = new ld_unet();
    u->cfg = *cfg;
    const int st = build(u);
    if (st != LD_OK) {
        u->pt.destroy();
        delete u;
        return st;
    }
    // sigma table of ModelSamplingDiscrete (LD.py:1300-1326): scaled-linear betas in fp64, log taken in fp64
    std::vector<float> ls(1000);
    {
        const double b0 = sqrt(0.00085), b1 = sqrt(0.012);
        double ac = 1.0;
        for (int i = 0; i < 1000; ++i) {
            const double sb = b0 + (b1 - b0) * (double)i / 999.0;
            ac *= 1.0 - sb * sb;
            ls[i] = (float)log(sqrt((1.0 - ac) / ac));
        }
    }
    u->w8_of_slot.assign(u->pt.slots.size(), -1);
    for (const ld_unet::W8& w : u->w8_list) u->w8_of_slot[w.slot] = (long long)w.off;
    if (u->w8_bytes > 0 && hipMalloc((void**)&u->w8_base, u->w8_bytes) != hipSuccess) u->w8_base = nullptr;   // (without the copies the general kernels run)
    u->ln_fold = gemm_ln_fold_available();
    if (u->ln_fold && u->fold_bytes > 0 && hipMalloc((void**)&u->fold_base, u->fold_bytes) != hipSuccess) {
        u->pt.destroy();
        delete u;
        return LD_ERR_HIP;
    }
    if (hipMalloc((void**)&u->sync_ws, LD_SYNC_INTS * sizeof(int)) != hipSuccess || hipMemset(u->sync_ws, 0, LD_SYNC_INTS * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&u->log_sigmas, 1000 * sizeof(float)) != hipSuccess ||
        hipMemcpy(u->log_sigmas, ls.data(), 1000 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        u->pt.destroy();
        delete u;
        return LD_ERR_HIP;
    }
    *out = u;
    return LD_OK;
}

void ld_unet_destroy(ld_unet* u) {
    if (u == nullptr) return;
    u->pt.destroy();
    u->timing.destroy();
    if (u->ws_base) (void)hipFree(u->ws_base);
    if (u->log_sigmas) (void)hipFree(u->log_sigmas);
    if (u->sync_ws) (void)hipFree(u->sync_ws);
    if (u->w8_base) (void)hipFree(u->w8_base);
    if (u->fold_base) (void)hipFree(u->fold_base);
    delete u;
}

int ld_unet_param_count(const ld_unet* u) { return u ? (int)u->pt.slots.size() : 0; }

int ld_unet_param_info(const ld_unet* u, int i, const char** name, int* ndim, int64_t shape[4]) {
    if (u == nullptr || i < 0 || i >= (int)u->pt.slots.size()) return LD_ERR_ARG;
    const ParamSlot& s = u->pt.slots[i];
    if (name) *name = s.name.c_str();
    if (ndim) *ndim = s.ndim;
    if (shape)
        for (int k = 0; k < 4; ++k) shape[k] = s.shape[k];
    return LD_OK;
}

int ld_unet_load_param(ld_unet* u, const char* name, const void* src, int dtype, void* stream) {
    if (u == nullptr || name == nullptr) return LD_ERR_ARG;
    u->fold_dirty = true;   // the LN-folded copies are re-derived at the next forward
    return u->pt.load(name, src, dtype, (hipStream_t)stream);
}

size_t ld_unet_workspace_bytes(const ld_unet* u) { return u ? u->ws_bytes : 0; }
size_t ld_unet_weight_bytes(const ld_unet* u) { return u ? u->pt.bytes + (u->fold_base ? u->fold_bytes : 0) + (u->w8_base ? u->w8_bytes : 0) : 0; }

int ld_unet_reserve(ld_unet* u, int max_n, int max_h, int max_w, int max_tok) {
    if (u == nullptr || max_n < 1 || max_h < 1 || max_w < 1 || max_tok < 1) return LD_ERR_ARG;
    if (u->ws_base) {
        (void)hipFree(u->ws_base);
        u->ws_base = nullptr;
    }
    // plan: dry-run the executor to find the activation peak
    u->arena = Arena();
    u->plan_n = u->plan_h = u->plan_w = u->plan_pair = 0;
    u->ctx_tok = max_tok;
    u->ctx_tpad = (max_tok + 7) & ~7;
    u->ctx_k.assign(u->st.size(), nullptr);
    u->ctx_vt.assign(u->st.size(), nullptr);
    size_t peak = 0;
    int st = run_forward(u, true, nullptr, nullptr, nullptr, max_n, max_h, max_w, 0, nullptr, &peak);
    if (st != LD_OK) return st;
    if (!(max_n & 1)) {   // the CFG-pair route (ld_unet_forward_pair) keeps the duplicated inputs in the arena as well
        size_t peak2 = 0;
        st = run_forward(u, true, nullptr, nullptr, nullptr, max_n, max_h, max_w, 0, nullptr, &peak2, true);
        if (st != LD_OK) return st;
        if (peak2 > peak) peak = peak2;
    }
    const size_t act = (peak + 4095) / 4096 * 4096;
    const size_t tpad = (size_t)u->ctx_tpad;
    size_t ctxb = ((size_t)max_n * tpad * u->cfg.context_dim * sizeof(half_t) + 255) / 256 * 256;
    for (const StW& s : u->st) ctxb += 2 * (((size_t)max_n * tpad * s.c * sizeof(half_t) + 255) / 256 * 256);
    u->splitk_bytes = (size_t)96 << 20;
    u->ws_bytes = act + ctxb + u->splitk_bytes + 4096;
    if (hipMalloc((void**)&u->ws_base, u->ws_bytes) != hipSuccess) {
        u->ws_base = nullptr;
        return LD_ERR_HIP;
    }
    char* p = u->ws_base;
    u->arena.base = p;
    u->arena.cap = act;
    u->arena.off = u->arena.peak = 0;
    p += act;
    u->splitk_ws = reinterpret_cast<float*>(p);
    p += u->splitk_bytes;
    u->ctx16 = reinterpret_cast<half_t*>(p);
    p += ((size_t)max_n * tpad * u->cfg.context_dim * sizeof(half_t) + 255) / 256 * 256;
    for (size_t i = 0; i < u->st.size(); ++i) {
        const size_t b = ((size_t)max_n * tpad * u->st[i].c * sizeof(half_t) + 255) / 256 * 256;
        u->ctx_k[i] = reinterpret_cast<half_t*>(p);
        p += b;
        u->ctx_vt[i] = reinterpret_cast<half_t*>(p);
        p += b;
    }
    if (getenv("LD_PROFILE_DUMP") != nullptr)   // (where the allocations landed: tools/alloc_probe.py)
        fprintf(stderr, "[ld_reserve] workspace %p (%zu MiB)  weights %p  folds %p  conv8 weights %p\n", (void*)u->ws_base, u->ws_bytes >> 20, (void*)u->pt.base,
                (void*)u->fold_base, (void*)u->w8_base);
    u->max_n = max_n;
    u->max_h = max_h;
    u->max_w = max_w;
    u->max_tok = max_tok;
    u->ctx_n = 0;
    return LD_OK;
}

int ld_unet_set_context(ld_unet* u, const void* ctx, int dtype, int n, int tokens, void* stream_) {
    if (u == nullptr || ctx == nullptr) return LD_ERR_ARG;
    if (u->ws_base == nullptr || !u->pt.all_loaded()) return LD_ERR_STATE;
    if (n < 1 || n > u->max_n || tokens < 1 || tokens > u->max_tok) return LD_ERR_SHAPE;
    hipStream_t stream = (hipStream_t)stream_;
    const int D = u->cfg.context_dim, Tp = (tokens + 7) & ~7;
    int st = ctx_pad_launch(ctx, dtype == LD_F32, n, tokens, Tp, D, u->ctx16, stream);
    if (st != LD_OK) return st;
    for (size_t i = 0; i < u->st.size(); ++i) {
        const StW& s = u->st[i];
        GemmParams k;   // K = ctx · Wk^T : [n*Tp][C]
        k.A = u->ctx16; k.lda = D;
        k.W = u->pt.ptr(s.k2_w); k.ldw = D;
        k.M = n * Tp; k.N = s.c; k.K = D;
        k.C = u->ctx_k[i]; k.ldc = s.c;
        st = gemm_launch(k, stream);
        if (st != LD_OK) return st;
        GemmParams v;   // V^T[b] = Wv · ctx_b^T : [C][Tp]
        v.A = u->pt.ptr(s.v2_w); v.lda = D; v.sA = 0;
        v.W = u->ctx16; v.ldw = D; v.sW = (long long)Tp * D;
        v.M = s.c; v.N = Tp; v.K = D; v.batch = n;
        v.C = u->ctx_vt[i]; v.ldc = Tp; v.sC = (long long)s.c * Tp;
        st = gemm_launch(v, stream);
        if (st != LD_OK) return st;
    }
    u->ctx_n = n;
    u->ctx_tok = tokens;
    u->ctx_tpad = Tp;
    return LD_OK;
}

static int forward_checked(ld_unet* u, const float* x, const float* sigma, float* out, int n, int h, int w, int eps_only, void* stream, bool pair) {
    if (u == nullptr || x == nullptr || sigma == nullptr || out == nullptr) return LD_ERR_ARG;
    if (u->ws_base == nullptr || !u->pt.all_loaded() || u->ctx_n == 0) return LD_ERR_STATE;
    if (n != u->ctx_n) return LD_ERR_SHAPE;
    if (n > u->max_n || h < 1 || w < 1) return LD_ERR_SHAPE;
    int st = LD_OK;
    if (n != u->plan_n || h != u->plan_h || w != u->plan_w || (int)pair != u->plan_pair) {   // new shape: plan it on the host before touching the GPU
        size_t peak = 0;
        st = run_forward(u, true, nullptr, nullptr, nullptr, n, h, w, eps_only, nullptr, &peak, pair);
        if (st != LD_OK) return st;
        if (peak > u->arena.cap) return LD_ERR_SHAPE;
        u->plan_n = n;
        u->plan_h = h;
        u->plan_w = w;
        u->plan_pair = (int)pair;
    }
    return run_forward(u, false, x, sigma, out, n, h, w, eps_only, (hipStream_t)stream, nullptr, pair);
}

int ld_unet_forward(ld_unet* u, const float* x, const float* sigma, float* out, int n, int h, int w, int eps_only, void* stream) {
    return forward_checked(u, x, sigma, out, n, h, w, eps_only, stream, false);
}

int ld_unet_forward_pair(ld_unet* u, const float* x, const float* sigma, float* out, int nb, int h, int w, void* stream) {
    if (nb < 1) return LD_ERR_SHAPE;
    return forward_checked(u, x, sigma, out, 2 * nb, h, w, 0, stream, true);
}

static int profile_any(ld_unet* u, const float* x, const float* sigma, float* out, int n, int h, int w, void* stream, double ms[6], double flops[6],
                       int launches[6], bool pair);

int ld_unet_profile(ld_unet* u, const float* x, const float* sigma, float* out, int n, int h, int w, void* stream, double ms[6],
                    double flops[6], int launches[6]) {
    return profile_any(u, x, sigma, out, n, h, w, stream, ms, flops, launches, false);
}

int ld_unet_profile_pair(ld_unet* u, const float* x, const float* sigma, float* out, int nb, int h, int w, void* stream, double ms[6],
                         double flops[6], int launches[6]) {
    return profile_any(u, x, sigma, out, 2 * nb, h, w, stream, ms, flops, launches, true);
}

static int profile_any(ld_unet* u, const float* x, const float* sigma, float* out, int n, int h, int w, void* stream, double ms[6], double flops[6],
                       int launches[6], bool pair) {
    if (u == nullptr || ms == nullptr || flops == nullptr || launches == nullptr) return LD_ERR_ARG;
    u->want_timing = true;
    int st = forward_checked(u, x, sigma, out, n, h, w, 0, stream, pair);
    u->want_timing = false;
    if (st != LD_OK) return st;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return LD_ERR_HIP;
    u->timing.collect();
    for (int i = 0; i < KC_COUNT; ++i) {
        ms[i] = u->timing.ms[i];
        flops[i] = u->timing.flops[i];
        launches[i] = u->timing.launches[i];
    }
    return LD_OK;
}

int ld_unet_profile_kernels(const ld_unet* u, char* buf, size_t buf_bytes) {
    if (u == nullptr || buf == nullptr || buf_bytes == 0) return LD_ERR_ARG;
    size_t off = 0;
    buf[0] = 0;
    for (const auto& kv : u->timing.per_kernel) {
        const int w = snprintf(buf + off, buf_bytes - off, "%s\t%d\t%.6f\t%.0f\n", kv.first.c_str(), kv.second.launches, kv.second.ms, kv.second.flops);
        if (w < 0 || (size_t)w >= buf_bytes - off) return LD_ERR_ARG;   // buffer too small
        off += (size_t)w;
    }
    return LD_OK;
}

int ld_unet_profile_launches(const ld_unet* u, char* buf, size_t buf_bytes) {
    if (u == nullptr) return LD_ERR_ARG;
    return u->timing.format_launches(buf, buf_bytes);
}

int ld_unet_last_launches(const ld_unet* u) { return u ? u->last_launches : 0; }
double ld_unet_last_flops(const ld_unet* u) { return u ? u->last_flops : 0.0; }

}  // extern "C"
