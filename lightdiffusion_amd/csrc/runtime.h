// Host-side runtime shared by the UNet and VAE executors: resident weight table (checkpoint names -> repacked
// fp16 device tensors), a bump arena for activations with mark/release (stack discipline, deterministic addresses
// so a whole forward can be captured into a hipGraph), and an Exec context that counts launches / FLOPs and
// doubles as a dry-run planner (sizes the arena without touching the GPU).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"

enum ParamKind { PK_VEC = 0, PK_MAT = 1, PK_CONV3 = 2, PK_GEGLU_W = 3, PK_GEGLU_B = 4 };

struct ParamSlot {
    std::string name;
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    int kind = PK_VEC;
    int geglu_bn = 0;
    size_t elems = 0, offset = 0;
    bool loaded = false;
};

struct ParamTable {
    std::vector<ParamSlot> slots;
    std::unordered_map<std::string, int> index;
    size_t bytes = 0;
    char* base = nullptr;

    int add(const std::string& name, int kind, std::initializer_list<int64_t> shape, size_t align = 256, int geglu_bn = 0) {
        ParamSlot s;
        s.name = name;
        s.kind = kind;
        s.geglu_bn = geglu_bn;
        s.ndim = (int)shape.size();
        s.elems = 1;
        int i = 0;
        for (int64_t d : shape) {
            s.shape[i++] = d;
            s.elems *= (size_t)d;
        }
        bytes = (bytes + align - 1) / align * align;
        s.offset = bytes;
        bytes += s.elems * sizeof(half_t);
        index[name] = (int)slots.size();
        slots.push_back(s);
        return (int)slots.size() - 1;
    }
    int finalize() {
        bytes = (bytes + 255) / 256 * 256;
        if (hipMalloc((void**)&base, bytes + 256) != hipSuccess) return LD_ERR_HIP;
        return LD_OK;
    }
    void destroy() {
        if (base) (void)hipFree(base);
        base = nullptr;
    }
    half_t* ptr(int slot) const { return slot < 0 ? nullptr : reinterpret_cast<half_t*>(base + slots[slot].offset); }
    bool all_loaded() const {
        for (const auto& s : slots)
            if (!s.loaded) return false;
        return true;
    }
    int load(const char* name, const void* src, int dtype, hipStream_t stream) {
        auto it = index.find(name);
        if (it == index.end() || src == nullptr || base == nullptr) return LD_ERR_ARG;
        ParamSlot& s = slots[it->second];
        half_t* dst = ptr(it->second);
        int st;
        const int f32 = dtype == 1;
        switch (s.kind) {
            case PK_CONV3: st = repack_conv3x3_launch(src, f32, (int)s.shape[0], (int)s.shape[1], dst, stream); break;
            case PK_GEGLU_W: st = repack_rows_launch(src, f32, (int)s.shape[0], (int)s.shape[1], dst, s.geglu_bn, stream); break;
            case PK_GEGLU_B: st = repack_rows_launch(src, f32, (int)s.shape[0], 1, dst, s.geglu_bn, stream); break;
            case PK_MAT: st = repack_rows_launch(src, f32, (int)s.shape[0], (int)(s.elems / s.shape[0]), dst, 0, stream); break;
            default: st = repack_rows_launch(src, f32, 1, (int)s.elems, dst, 0, stream); break;
        }
        if (st == LD_OK) s.loaded = true;
        return st;
    }
};

struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    void* alloc(size_t bytes) {
        off = (off + 255) / 256 * 256;
        void* p = base + off;
        off += bytes;
        if (off > peak) peak = off;
        return p;
    }
    half_t* halfs(size_t n) { return reinterpret_cast<half_t*>(alloc(n * sizeof(half_t))); }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

// kernel classes for the per-class timing mode (ld_unet_profile): HIP events recorded on the launch stream
enum KClass { KC_CONV3 = 0, KC_GEMM = 1, KC_ATTN = 2, KC_GNORM = 3, KC_LNORM = 4, KC_MISC = 5, KC_COUNT = 6 };

struct Timing {
    std::vector<hipEvent_t> ev;     // pool, pairs (start, stop)
    std::vector<int> cls;
    std::vector<std::string> desc;   // per launch, only when LD_PROFILE_DUMP is set
    std::vector<const char*> kname;  // per launch: the kernel instantiation that ran (static strings)
    std::vector<double> kflops;      // per launch: algorithmic FLOPs
    struct PerKernel {
        double ms = 0, flops = 0;
        int launches = 0;
    };
    std::vector<std::pair<std::string, PerKernel>> per_kernel;   // filled by collect(), in first-seen order
    struct Launch {                  // one record per timed launch (ld_*_profile_launches)
        const char* what = "";
        long long a = 0, b = 0, d = 0, e = 0;
        float us = 0.f;
    };
    std::vector<Launch> rec;
    bool verbose = false;
    size_t used = 0;
    double ms[KC_COUNT] = {0}, flops[KC_COUNT] = {0};
    int launches[KC_COUNT] = {0};
    hipEvent_t next() {
        if (used == ev.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            ev.push_back(e);
        }
        return ev[used++];
    }
    void reset() {
        used = 0;
        cls.clear();
        desc.clear();
        kname.clear();
        kflops.clear();
        per_kernel.clear();
        rec.clear();
        verbose = getenv("LD_PROFILE_DUMP") != nullptr;
        for (int i = 0; i < KC_COUNT; ++i) ms[i] = flops[i] = 0, launches[i] = 0;
    }
    void collect() {
        for (size_t i = 0; i + 1 < used; i += 2) {
            float t = 0.f;
            (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
            ms[cls[i / 2]] += t;
            const char* kn = i / 2 < kname.size() && kname[i / 2] ? kname[i / 2] : "?";
            size_t j = 0;
            while (j < per_kernel.size() && per_kernel[j].first != kn) ++j;
            if (j == per_kernel.size()) per_kernel.push_back({kn, PerKernel()});
            per_kernel[j].second.ms += t;
            per_kernel[j].second.flops += i / 2 < kflops.size() ? kflops[i / 2] : 0.0;
            per_kernel[j].second.launches += 1;
            if (i / 2 < rec.size()) rec[i / 2].us = t * 1e3f;
            if (verbose && i / 2 < desc.size()) fprintf(stderr, "[ld_profile] %8.1f us  %s  %s\n", t * 1e3, desc[i / 2].c_str(), kn);
        }
    }
    void destroy() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
    }
    // "what\ta\tb\tc\td\tflops\tmicroseconds\tkernel\n" per timed launch of the last profiled run, in launch order
    int format_launches(char* buf, size_t buf_bytes) const {
        if (buf == nullptr || buf_bytes == 0) return LD_ERR_ARG;
        size_t off = 0;
        buf[0] = 0;
        for (size_t i = 0; i < rec.size(); ++i) {
            const Launch& r = rec[i];
            const int w = snprintf(buf + off, buf_bytes - off, "%s\t%lld\t%lld\t%lld\t%lld\t%.0f\t%.2f\t%s\n", r.what, r.a, r.b, r.d, r.e,
                                   i < kflops.size() ? kflops[i] : 0.0, r.us, i < kname.size() && kname[i] ? kname[i] : "?");
            if (w < 0 || (size_t)w >= buf_bytes - off) return LD_ERR_ARG;   // buffer too small
            off += (size_t)w;
        }
        return LD_OK;
    }
};

struct Exec {
    Timing* timing = nullptr;
    hipStream_t stream = nullptr;
    bool dry = false;
    Arena* arena = nullptr;
    int status = LD_OK;
    int launches = 0;
    double flops = 0.0;
    float* splitk_ws = nullptr;
    size_t splitk_bytes = 0;
    int ab_flags = 0;                // A/B build only (ld_debug_unet_flags): 4 = row-resident convolution (conv8.hip) off
    int* sync_ws = nullptr;          // LD_SYNC_INTS zeroed ints for the in-launch reductions (gemm.h GemmParams::sync); null: those kernels are not used

    void note(int st) {
        if (st != LD_OK && status == LD_OK) status = st;
    }
    bool overflow() const { return !dry && arena->peak > arena->cap; }

    void t_begin(int c, double fl, int nl, const char* what = "", long long a = 0, long long b = 0, long long d = 0, long long e = 0) {
        if (timing == nullptr || dry) return;
        timing->cls.push_back(c);
        if (timing->verbose) {
            char buf[160];
            snprintf(buf, sizeof buf, "%-10s %8lld %6lld %6lld %4lld  %.2f GFLOP", what, a, b, d, e, fl * 1e-9);
            timing->desc.push_back(buf);
        }
        timing->flops[c] += fl;
        timing->launches[c] += nl;
        timing->kflops.push_back(fl);
        Timing::Launch r;
        r.what = what;
        r.a = a; r.b = b; r.d = d; r.e = e;
        timing->rec.push_back(r);
        (void)hipEventRecord(timing->next(), stream);
    }
    void t_end(const char* kernel_name = "misc") {
        if (timing == nullptr || dry) return;
        (void)hipEventRecord(timing->next(), stream);
        timing->kname.push_back(kernel_name);
    }

    void gemm(GemmParams p) {
        const double fl = 2.0 * p.M * (double)p.N * p.K * p.batch;
        flops += fl;
        if (p.batch == 1) {
            p.partial = splitk_ws;
            p.partial_bytes = splitk_bytes;
            p.sync = sync_ws;
        }
        if (ab_flags & 4) p.W8 = nullptr;
        launches += 1;
        if (dry || status != LD_OK) return;
        t_begin(p.conv && p.ksize == 3 ? KC_CONV3 : KC_GEMM, fl, 1, p.conv ? (p.ksize == 3 ? "conv3" : "conv1") : (p.act == 2 ? "geglu" : "gemm"),
                p.M, p.N, p.K, p.batch);
        note(gemm_launch(p, stream));
        t_end(gemm_last_kernel_name());
    }
    // would gemm() run this 3x3 convolution on a kernel that takes a second K segment (gemm.h S1 / S2)?
    bool conv_takes_skip_segment(GemmParams p) const {
        p.partial = splitk_ws;
        p.partial_bytes = splitk_bytes;
        p.sync = sync_ws;
        if (ab_flags & 4) p.W8 = nullptr;
        return gemm_conv_takes_skip_segment(p);
    }
    // `ready` / `ready_P`: partial statistics the producer already wrote (gemm.h gn_part; ready_P pixel chunks per image): the statistics
    // launch is skipped
    void groupnorm(const half_t* x1, int C1, const half_t* x2, int C2, int n, int HW, const half_t* g, const half_t* b, float eps,
                   int silu, half_t* y, float* ready = nullptr, int ready_P = 0) {
        const size_t m = arena->mark();
        if (ready_P <= 0) ready = nullptr;
        float* ws = ready != nullptr ? ready : reinterpret_cast<float*>(arena->alloc(groupnorm_workspace_bytes(n, HW)));
        const int nl = ready != nullptr ? 1 : 2;
        launches += nl;
        t_begin(KC_GNORM, 0.0, nl, "groupnorm", n, HW, C1 + C2, silu);
        if (!dry && status == LD_OK) note(groupnorm_launch(x1, C1, x2, C2, n, HW, g, b, eps, silu, y, ws, stream, ready != nullptr ? ready_P : 0));
        t_end(ready != nullptr ? "gn_apply_kernel" : "gn_stats_kernel+gn_apply_kernel");
        arena->release(m);
    }
    // GroupNorm(32) + SiLU feeding a 3x3 convolution.  When the convolution runs on the halo-tile kernel the normalisation is fused
    // into its A operand: only the statistics pass + a tiny finalize run here (scale / shift per image and channel), the conv reads the
    // RAW tensor(s) — no normalised copy is written or read back.  Otherwise: the two-pass GroupNorm into `g` (caller-provided), then the conv.
    // `p`: the convolution with A / A2 = the RAW sources; returns through p.C as usual.
    // `ready` / `ready_P`: GroupNorm partial statistics of the input that its producer already wrote (see groupnorm)
    void gn_silu_conv(GemmParams p, int n_img, int HW, const half_t* gamma, const half_t* beta, float eps, half_t* g, float* ready = nullptr, int ready_P = 0) {
        p.partial = splitk_ws;
        p.partial_bytes = splitk_bytes;
        p.sync = sync_ws;
        if (ready_P <= 0) ready = nullptr;
        if (ab_flags & 4) p.W8 = nullptr;
        if (gemm_conv_fuses_groupnorm(p)) {
            const int C = p.C1 + p.C2;
            const size_t m = arena->mark();
            float* ws = ready != nullptr ? ready : reinterpret_cast<float*>(arena->alloc(groupnorm_workspace_bytes(n_img, HW)));
            float* scale = reinterpret_cast<float*>(arena->alloc((size_t)n_img * C * sizeof(float)));
            float* shift = reinterpret_cast<float*>(arena->alloc((size_t)n_img * C * sizeof(float)));
            const int nl = ready != nullptr ? 1 : 2;
            launches += nl;
            t_begin(KC_GNORM, 0.0, nl, "gn_stats", n_img, HW, C, 1);
            if (!dry && status == LD_OK)
                note(groupnorm_scale_shift_launch(p.A, p.C1, p.A2, p.C2, n_img, HW, gamma, beta, eps, ws, scale, shift, stream, ready != nullptr ? ready_P : 0));
            t_end(ready != nullptr ? "gn_finalize_kernel" : "gn_stats_kernel+gn_finalize_kernel");
            p.gn_scale = scale;
            p.gn_shift = shift;
            p.gn_silu = 1;
            gemm(p);
            arena->release(m);
            return;
        }
        groupnorm(p.A, p.C1, p.A2, p.C2, n_img, HW, gamma, beta, eps, 1, g, ready, ready_P);
        p.A = g;
        p.A2 = nullptr;
        p.C1 = p.C1 + p.C2;
        p.C2 = 0;
        gemm(p);
    }
    void layernorm(const half_t* x, const half_t* g, const half_t* b, half_t* y, int rows, int C) {
        launches += 1;
        t_begin(KC_LNORM, 0.0, 1, "layernorm", rows, C);
        if (!dry && status == LD_OK) note(layernorm_launch(x, g, b, y, rows, C, 1e-5f, stream));
        t_end("layernorm_kernel");
    }
    void attention(const AttnParams& p) {
        const double fl = 4.0 * p.B * p.H * (double)p.Lq * p.Lk * p.d;
        flops += fl;
        launches += 1;
        t_begin(KC_ATTN, fl, 1, "attention", (long long)p.B * p.H, p.Lq, p.Lk, p.d);
        if (!dry && status == LD_OK) note(attention_launch(p, stream));
        t_end(attention_last_kernel_name());
    }
};
