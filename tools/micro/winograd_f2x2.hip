// Round 6 gate: Winograd F(2x2, 3x3) on MFMA for the stride-1 3x3 convolutions, stand-alone.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          (ResBlock1 convs LD.py:5224 / 5253, VAE ResnetBlock LD.py:3531-3576)
//
// One workgroup = 8 x 8 output tiles of 2 x 2 pixels (a 16 x 16 pixel patch of one image) x 64 output channels x all 16 Winograd
// positions, 8 waves.  Wave w owns the two positions (i, 2jp), (i, 2jp + 1) with i = w & 3, jp = w >> 2, over all 64 tiles and 64 output
// channels: 2 positions x 2 tile halves x 2 channel halves = 8 accumulator tiles of 32 x 32 = 128 VGPRs (the 16 positions of a tile need
// 4x the accumulators of the direct form: 64 tiles x 64 channels fill half the CU's register file).
// Per 32-channel K step:
//   * staging (all 512 threads): the 18 x 18 pixel halo of the patch is read from global memory and COLUMN-combined on the way into LDS
//     (C_j[r][tx] = the j-th row of B^T applied along x for tile column tx: shared by all four i and both channel halves),
//   * every wave builds its B operands on the fly: V_ij = C_j[2ty + ka] +- C_j[2ty + kb]  (two ds_read_b128 + four v_pk_fma_f16 each),
//   * the transformed weights U[p][cout][cin] (fp16, transformed in fp32 and rounded once) are wave-private (a wave's positions are its
//     own), so they go global -> register in MFMA operand order, no LDS,
//   * 16 MFMAs 32x32x16 per wave.
// Epilogue: the output transform needs all 16 positions of a (tile, cout): the waves meet in LDS (fp32, two passes of 136 KB, one per
// output column b), every thread sums the eight partials of one (tile, 8-channel chunk), adds the bias and stores two pixels.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TXN = 8, TYN = 8;                  // tiles per workgroup
constexpr int HR = 2 * TYN + 2;                  // halo rows
constexpr int KC = 32;                           // channels per K step
constexpr int BLK = KC * 2;                      // bytes of one (j, r, tx) block
constexpr int JSTRIDE = HR * TXN * BLK;          // 9216
constexpr int STAGE = 4 * JSTRIDE;               // 36864
constexpr int PT = 68;                           // floats per tile row of an epilogue partial (64 + pad: conflict-free b128 stores)
constexpr int WSTRIDE = 64 * PT * 4;             // bytes of one wave's partial
constexpr int SMEM = 8 * WSTRIDE;                // 139264 >= 2 * STAGE

__device__ __forceinline__ half8 ld16(const half_t* p) { return *reinterpret_cast<const half8*>(p); }

// ABL (timing only, wrong results): 1 no MFMAs, 2 no operand reads from LDS, 4 no staging (halo loads + column combine + LDS stores),
// 8 no weight loads, 16 no epilogue
template <int ABL>
__global__ __launch_bounds__(512) void wino_f2x2_kernel(const half_t* __restrict__ x, const half_t* __restrict__ U, const half_t* __restrict__ bias,
                                                        half_t* __restrict__ y, int H, int W, int Cin, int Cout, int nsp, int G) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int bxn = W / 16, byn = H / 16;
    // Workgroup -> (spatial patch, 64-channel block).  A workgroup's output tile is only 64 channels wide (the 16 positions take the
    // accumulators), so a patch's input is read Cout / 64 times and a channel block's weights once per patch: both must come from the
    // XCD's L2.  G > 0: XCD x = blockIdx % 8 owns the patches sp = x (mod 8) and walks them in groups of G channel blocks (their weights,
    // G x 16 x 64 x Cin fp16, sized to stay in the 4 MB L2) — group outermost, patch, then block innermost: the input of a patch is
    // re-read from L2 by its G workgroups, which run together, and from memory once per group.  G == 0: plain order (block slowest).
    const int ncb = Cout / 64;
    int sp, cb;
    if (G > 0) {
        const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3, per = nsp >> 3;
        int g = l / (per * G);
        const int ng = (ncb + G - 1) / G;
        if (g > ng - 1) g = ng - 1;
        const int l2 = l - g * per * G;
        const int gs = min(G, ncb - g * G);
        sp = xcd + 8 * (l2 / gs);
        cb = g * G + l2 % gs;
    } else {
        sp = blockIdx.x % nsp;
        cb = blockIdx.x / nsp;
    }
    const int bx = sp % bxn; sp /= bxn;
    const int by = sp % byn;
    const int img = sp / byn;
    const int nk = Cin / KC;

    // ---------------- staging items: (halo row r, tile column tx, 8-channel chunk c)
    const int it0 = tid, it1 = 512 + tid;            // 576 items: the second only for tid < 64
    const bool has1 = tid < 64;
    auto item = [&](int it, int& r, int& tx, int& c) { c = it & 3; tx = (it >> 2) & 7; r = it >> 5; };
    int r0, tx0, c0, r1, tx1, c1;
    item(it0, r0, tx0, c0);
    item(has1 ? it1 : it0, r1, tx1, c1);
    const half_t* gp0[4]; bool ok0[4];
    const half_t* gp1[4]; bool ok1[4];
    auto setup = [&](int r, int tx, int c, const half_t** gp, bool* ok) {
        const int yy = by * 16 - 1 + r;
        const bool yok = yy >= 0 && yy < H;
        const int yc = min(max(yy, 0), H - 1);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int xx = bx * 16 - 1 + 2 * tx + l;
            ok[l] = yok && xx >= 0 && xx < W;
            const int xc = min(max(xx, 0), W - 1);
            gp[l] = x + ((size_t)(img * H + yc) * W + xc) * Cin + c * 8;
        }
    };
    setup(r0, tx0, c0, gp0, ok0);
    setup(r1, tx1, c1, gp1, ok1);
    const int so0 = (r0 * TXN + tx0) * BLK + ((c0 ^ ((r0 >> 1) & 3)) << 4);
    const int so1 = (r1 * TXN + tx1) * BLK + ((c1 ^ ((r1 >> 1) & 3)) << 4);
    const half8 hz = {0, 0, 0, 0, 0, 0, 0, 0};

    auto combine_store = [&](unsigned char* buf, int so, const half8* d, const bool* ok) {
        const half8 d0 = ok[0] ? d[0] : hz, d1 = ok[1] ? d[1] : hz, d2 = ok[2] ? d[2] : hz, d3 = ok[3] ? d[3] : hz;
        *reinterpret_cast<half8*>(buf + 0 * JSTRIDE + so) = d0 - d2;
        *reinterpret_cast<half8*>(buf + 1 * JSTRIDE + so) = d1 + d2;
        *reinterpret_cast<half8*>(buf + 2 * JSTRIDE + so) = d2 - d1;
        *reinterpret_cast<half8*>(buf + 3 * JSTRIDE + so) = d1 - d3;
    };

    // ---------------- wave roles
    const int wi = wv & 3, jp = wv >> 2;
    const int n = lane & 31, kg = lane >> 5;
    // B^T row i = d[ka] + sg * d[kb]
    const int ka = (wi == 0) ? 0 : (wi == 2 ? 2 : 1);
    const int kb = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const half_t sgh = (wi == 1) ? (half_t)1.f : (half_t)-1.f;
    const half8 sg = {sgh, sgh, sgh, sgh, sgh, sgh, sgh, sgh};
    int offA[2], offB[2];                          // LDS byte offsets inside plane j, substep 0 (substep 1: ^ 16)
#pragma unroll
    for (int th = 0; th < 2; ++th) {
        const int ty = th * 4 + (n >> 3), tx = n & 7;
        const int ra = 2 * ty + ka, rb = 2 * ty + kb;
        offA[th] = (ra * TXN + tx) * BLK + (((kg * 2) ^ ((ra >> 1) & 3)) << 4);
        offB[th] = (rb * TXN + tx) * BLK + (((kg * 2) ^ ((rb >> 1) & 3)) << 4);
    }
    // weights, packed in MFMA operand order (tools/winograd_gate.py: pack_weight): U[cb][kk][p][hh][s][lane][8], p = 4 i + j — one
    // wave-instruction reads one contiguous KB; a wave's 8 KB of a K step are contiguous.  (Unpacked [p][cout][cin] rows cost 70 of 172 us
    // at 65536x320x320: 32 rows x 2 16-byte pieces per instruction.)
    const half_t* ubase = U + ((size_t)cb * nk * 16 + 4 * wi + 2 * jp) * 2048 + lane * 8;
    auto uptr = [&](int kk_, int jj, int hh, int s_) { return ubase + (size_t)kk_ * 32768 + jj * 2048 + hh * 1024 + s_ * 512; };

    f32x16 acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][c][e] = 0.f;

    // ---------------- prologue: stage 0
    half8 d0[4], d1[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) d0[l] = ld16(gp0[l]);
    if (has1) {
#pragma unroll
        for (int l = 0; l < 4; ++l) d1[l] = ld16(gp1[l]);
    }
    half8 bcur[2][2], bnxt[2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) bcur[jj][hh] = ld16(uptr(0, jj, hh, 0));
    combine_store(smem, so0, d0, ok0);
    if (has1) combine_store(smem, so1, d1, ok1);
    __syncthreads();

    // Rolling schedule over half substeps h = (k16 substep s, position jj): the LDS reads of h + 1 are issued before the four MFMAs of h and
    // turned into operands behind them; the weights of substep s + 1 are issued at the top of substep s.  sched_barrier pins the order: left
    // alone hipcc sinks every load to just in front of its first use (weights waited for within a few instructions of their issue: 172 ->
    // 124 us came from packing them, the wait itself stayed).
    half8 v[2][2];                                  // operands of the half substep in flight, [jj][th]
    half8 ra[2], rb[2];
    auto issue_reads = [&](const unsigned char* buf, int s_, int jj) {
        const unsigned char* pl = buf + (2 * jp + jj) * JSTRIDE;
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            if (!(ABL & 2)) {
                ra[th] = *reinterpret_cast<const half8*>(pl + (offA[th] ^ (s_ << 4)));
                rb[th] = *reinterpret_cast<const half8*>(pl + (offB[th] ^ (s_ << 4)));
            } else { ra[th] = bcur[jj][th]; rb[th] = bcur[th][jj]; asm volatile("" : "+v"(ra[th]), "+v"(rb[th])); }
        }
    };
    auto finish_reads = [&](int jj) {
#pragma unroll
        for (int th = 0; th < 2; ++th) v[jj][th] = rb[th] * sg + ra[th];
    };
    issue_reads(smem, 0, 0);
    finish_reads(0);
    for (int kk = 0; kk < nk; ++kk) {
        unsigned char* cur = smem + (kk & 1) * STAGE;
        unsigned char* nxt = smem + ((kk + 1) & 1) * STAGE;
        const bool more = kk + 1 < nk;
        const int kn = more ? (kk + 1) * KC : kk * KC;      // clamped: the last step re-loads its own (unused) data
        const int kkn = more ? kk + 1 : kk;
        if (!(ABL & 4)) {
#pragma unroll
            for (int l = 0; l < 4; ++l) d0[l] = ld16(gp0[l] + kn);
            if (has1) {
#pragma unroll
                for (int l = 0; l < 4; ++l) d1[l] = ld16(gp1[l] + kn);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if (jj == 0) {
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            if (!(ABL & 8)) bnxt[j2][hh] = ld16(s == 0 ? uptr(kk, j2, hh, 1) : uptr(kkn, j2, hh, 0));   // next substep's weights
                            else { bnxt[j2][hh] = bcur[j2][hh]; asm volatile("" : "+v"(bnxt[j2][hh])); }
                        }
                }
                // operands of the next half substep (the first one of the next K step is read behind the barrier)
                if (jj == 0) issue_reads(cur, s, 1);
                else if (s == 0) issue_reads(cur, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int th = 0; th < 2; ++th)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        if (!(ABL & 1)) acc[jj][th][hh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bcur[jj][hh], v[jj][th], acc[jj][th][hh], 0, 0, 0);
                        else asm volatile("" : "+v"(acc[jj][th][hh]) : "v"(bcur[jj][hh]), "v"(v[jj][th]));
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (jj == 0) finish_reads(1);
                else if (s == 0) finish_reads(0);
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) bcur[jj][hh] = bnxt[jj][hh];
        }
        if (more && !(ABL & 4)) {
            combine_store(nxt, so0, d0, ok0);
            if (has1) combine_store(nxt, so1, d1, ok1);
        }
        __syncthreads();
        if (more) {
            issue_reads(nxt, 0, 0);
            finish_reads(0);
        }
    }

    // ---------------- epilogue: output transform through LDS, one pass per output column b
    if (ABL & 16) {
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[a >> 2][(a >> 1) & 1][a & 1][e];
        if (t == 12345.678f) y[tid] = (half_t)t;
        return;
    }
    const int tile_e = tid >> 3, ch_e = tid & 7;
    const int ty_e = tile_e >> 3, tx_e = tile_e & 7;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = bias ? (float)bias[cb * 64 + ch_e * 8 + e] : 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        // this wave's share of R[b] = sum_j At[b][j] M[i][j]:  jp = 0: b0: M0 + M1, b1: M1;   jp = 1: b0: M2, b1: -M2 - M3
        unsigned char* mine = smem + wv * WSTRIDE;
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m0 = acc[0][th][hh][g * 4 + e], m1 = acc[1][th][hh][g * 4 + e];
                        o[e] = (jp == 0) ? (b == 0 ? m0 + m1 : m1) : (b == 0 ? m0 : -m0 - m1);
                    }
                    const int tile = th * 32 + n, co = hh * 32 + g * 8 + kg * 4;
                    *reinterpret_cast<f32x4*>(mine + (tile * PT + co) * 4) = o;
                }
        __syncthreads();
        float y0[8], y1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { y0[e] = bv[e]; y1[e] = bv[e]; }
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const unsigned char* p = smem + w * WSTRIDE + (tile_e * PT + ch_e * 8) * 4;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(p), q1 = *reinterpret_cast<const f32x4*>(p + 16);
            const int i = w & 3;
            // A^T: row 0 = [1 1 1 0], row 1 = [0 1 -1 -1]
            const float s0 = (i < 3) ? 1.f : 0.f, s1 = (i == 0) ? 0.f : (i == 1 ? 1.f : -1.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y0[e] += s0 * q0[e]; y0[4 + e] += s0 * q1[e];
                y1[e] += s1 * q0[e]; y1[4 + e] += s1 * q1[e];
            }
        }
        half8 h0, h1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { h0[e] = (half_t)y0[e]; h1[e] = (half_t)y1[e]; }
        const int oy = by * 16 + 2 * ty_e, ox = bx * 16 + 2 * tx_e + b;
        half_t* yp = y + ((size_t)(img * H + oy) * W + ox) * Cout + cb * 64 + ch_e * 8;
        *reinterpret_cast<half8*>(yp) = h0;
        *reinterpret_cast<half8*>(yp + (size_t)W * Cout) = h1;
        __syncthreads();
    }
}
}  // namespace

// channel blocks per group of the XCD-blocked order: as many as keep the group's weights within ~3.5 MB of the 4 MB L2 (WINO_G overrides;
// 0 = plain order); needs the patch count to be a multiple of the 8 XCDs
static int group_size(int nsp, int Cin, int Cout) {
    const char* e = getenv("WINO_G");
    if (e) return atoi(e);
    if (nsp % 8) return 0;
    const long long per_cb = 16LL * 64 * Cin * 2;
    long long g = (3584LL << 10) / per_cb;
    const int ncb = Cout / 64;
    return (int)(g < 1 ? 1 : (g > ncb ? ncb : g));
}

// x [N,H,W,Cin] fp16 NHWC, U [16][Cout][Cin] fp16 (G g G^T, position p = 4 i + j), bias [Cout] fp16 or null, y [N,H,W,Cout] fp16.
// H, W multiples of 16; Cin multiple of 32; Cout multiple of 64.  Returns 0, or -1 on a shape it does not take.
extern "C" int wino_f2x2_conv(const void* x, const void* U, const void* bias, void* y, int N, int H, int W, int Cin, int Cout, void* stream) {
    if (H % 16 || W % 16 || Cin % KC || Cout % 64 || N <= 0) return -1;
    const int nsp = N * (H / 16) * (W / 16);
    dim3 grid((unsigned)(nsp * (Cout / 64)));
    hipLaunchKernelGGL(wino_f2x2_kernel<0>, grid, dim3(512), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)U, (const half_t*)bias,
                       (half_t*)y, H, W, Cin, Cout, nsp, group_size(nsp, Cin, Cout));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// timing-only ablations of the same launch (results are wrong by construction)
extern "C" int wino_f2x2_conv_abl(const void* x, const void* U, const void* bias, void* y, int N, int H, int W, int Cin, int Cout, int abl, void* stream) {
    if (H % 16 || W % 16 || Cin % KC || Cout % 64 || N <= 0) return -1;
    const int nsp = N * (H / 16) * (W / 16);
    dim3 grid((unsigned)(nsp * (Cout / 64)));
    const int G = group_size(nsp, Cin, Cout);
#define ABL_CASE(v) case v: hipLaunchKernelGGL(wino_f2x2_kernel<v>, grid, dim3(512), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)U, \
                                               (const half_t*)bias, (half_t*)y, H, W, Cin, Cout, nsp, G); break;
    switch (abl) {
        ABL_CASE(0) ABL_CASE(1) ABL_CASE(2) ABL_CASE(4) ABL_CASE(8) ABL_CASE(16) ABL_CASE(6) ABL_CASE(14) ABL_CASE(30) ABL_CASE(31) ABL_CASE(29)
        default: return -3;
    }
#undef ABL_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
