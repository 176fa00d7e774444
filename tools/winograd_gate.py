#!/usr/bin/env python3
"""Round 6 gate: Winograd F(2x2, 3x3) on MFMA against the product's 3x3 convolution route, shape by shape.

For each of the five shapes that carry the conv time (UNet 65536x320x320, 16384x640x640, 4096x1280x1280; VAE 512 ch at 128-pixel rows,
256 ch at 256-pixel rows) prints: microseconds of the product route (`ld_op_conv`: conv6 / gemm3 + split-K, whatever the dispatcher takes),
microseconds of the stand-alone fused Winograd kernel (tools/micro/winograd_f2x2.hip), and both results' rel-L2 against fp32
`F.conv2d` on the same fp16-rounded inputs.  Build first:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC tools/micro/winograd_f2x2.hip -o tools/micro/libwinograd_gate.so
"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def transform_weight(w_oihw: torch.Tensor) -> torch.Tensor:
    """[O,I,3,3] -> U [16][O][I] fp16: G g G^T in fp32 (fp64 here), rounded to fp16 ONCE."""
    g = w_oihw.double()
    u = torch.einsum("ik,ockl,jl->ijoc", G.to(g.device), g, G.to(g.device))
    return u.reshape(16, g.shape[0], g.shape[1]).to(torch.float16).contiguous()


def pack_weight(U: torch.Tensor) -> torch.Tensor:
    """U [16][O][I] -> the kernel's operand order [O/64][I/32][16][hh 2][s 2][kg 2][n 32][8]:
    cout = cb*64 + hh*32 + n, cin = kk*32 + kg*16 + s*8 + e, lane = kg*32 + n (one contiguous KB per wave-instruction)."""
    P, O, I = U.shape
    u = U.view(P, O // 64, 2, 32, I // 32, 2, 2, 8)            # p, cb, hh, n, kk, kg, s, e
    return u.permute(1, 4, 0, 2, 6, 5, 3, 7).contiguous()


def winograd_host(x_nhwc: torch.Tensor, w_oihw: torch.Tensor) -> torch.Tensor:
    """The same algebra on the host in fp64 (checks the matrices and the tile / halo conventions, tiny shapes only)."""
    n, h, w, c = x_nhwc.shape
    o = w_oihw.shape[0]
    xp = F.pad(x_nhwc.double().permute(0, 3, 1, 2), (1, 1, 1, 1))
    u = torch.einsum("ik,ockl,jl->ijoc", G, w_oihw.double(), G)
    y = torch.zeros(n, o, h, w, dtype=torch.float64)
    for ty in range(h // 2):
        for tx in range(w // 2):
            d = xp[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            v = torch.einsum("ik,nckl,jl->nijc", BT, d, BT)
            m = torch.einsum("nijc,ijoc->nijo", v, u)
            y[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = torch.einsum("ai,nijo,bj->noab", AT, m, AT)
    return y.permute(0, 2, 3, 1)


def conv_ref(x_hwc: torch.Tensor, w16: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """fp32 3x3 convolution (pad 1) of one NHWC image as nine shifted matrix products (plain fp32 GEMMs: no MIOpen search on the box)."""
    h, w, c = x_hwc.shape
    xp = F.pad(x_hwc.float().permute(2, 0, 1), (1, 1, 1, 1)).permute(1, 2, 0)
    wf = w16.float()
    out = bias.float().expand(h * w, -1).clone()
    for ky in range(3):
        for kx in range(3):
            out += xp[ky:ky + h, kx:kx + w].reshape(h * w, c) @ wf[:, :, ky, kx].t()
    return out.reshape(h, w, -1)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(3):
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best.append(1e3 * e0.elapsed_time(e1) / reps)
    return min(best), sorted(best)[1]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    # host check of the algebra (no GPU involved)
    g = torch.Generator().manual_seed(0)
    xs, ws = torch.randn(1, 4, 6, 3, generator=g), torch.randn(5, 3, 3, 3, generator=g)
    ref = F.conv2d(xs.permute(0, 3, 1, 2).double(), ws.double(), padding=1).permute(0, 2, 3, 1)
    emit(f"# host algebra check (fp64): rel-L2 {rel_l2(winograd_host(xs, ws), ref):.1e}")

    from lightdiffusion_amd import ops
    lib = C.CDLL(os.path.join(ROOT, "tools", "micro", "libwinograd_gate.so"))
    lib.wino_f2x2_conv.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
    lib.wino_f2x2_conv.restype = C.c_int
    dev = torch.device("cuda:0")
    stream = lambda: torch.cuda.current_stream().cuda_stream

    shapes = [("UNet 65536x320x320  (N=16, 64x64)", 16, 64, 64, 320, 320),
              ("UNet 16384x640x640  (N=16, 32x32)", 16, 32, 32, 640, 640),
              ("UNet 4096x1280x1280 (N=16, 16x16)", 16, 16, 16, 1280, 1280),
              ("VAE 512 ch, 128-pixel rows (b=8)", 8, 128, 128, 512, 512),
              ("VAE 256 ch, 256-pixel rows (b=8)", 8, 256, 256, 256, 256)]
    if os.environ.get("WINO_SMALL"):
        shapes = [("small 2x32x32 64->64", 2, 32, 32, 64, 64)] + shapes[:1]
    emit("# shape | GFLOP direct | product route us (PFLOP/s) | winograd us (PFLOP/s-equivalent, MFMA PFLOP/s) | speed-up | rel-L2 product | rel-L2 winograd")
    NB = 3
    for name, n, h, w, cin, cout in shapes:
        gen = torch.Generator().manual_seed(cin + h)
        xs_ = [torch.randn(n, h, w, cin, generator=gen).to(dev, torch.float16) for _ in range(NB)]
        wt = (torch.randn(cout, cin, 3, 3, generator=gen) * (9 * cin) ** -0.5).to(dev)
        bias = (torch.randn(cout, generator=gen) * 0.1).to(dev, torch.float16)
        w16 = wt.to(torch.float16)
        wp = ops.repack_conv_weight(w16)
        U = pack_weight(transform_weight(wt))
        ys = [torch.empty(n, h, w, cout, dtype=torch.float16, device=dev) for _ in range(NB)]

        def wino(i=0):
            rc = lib.wino_f2x2_conv(xs_[i % NB].data_ptr(), U.data_ptr(), bias.data_ptr(), ys[i % NB].data_ptr(), n, h, w, cin, cout, stream())
            assert rc == 0, rc

        def prod(i=0):
            return ops.conv2d(xs_[i % NB], wp, bias)

        # parity on buffer 0 (reference in fp32 on the fp16-rounded operands, in image chunks to bound memory)
        wino(0)
        yp = prod(0)
        torch.cuda.synchronize()
        num_w = num_p = den = 0.0
        for k in range(n):
            r = conv_ref(xs_[0][k], w16, bias)[None]
            num_w += float((ys[0][k:k + 1].float() - r).double().pow(2).sum())
            num_p += float((yp[k:k + 1].float() - r).double().pow(2).sum())
            den += float(r.double().pow(2).sum())
        e_w, e_p = (num_w / den) ** 0.5, (num_p / den) ** 0.5
        reps = 20
        t_p, _ = timed(prod, reps)
        t_w, _ = timed(wino, reps)
        gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
        emit(f"{name} | {gf:7.1f} | {t_p:7.1f} ({gf / t_p:.2f}) | {t_w:7.1f} ({gf / t_w:.2f}, {gf / 2.25 / t_w:.2f}) | "
             f"{t_p / t_w:.2f}x | {e_p:.2e} | {e_w:.2e}")
        if os.environ.get("WINO_ABL"):
            lib.wino_f2x2_conv_abl.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
            names = {0: "full", 1: "no MFMA", 2: "no LDS operand reads", 4: "no staging", 8: "no weight loads", 16: "no epilogue", 6: "no LDS reads, no staging",
                     14: "MFMA + epilogue only", 30: "MFMA only", 29: "LDS operand reads only", 31: "empty loop"}
            for abl, nm in names.items():
                def f(i=0, abl=abl):
                    assert lib.wino_f2x2_conv_abl(xs_[i % NB].data_ptr(), U.data_ptr(), bias.data_ptr(), ys[i % NB].data_ptr(), n, h, w, cin, cout, abl, stream()) == 0
                emit(f"    ablation {abl:2d} ({nm}): {timed(f, reps)[0]:7.1f} us")
        del xs_, ys, yp
        torch.cuda.empty_cache()
    if out_path:
        with open(out_path, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
