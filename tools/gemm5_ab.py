"""A/B of the 256x320 staggered kernel (gemm5) against the 128x160 kernel (gemm3) on the SD1.5 UNet's batch-8 contraction shapes.
Needs the A/B build (make -C lightdiffusion_amd/csrc ab): LD_MI355X_LIB=lightdiffusion_amd/libld_mi355x_ab.so python tools/gemm5_ab.py
Interleaved rounds in ONE process (cdna guide rule 24), random data, launches replayed from a hipGraph."""
import math, os, sys
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
from lightdiffusion_amd._lib import lib, check

DEV = "cuda:0"
L = lib()
WS = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)


def graph_time(fn, reps):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps)
    return min(ts)


def lin(m, n, k, act=0, res=False):
    x = torch.randn(m, k, device=DEV, dtype=torch.float16)
    w = torch.randn(n, k, device=DEV, dtype=torch.float16) / math.sqrt(k)
    b = torch.randn(n, device=DEV, dtype=torch.float16) * 0.1
    on = n // 2 if act == 2 else n
    y = torch.empty(m, on, device=DEV, dtype=torch.float16)
    r = torch.randn(m, on, device=DEV, dtype=torch.float16) if res else None
    def fn():
        check(L.ld_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(), m, n, k, 1.0, act,
                             WS.data_ptr(), WS.numel(), torch.cuda.current_stream().cuda_stream), "lin")
    return fn, 2.0 * m * n * k, f"{'geglu' if act == 2 else 'gemm '} {m}x{n}x{k}{' +res' if res else ''}"


def conv(nimg, h, cin, cout, res=False):
    x = torch.randn(nimg, h, h, cin, device=DEV, dtype=torch.float16)
    wt = torch.randn(cout, 9 * cin, device=DEV, dtype=torch.float16) / math.sqrt(9 * cin)
    b = torch.randn(cout, device=DEV, dtype=torch.float16) * 0.1
    y = torch.empty(nimg, h, h, cout, device=DEV, dtype=torch.float16)
    r = torch.randn(nimg, h, h, cout, device=DEV, dtype=torch.float16) if res else None
    def fn():
        check(L.ld_op_conv(x.data_ptr(), cin, None, 0, nimg, h, h, h, h, 1, 3, wt.data_ptr(), b.data_ptr(), None, None if r is None else r.data_ptr(),
                           y.data_ptr(), cout, WS.data_ptr(), WS.numel(), torch.cuda.current_stream().cuda_stream), "conv")
    return fn, 2.0 * nimg * h * h * cout * 9 * cin, f"conv3 {nimg * h * h}x{cout}x{9 * cin}{' +res' if res else ''}"


cases = [lin(65536, 320, 320, res=True), lin(65536, 320, 320), lin(65536, 640, 320), lin(65536, 2560, 320, act=2), lin(65536, 320, 1280, res=True),
         lin(16384, 5120, 640, act=2), lin(16384, 1280, 640), lin(4096, 10240, 1280, act=2), lin(8192, 8000, 8192) if False else lin(8192, 8320, 8192),
         conv(16, 64, 320, 320, True), conv(16, 64, 640, 320), conv(16, 64, 960, 320), conv(16, 64, 640, 640), conv(16, 32, 1280, 1280),
         conv(16, 32, 640, 640), conv(16, 16, 1280, 1280)]
MODE = os.environ.get("AB_MODE", "v5")      # "v5": v3 vs (v5 + v6 + v7);  "v6": v5 (no halo kernel) vs v6;  "v7": without / with the row-panel kernel
OFF = {"v5": 11 + 16, "v6": 2, "v7": 8 + 16}[MODE]      # ld_debug_gemm_no_v5 bits: 1 = no v5, 2 = no v6, 4 = no fused GN, 8 = no v7, 16 = GEGLU not on v7
ON = int(os.environ.get("AB_ON", "0"))        # bits of the "on" arm
print(f"{'shape':34s} {'off us':>9s} {'TF/s':>7s} {'on us':>9s} {'TF/s':>7s}  off/on   ({MODE})")
for fn, fl, name in cases:
    reps = max(3, min(50, int(2e-3 / (fl / 0.8e15)) + 1))
    t = {ON: [], OFF: []}
    for _ in range(3):
        for off in (OFF, ON):
            L.ld_debug_gemm_no_v5(off)
            t[off].append(graph_time(fn, reps))
    L.ld_debug_gemm_no_v5(0)
    a, b = min(t[OFF]), min(t[ON])
    print(f"{name:34s} {a * 1e3:9.1f} {fl / a / 1e9:7.0f} {b * 1e3:9.1f} {fl / b / 1e9:7.0f}  {a / b:5.2f}", flush=True)
