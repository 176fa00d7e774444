"""What the strip epilogue of the 256 x 320 kernels (conv6 halo convolution, gemm5) costs per launch (A/B build; wrong results on purpose, timing only)."""
import math, os, sys
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
src = open(os.path.join("tools", "gemm5_ab.py")).read().split("cases = [")[0]
exec(src)
cases = [conv(16, 64, 320, 320, True), conv(16, 64, 640, 320), conv(16, 32, 640, 640, True), conv(16, 16, 1280, 1280, True), lin(4096, 10240, 1280, act=2), lin(65536, 640, 2560, res=True)]
names = {0: "baseline", 4096: "no epilogue", 1: "no DMA issue", 2: "no fragment reads", 4: "4 of 40 MFMAs"}
for fn, fl, name in cases:
    print(name)
    for bits, what in names.items():
        L.ld_debug_gemm_v5_dbg(bits)
        t = min(graph_time(fn, 10) for _ in range(3))
        print(f"   {what:22s} {t * 1e3:9.1f} us  {fl / t / 1e9:7.0f} TF/s-equivalent", flush=True)
    L.ld_debug_gemm_v5_dbg(0)
