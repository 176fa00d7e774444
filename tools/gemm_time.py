"""Time a few contraction shapes with the library LD_MI355X_LIB names (default: the shipped one) — no debug hooks, graph replay."""
import os, sys
import torch
sys.path.insert(0, '.')
src = open(os.path.join("tools", "gemm5_ab.py")).read().split("cases = [")[0].replace('os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))', '')
exec(src)
cases = [lin(65536, 320, 1280, res=True), lin(16384, 1280, 640), lin(16384, 640, 640, res=True), lin(16384, 640, 640), lin(4096, 1280, 1280, res=True),
         lin(4096, 2560, 1280), lin(16384, 640, 2560, res=True), lin(65536, 320, 320, res=True), lin(65536, 2560, 320, act=2), lin(16384, 5120, 640, act=2), lin(4096, 10240, 1280, act=2),
         conv(16, 64, 320, 320, True), conv(16, 32, 640, 640), conv(16, 16, 1280, 1280), conv(16, 8, 1280, 1280),
         lin(2048, 640, 640, res=True), lin(512, 1280, 1280, res=True), lin(8192, 320, 320, res=True), conv(2, 8, 1280, 1280), conv(2, 16, 1280, 1280)]
for fn, fl, name in cases:
    reps = max(3, min(50, int(2e-3 / (fl / 0.8e15)) + 1))
    t = min(graph_time(fn, reps) for _ in range(3))
    print(f"{name:34s} {t * 1e3:9.1f} us {fl / t / 1e9:7.0f} TF/s", flush=True)
