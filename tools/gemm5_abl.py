"""Ablations of the gemm5 main loop (A/B build; wrong results on purpose, timing only): what does each phase cost?"""
import math, os, sys
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("ab", os.path.join("tools", "gemm5_ab.py"))
src = open(os.path.join("tools", "gemm5_ab.py")).read().split("cases = [")[0]
exec(src)
cases = [conv(16, 64, 640, 320), conv(16, 64, 320, 320), conv(16, 32, 1280, 1280), lin(65536, 320, 1280)]
names = {0: "baseline", 1: "no DMA issue", 2: "no fragment reads", 4: "4 of 40 MFMAs", 3: "no DMA, no reads", 6: "no reads, 4 MFMAs", 5: "no DMA, 4 MFMAs"}
if os.environ.get("ABL_SET", "") == "7":
    names = {0: "baseline", 128: "1 of 10 k-steps", 256: "no epilogue", 384: "neither", 512: "no gelu", 1024: "no output stores", 1536: "no stores, no gelu"}
    cases = [lin(65536, 2560, 320, act=2), lin(65536, 320, 320, res=True), lin(65536, 640, 320), lin(65536, 2560, 320)]
    elif os.environ.get("ABL_SET", "") == "2":
    names = {0: "baseline", 64: "all WGs stream tile (0,0)", 65: "same tile, no DMA issue"}
elif len(os.environ.get("ABL_SET", "")):
    names = {0: "baseline", 8: "DMA after the reads", 16: "no setprio(1) on MFMA", 32: "read phase at prio 2", 48: "read prio 2, no mfma prio", 24: "DMA after reads, no mfma prio",
             40: "DMA after reads, read prio 2", 56: "all three"}
for fn, fl, name in cases:
    print(name)
    for bits, what in names.items():
        L.ld_debug_gemm_v5_dbg(bits)
        t = min(graph_time(fn, 10) for _ in range(3))
        print(f"   {what:22s} {t * 1e3:9.1f} us  {fl / t / 1e9:7.0f} TF/s-equivalent", flush=True)
    L.ld_debug_gemm_v5_dbg(0)
