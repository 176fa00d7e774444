"""Ablations of the producer / consumer kernel (gemm4, 64 x 64 tiles) on the batch-1 step's skinny projections (A/B build; wrong results on
purpose, timing only), and ring 8 / two workgroups per CU against round 4's rings (ld_debug_gemm_no_v5 bit 2048)."""
import math, os, sys
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
src = open(os.path.join("tools", "gemm5_ab.py")).read().split("cases = [")[0]
exec(src)
cases = [lin(512, 1280, 1280, res=True), lin(2048, 640, 640, res=True), lin(8192, 320, 320, res=True), lin(512, 3840, 1280), lin(128, 1280, 1280, res=True)]
names = {0: "baseline", 1: "no DMA behind the prologue", 2: "no fragment reads", 4: "no MFMAs", 6: "no reads, no MFMAs", 7: "only barriers", 8: "no epilogue", 15: "nothing"}
for fn, fl, name in cases:
    for ring_off in (0, 2048):
        L.ld_debug_gemm_no_v5(ring_off)
        print(name, "| round-4 rings" if ring_off else "| round-5 rings")
        for bits, what in names.items():
            L.ld_debug_gemm_v5_dbg(bits)
            t = min(graph_time(fn, 20) for _ in range(3))
            print(f"   {what:28s} {t * 1e3:9.2f} us", flush=True)
        L.ld_debug_gemm_v5_dbg(0)
L.ld_debug_gemm_no_v5(0)
