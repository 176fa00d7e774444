"""Same-process A/B of executor-level switches (A/B build): the SD1.5 UNet forward replayed from a hipGraph, interleaved rounds.
Usage: LD_MI355X_LIB=lightdiffusion_amd/libld_mi355x_ab.so python tools/ab_unet.py <flag bits to compare against 0> [batches...]"""
import os, sys, statistics
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
from lightdiffusion_amd import weights as W
from lightdiffusion_amd._lib import lib
from lightdiffusion_amd.unet import synthetic_unet
bits = int(sys.argv[1])          # executor flags (ld_debug_unet_flags); add 256 * n to also set ld_debug_gemm_no_v5(n) for the B arm
batches = [int(a) for a in sys.argv[2:]] or [8, 1]
L = lib()
for B in batches:
    u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(64, 64))
    g = torch.Generator().manual_seed(0)
    u.set_context(torch.randn(2 * B, 77, 768, generator=g))
    x = (torch.randn(2 * B, 4, 64, 64, generator=g) * 3.0).cuda()
    s = torch.full((2 * B,), 3.0, device="cuda")
    out = torch.empty_like(x)
    graphs = {}
    for f in (0, bits):
        L.ld_debug_unet_flags(f & 255)
        L.ld_debug_gemm_no_v5(f >> 8)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            u.forward(x, s, out=out)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                u.forward(x, s, out=out)
        torch.cuda.synchronize()
        graphs[f] = (gr, u.last_launches)
    L.ld_debug_unet_flags(0)
    L.ld_debug_gemm_no_v5(0)
    t = {f: [] for f in graphs}
    for _ in range(7):
        for f, (gr, _) in graphs.items():
            gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10 if B > 1 else 30):
                gr.replay()
            e1.record(); torch.cuda.synchronize()
            t[f].append(e0.elapsed_time(e1) / (10 if B > 1 else 30))
    for f in graphs:
        print(f"B={B} flags={f}: median {statistics.median(t[f]):.3f} ms  min {min(t[f]):.3f} ms  launches {graphs[f][1]}", flush=True)
    del u
    torch.cuda.empty_cache()
