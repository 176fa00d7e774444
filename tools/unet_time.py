"""Graph-replayed SD1.5 UNet forward time with the library LD_MI355X_LIB names (default: the shipped one).  Usage: python tools/unet_time.py [batches...]"""
import os, sys, statistics
import torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_unet
hw = int(os.environ.get("AB_HW", "64"))
for B in [int(a) for a in sys.argv[1:]] or [8, 1]:
    u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(hw, hw))
    g = torch.Generator().manual_seed(0)
    u.set_context(torch.randn(2 * B, 77, 768, generator=g))
    x = (torch.randn(2 * B, 4, hw, hw, generator=g) * 3.0).cuda()
    s = torch.full((2 * B,), 3.0, device="cuda")
    out = torch.empty_like(x)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        u.forward(x, s, out=out)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            u.forward(x, s, out=out)
        for _ in range(5): gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); gr.replay(); e1.record(st); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    print(f"B={B} hw={hw}: median {statistics.median(ts):.3f} ms  min {min(ts):.3f} ms  checksum {out.float().abs().mean().item():.6f}", flush=True)
    del u
