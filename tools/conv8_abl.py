"""Ablations of conv8's main loop (A/B build, LD_C8_ABL bits: 1 no fragment reads, 2 no MFMAs, 4 no halo staging, 8 no weight DMA inside the loop; wrong results,
timing only): what the consumer waves wait for.  Usage: python tools/conv8_abl.py"""
import math, os, sys, torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
from lightdiffusion_amd import ops
def r16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half()
names = {0: "full", 1: "no fragment reads", 2: "no MFMAs", 3: "no reads, no MFMAs", 4: "no halo staging", 8: "no weight DMA", 12: "no producers", 13: "MFMAs only", 14: "reads only", 15: "barriers only"}
for (n, hw, cin, cout) in [(2, 16, 1280, 1280), (2, 64, 320, 320), (2, 8, 1280, 1280), (2, 32, 640, 640)]:
    x = r16((n, hw, hw, cin), 1).cuda(); w = ops.repack_conv_weight(r16((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin)).cuda()); b = r16((cout,), 3, 0.1).cuda()
    res = []
    for bits in (0, 1, 2, 3, 4, 8, 12, 13, 14, 15):
        os.environ["LD_C8_ABL"] = str(bits)
        for _ in range(3): ops.conv2d(x, w, b)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv2d(x, w, b)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res.append(f"{names[bits]} {min(ts):.1f}")
    print(f"conv {n}x{hw}x{hw}x{cin}->{cout} (us per call incl. the per-call weight repack + memset of the op entry): " + " | ".join(res), flush=True)
