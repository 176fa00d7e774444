"""Driver for rocprofv3 --pmc passes: N eager forwards of the SD1.5 UNet at image batch B (UNet batch 2B, 64x64 latents or the
hires 128x128) — exactly the per-step kernel sequence bench.py times, without the sampler stack around it (bench.py itself
crashes / hangs inside rocprofiler's counter-collection path on this ROCm build; the kernel-trace passes do use bench.py).
Usage: python3 tools/pmc_unet.py <B> [latent side] [forwards]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_unet
B = int(sys.argv[1])
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(L, L))
g = torch.Generator().manual_seed(0)
u.set_context(torch.randn(2 * B, 77, 768, generator=g))
x = (torch.randn(2 * B, 4, L, L, generator=g) * 3.0).cuda()
s = torch.full((2 * B,), 3.0, device="cuda")
for _ in range(N):
    y = u.forward(x, s)
torch.cuda.synchronize()
print("pmc_unet: forwards", N, "launches/forward", u.last_launches, "finite", bool(torch.isfinite(y).all()))
