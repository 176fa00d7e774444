"""SD1.5 VAE decode time with the library LD_MI355X_LIB names.  Usage: python tools/vae_time.py [batch=8] [latent=64]"""
import os, sys, statistics
import torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_vae
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = int(sys.argv[2]) if len(sys.argv) > 2 else 64
v = synthetic_vae(W.sd15_vae_config(), max_batch=b, max_hw=(h, h))
z = torch.randn(b, 4, h, h, generator=torch.Generator().manual_seed(0)).cuda()
for _ in range(2): out = v.decode_device(z)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = v.decode_device(z); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"VAE decode b={b} latent {h}x{h}: median {statistics.median(ts):.2f} ms  min {min(ts):.2f} ms  checksum {out.float().mean().item():.6f}")
