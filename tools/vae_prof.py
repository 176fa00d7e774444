"""VAE decode loop for rocprofv3 --kernel-trace --stats (per-kernel share of a 512x512 decode)."""
import sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_vae
vae = synthetic_vae(W.sd15_vae_config(), max_batch=1, max_hw=(64, 64))
z = torch.randn(1, 4, 64, 64, device='cuda') * 0.5
for _ in range(12): vae.decode_device(z)
torch.cuda.synchronize()
