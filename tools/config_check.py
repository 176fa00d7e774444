"""Exercise BASELINE configs beyond the default bench: VAE decode 512^2 / 1024^2, hires-fix UNet at 128x128 latents."""
import sys, time
import torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_unet, synthetic_vae
from lightdiffusion_amd.pipeline import CFGDenoiser

def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

vae = synthetic_vae(W.sd15_vae_config(), max_batch=4, max_hw=(128, 128))
for b, hw in ((1, 64), (8, 64), (4, 128)):
    if b * hw * hw > 4 * 128 * 128: continue
    z = torch.randn(b, 4, hw, hw, device='cuda') * 0.5
    dt = t(lambda: vae.decode_device(z))
    img = vae.decode_device(z)
    print(f"VAE decode b={b} {hw*8}x{hw*8}: {dt*1e3:8.2f} ms  {b/dt:7.2f} img/s  {vae.last_flops/dt/1e12:7.1f} TF/s  launches={vae.last_launches} finite={bool(torch.isfinite(img).all())}", flush=True)
del vae; torch.cuda.empty_cache()
unet = synthetic_unet(W.sd15_unet_config(), max_batch=8, max_hw=(128, 128))
d = CFGDenoiser(unet, 4, 128, 128, 8.0)
d.set_context(torch.randn(1, 77, 768), torch.randn(1, 77, 768))
x = torch.randn(4, 4, 128, 128, device='cuda') * 1.3
dt = t(lambda: d(x, 1.2768))
print(f"hires UNet CFG step b=4 128x128 latents: {dt*1e3:8.2f} ms/step  {1/dt:6.2f} steps/s  {unet.last_flops/dt/1e12:7.1f} TF/s finite={bool(torch.isfinite(d.den).all())}", flush=True)
p = unet.profile_pair(d.x1, d.sigma1)
print({k: round(v[0], 2) for k, v in p.items()})
