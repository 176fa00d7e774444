"""Same-process A/B (A/B library): the VAE decode with the GroupNorm of its one-tile-wide stages fused into the halo convolution (round 5)
against the two-pass GroupNorm in front of the same kernels (ld_debug_gemm_no_v5 bit 16384).  Usage: python3 tools/vae_gn_ab.py [b:h ...]"""
import os, statistics, sys
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
from lightdiffusion_amd import weights as W
from lightdiffusion_amd._lib import lib
from lightdiffusion_amd.unet import synthetic_vae

L = lib()


def timed(vae, z, n=12):
    for _ in range(3):
        vae.decode_device(z)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); vae.decode_device(z); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts), min(ts)


for spec in (sys.argv[1:] or ["8:64", "4:128"]):
    b, h = (int(v) for v in spec.split(":"))
    z = torch.randn(b, 4, h, h, generator=torch.Generator().manual_seed(0)).cuda() * 0.5
    outs = {}
    for rnd in range(2):
        for flags in (16384, 0):
            L.ld_debug_gemm_no_v5(flags)
            vae = synthetic_vae(W.sd15_vae_config(), max_batch=b, max_hw=(h, h))
            med, mn = timed(vae, z)
            outs[flags] = vae.decode_device(z).float().cpu()
            print(f"b={b} latent {h}x{h}  {'two-pass' if flags else 'fused   '}: median {med:.3f} ms  min {mn:.3f} ms", flush=True)
            del vae
            torch.cuda.empty_cache()
    L.ld_debug_gemm_no_v5(0)
    d = (outs[0] - outs[16384]).abs()
    print(f"   fused vs two-pass images: max-abs {float(d.max()) * 255:.3f} / 255, mean-abs {float(d.mean()) * 255:.4f} / 255", flush=True)
