"""Per-launch table of one SD1.5 VAE decode (HIP events around every launch, `ld_vae_profile`): what, shape, algorithmic FLOPs,
minimal HBM bytes (inputs + weights + output once, fp16), microseconds, TFLOP/s, GB/s, kernel — the per-layer VAE table of
profiles/ (BASELINE.md §4 row 5).  Usage: python3 tools/vae_prof.py [batch=8] [latent side=64]"""
import collections, statistics, sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_vae

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = int(sys.argv[2]) if len(sys.argv) > 2 else 64
vae = synthetic_vae(W.sd15_vae_config(), max_batch=b, max_hw=(h, h))
z = torch.randn(b, 4, h, h, generator=torch.Generator().manual_seed(0)).cuda() * 0.5
for _ in range(2):
    vae.decode_device(z)
runs = [vae.profile_decode(z) for _ in range(5)]
rows = runs[0]
us = [statistics.median(r[i][3] for r in runs) for i in range(len(rows))]


def min_bytes(what, d):
    a, n, k, bt = d
    if what in ("conv3", "conv1", "gemm", "geglu"):
        cin = k // 9 if what == "conv3" else k
        return 2 * bt * (a * cin + n * k + a * n)      # input pixels x Cin (not the im2col), weights, output
    if what == "groupnorm":
        return 2 * a * n * k * 3                        # (images, pixels, channels): read twice (statistics, apply), write once
    if what == "softmax":
        return 2 * a * n * 2
    if what == "conv_in":
        return 2 * a * n + 4 * a * 4
    if what == "conv_out":
        return 2 * a * k // 9 + 4 * a * n
    return 0


print(f"# SD1.5 VAE decode, batch {b}, latent {h}x{h} -> {8 * h}x{8 * h}; median of 5 profiled decodes; {len(rows)} launches")
print(f"{'what':10s} {'M':>9s} {'N':>6s} {'K':>6s} {'b':>3s} {'GFLOP':>9s} {'MB(min)':>9s} {'us':>9s} {'TFLOP/s':>8s} {'GB/s':>7s}  kernel")
tot_us = tot_fl = 0.0
agg = collections.OrderedDict()
for (what, d, fl, _, kern), t in zip(rows, us):
    by = min_bytes(what, d)
    print(f"{what:10s} {d[0]:9d} {d[1]:6d} {d[2]:6d} {d[3]:3d} {fl / 1e9:9.2f} {by / 1e6:9.1f} {t:9.1f} {fl / t / 1e6 if t else 0:8.1f} {by / t / 1e3 if t else 0:7.0f}  {kern}")
    tot_us += t
    tot_fl += fl
    key = (what, d, kern)
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += t; a[2] += fl; a[3] += by
print(f"# total {tot_us / 1e3:.2f} ms of kernel time, {tot_fl / 1e12:.2f} TFLOP -> {tot_fl / tot_us / 1e6:.0f} TFLOP/s")
print("# by shape (launches, total us, share, TFLOP/s, GB/s):")
for (what, d, kern), (n, t, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"#  {what:10s} {d[0]:9d} {d[1]:6d} {d[2]:6d} n={n:2d} {t:9.1f} us {100 * t / tot_us:5.1f}%  {fl / t / 1e6:7.1f} TF/s {by / t / 1e3:6.0f} GB/s  {kern}")
