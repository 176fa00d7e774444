"""Per-shape launch table of the UNet forward (ld_unet_profile + LD_PROFILE_DUMP) with whatever library LD_MI355X_LIB names (no debug hooks needed):
for same-box comparisons of two builds.  Usage: [LD_MI355X_LIB=...] python tools/launch_table.py [batch=1] [pair=0]"""
import collections, os, re, sys, tempfile
import torch
sys.path.insert(0, '.')
os.environ['LD_PROFILE_DUMP'] = '1'
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pair = len(sys.argv) > 2 and sys.argv[2] == "1"
hw = int(os.environ.get("AB_HW", "64"))
u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(hw, hw))
u.set_context(torch.randn(2 * B, 77, 768))
n = B if pair else 2 * B
x = torch.randn(n, 4, hw, hw, device='cuda'); s = torch.full((n,), 3.0, device='cuda')
for _ in range(2): (u.forward_pair if pair else u.forward)(x, s)
torch.cuda.synchronize()
best = None
for rep in range(5):
    tf = tempfile.TemporaryFile(mode="w+b")
    sys.stderr.flush()
    old = os.dup(2); os.dup2(tf.fileno(), 2)
    try:
        (u.profile_pair if pair else u.profile)(x, s)
    finally:
        os.dup2(old, 2); os.close(old)
    tf.seek(0)
    rows = []
    for l in tf.read().decode(errors="replace").splitlines():
        m = re.match(r"\[ld_profile\]\s+([\d.]+) us\s+(\S+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+) GFLOP\s+(\S.*)$", l)
        if m: rows.append((float(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), m.group(8)))
    best = rows if best is None else [(min(a[0], b[0]),) + a[1:] for a, b in zip(best, rows)]
agg = collections.OrderedDict()
for r in best:
    a = agg.setdefault(r[1:6], [0, 0.0, set()])
    a[0] += 1; a[1] += r[0]; a[2].add(r[6])
print(f"B={B} hw={hw} pair={int(pair)} lib={os.environ.get('LD_MI355X_LIB', 'shipped')}: {len(best)} timed launches, sum {sum(r[0] for r in best):.0f} us")
cls = collections.defaultdict(float)
for k, (n_, t, ks) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    cls[k[0]] += t
    print(f"{k[0]:10s}{k[1]:8d}{k[2]:7d}{k[3]:7d} b{k[4]:<3d} n={n_:3d} {t:9.1f} us  {t / n_:8.1f} each  [{','.join(sorted(ks))}]")
print("classes: " + "  ".join(f"{k}: {v:.0f}" for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
