"""In-forward sweep of (tile height, split-K) for single shapes of the 128 x 160 kernel family (A/B build): the UNet forward is profiled
launch by launch (ld_unet_profile + LD_PROFILE_DUMP) with ONE shape overridden at a time, and the launches of that shape are summed.
(An isolated, graph-replayed sweep keeps a shape's weights in L2 / MALL and mis-ranks the candidates — profiles/README.md.)
Usage: python tools/ab_shape.py [batch=1]"""
import os, re, sys, tempfile
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
os.environ['LD_PROFILE_DUMP'] = '1'
from lightdiffusion_amd import weights as W
from lightdiffusion_amd._lib import lib
from lightdiffusion_amd.unet import synthetic_unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = lib()
hw = int(os.environ.get("AB_HW", "64"))
u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(hw, hw))
u.set_context(torch.randn(2 * B, 77, 768))
x = torch.randn(2 * B, 4, hw, hw, device='cuda'); s = torch.full((2 * B,), 3.0, device='cuda')


def profile():
    best = None
    for rep in range(3):
        tf = tempfile.TemporaryFile(mode="w+b")
        sys.stderr.flush()
        old = os.dup(2); os.dup2(tf.fileno(), 2)
        try:
            u.profile(x, s)
        finally:
            os.dup2(old, 2); os.close(old)
        tf.seek(0)
        rows = []
        for l in tf.read().decode(errors="replace").splitlines():
            m = re.match(r"\[ld_profile\]\s+([\d.]+) us\s+(\S+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+) GFLOP\s+(\S.*)$", l)
            if m: rows.append((float(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), m.group(8)))
        best = rows if best is None else [(min(a[0], b[0]),) + a[1:] for a, b in zip(best, rows)]
    return best


for _ in range(2): u.forward(x, s)
torch.cuda.synchronize()
base = profile()
shapes = {}
for r in base:
    if r[1] in ("gemm", "conv3", "conv1") and r[5] == 1 and ("gemm3" in r[6] or "gemm4" in r[6]):
        k = (r[2], r[3], r[4])
        shapes.setdefault(k, [0, 0.0, r[6]])
        shapes[k][0] += 1; shapes[k][1] += r[0]
top = sorted(shapes.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("AB_TOP", "14"))]
tot0 = sum(r[0] for r in base)
print(f"B={B}: forward (event-timed launches) {tot0:.0f} us; shape: launches, us now [kernel] | candidates bm/sk: us (whole forward delta)")
cands = [(64, 1), (64, 2), (64, 3), (64, 4), (64, 6), (64, 8), (64, 12), (64, 16), (128, 1), (128, 2), (128, 3), (128, 4), (128, 6), (128, 8), (128, 16)]
for (M, N, K), (n, t, kern) in top:
    res = []
    for bm, sk in cands:
        if sk > 1 and K // sk < 512: continue
        L.ld_debug_gemm_shape_override(M, N, K, bm, sk)
        try:
            rows = profile()
        except Exception:
            continue
        tt = sum(r[0] for r in rows if (r[2], r[3], r[4]) == (M, N, K) and r[5] == 1)
        res.append((tt, bm, sk, sum(r[0] for r in rows) - tot0))
    L.ld_debug_gemm_shape_override(0, 0, 0, 0, 0)
    res.sort()
    print(f"{M:6d}x{N:5d}x{K:6d} n={n:2d} {t:7.1f} [{kern.split('_kernel')[0]}{'+sk' if 'splitk' in kern else ''}] | " +
          "  ".join(f"{bm}/{sk}: {tt:.1f} ({d:+.0f})" for tt, bm, sk, d in res[:5]), flush=True)
