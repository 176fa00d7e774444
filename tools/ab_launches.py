"""Per-launch A/B of the contraction dispatch inside the real UNet forward (A/B build): the forward is profiled launch by launch
(ld_unet_profile + LD_PROFILE_DUMP) once per ld_debug_gemm_no_v5 flag set, and the launches are joined by position.
Usage: python tools/ab_launches.py [batch=8] [flag sets, default "0 1 2 3 8 16"]   (1 no v5, 2 no v6, 8 no v7, 16 GEGLU not on v7)"""
import collections, os, re, sys, tempfile
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
os.environ['LD_PROFILE_DUMP'] = '1'
from lightdiffusion_amd import weights as W
from lightdiffusion_amd._lib import lib
from lightdiffusion_amd.unet import synthetic_unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sets = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 3, 8, 16]
hw = int(os.environ.get("AB_HW", "64"))
L = lib()
u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(hw, hw))
u.set_context(torch.randn(2 * B, 77, 768))
x = torch.randn(2 * B, 4, hw, hw, device='cuda'); s = torch.full((2 * B,), 3.0, device='cuda')
runs = {}
for f in sets:
    tile = f in (1064, 1128)                                      # 1064 / 1128: force tile height 64 / 128 on the 128 x 160 family; anything else: ld_debug_gemm_no_v5 bits
    L.ld_debug_gemm_no_v5(0 if tile else f)
    L.ld_debug_gemm_override(f - 1000 if tile else 0, 0)
    for _ in range(2): u.forward(x, s)
    torch.cuda.synchronize()
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
        if best is None: best = rows
        else: best = [(min(a[0], b[0]),) + a[1:] for a, b in zip(best, rows)]
    runs[f] = best
L.ld_debug_gemm_no_v5(0)
L.ld_debug_gemm_override(0, 0)
n = len(runs[sets[0]])
agg = collections.OrderedDict()
for i in range(n):
    r0 = runs[sets[0]][i]
    if r0[1] not in ("gemm", "geglu", "conv3", "conv1"): continue
    key = r0[1:6]
    a = agg.setdefault(key, {"n": 0, "t": {f: 0.0 for f in sets}, "k": {f: set() for f in sets}})
    a["n"] += 1
    for f in sets:
        a["t"][f] += runs[f][i][0]
        a["k"][f].add(runs[f][i][6].split("_kernel")[0].replace("gemm", "g").replace("conv", "c") + ("+sk" if "splitk" in runs[f][i][6] else ""))
print(f"B={B} hw={hw}; per-shape total us per forward under each flag set (kernel in brackets); * = best")
tot = {f: 0.0 for f in sets}
for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"][sets[0]]):
    bestf = min(sets, key=lambda f: a["t"][f])
    cells = []
    for f in sets:
        tot[f] += a["t"][f]
        cells.append(f"{a['t'][f]:8.1f}{'*' if f == bestf else ' '}[{','.join(sorted(a['k'][f]))}]")
    print(f"{key[0]:6s}{key[1]:7d}{key[2]:6d}{key[3]:6d} b{key[4]:<3d} n={a['n']:2d} " + "  ".join(cells))
print("totals: " + "  ".join(f"{f}: {tot[f]:.0f}" for f in sets))
