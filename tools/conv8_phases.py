"""Phase clocks of the row-resident convolution (csrc/conv8.hip) from the A/B build: waves 0 (consumer) and 4 (producer) of every workgroup
stamp s_memrealtime at start / prologue done / main loop done / slabs stored / tile complete / end.
Usage: LD_MI355X_LIB=lightdiffusion_amd/libld_mi355x_ab.so LD_C8_STAMPS=1 python tools/conv8_phases.py"""
import math, os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from lightdiffusion_amd import ops
from lightdiffusion_amd._lib import lib

lib()
DEV = "cuda:0"
names = ["prologue", "main loop", "slab store", "wait peers", "reduce/epilogue"]
for (n, hw, c1, c2, cout) in [(2, 8, 1280, 0, 1280), (2, 16, 1280, 0, 1280), (2, 16, 1280, 1280, 1280), (2, 32, 640, 0, 640), (2, 64, 320, 0, 320), (2, 64, 640, 320, 320)]:
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(n, hw, hw, c1, generator=g).half().to(DEV)
    x2 = torch.randn(n, hw, hw, c2, generator=g).half().to(DEV) if c2 else None
    cin = c1 + c2
    ga, be = torch.ones(cin).half().to(DEV), torch.zeros(cin).half().to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().to(DEV)
    wp = ops.repack_conv_weight(wt)
    b = torch.zeros(cout).half().to(DEV)
    plain = os.environ.get("C8_PLAIN") is not None          # the convolution alone (no fused GroupNorm): ld_op_conv
    for _ in range(3):
        y = ops.conv2d(x1, wp, b, 3, 1, x2) if plain else ops.group_norm_silu_conv2d(x1, ga, be, 1e-5, wp, b, x2)
    torch.cuda.synchronize()
    ws = ops._ws(0, x1.device)
    need = (192 << 20) if plain else lib().ld_op_groupnorm_conv_ws_bytes(c1, c2, n, hw, hw, cout)      # the stamps sit in the last 64 KB of the split-K region = of the operator's scratch
    raw = ws.view(torch.uint8)[need - 65536:need].cpu().numpy().view(np.uint64).reshape(-1, 4, 8)
    nb = int((raw[:, 0, 0] != 0).sum())
    if os.environ.get("LD_C8_STAMPS") is None or nb == 0:
        print(f"n={n} {hw}x{hw} Cin={cin} Cout={cout}: ran (no stamps: build without LD_AB_BUILD or LD_C8_STAMPS unset)")
        continue
    st = raw[:nb].astype(np.float64) / 100.0          # us
    t0 = st[:, :, 0].min()
    print(f"n={n} {hw}x{hw} Cin={cin} Cout={cout}: {nb} workgroups; kernel span {st[:, :, 5].max() - t0:.1f} us (first start -> last end)")
    xcc, qq = raw[:nb, 0, 6].astype(int), (raw[:nb, 0, 7] >> np.uint64(32)).astype(int)
    spread = [len(set(xcc[qq == v])) for v in sorted(set(qq))]
    print(f"   placement: blockIdx % 8 == XCC id for {int((xcc == np.arange(nb) % 8).sum())} of {nb} blocks; XCDs per weight slab: min {min(spread)} max {max(spread)}")
    a = raw[:nb].astype(np.float64)
    if a[:, 2, 4].max() > 0:
        lt = np.median(a[:, 2, 4])
        print(f"   main loop, shader clocks (median over workgroups): {lt:.0f} total; consumer wave in its barrier {np.median(a[:, 2, 0]):.0f}; halo wave in its barrier "
              f"{np.median(a[:, 2, 1]):.0f}; DMA wave in its barrier {np.median(a[:, 2, 2]):.0f}, in its vmcnt wait {np.median(a[:, 2, 3]):.0f}")
    for role, nm in ((0, "consumer wave 0"), (1, "producer wave 4")):
        d = np.diff(st[:, role, :6], axis=1)
        ok = st[:, role, 5] > 0
        print(f"   {nm}: " + "  ".join(f"{names[i]} {np.median(d[ok, i]):.1f} (max {d[ok, i].max():.1f})" for i in range(5) if np.isfinite(d[ok, i]).all() and (d[ok, i] < 1e6).all()))
