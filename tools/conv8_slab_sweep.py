#!/usr/bin/env python3
"""Slab-count sweep of the row-resident 3x3 convolution (conv8) inside the batch-1 forward ("conv8's 3.6x
traffic — fewer, deeper slabs wherever tiles >= CUs: show the slab-count sweep").

conv8 splits a tile's input channels over S workgroups ("slabs") whose fp32 partial sums meet in HBM inside the launch; `conv8_plan`
takes S = 256 / tiles (as many workgroups as CUs), capped by the channel count.  The A/B build reads LD_C8_S_W<width> = S for the launches of
one image width; this script times the graph-replayed batch-1 forward (tools/unet_time.py 1, a fresh process per point) for every width and a
range of S, everything else at its default.  Usage (gpurun): python tools/conv8_slab_sweep.py > gpurun_out/r06_conv8_slab_sweep.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "lightdiffusion_amd", "libld_mi355x_ab.so")


def run(extra):
    env = dict(os.environ, LD_MI355X_LIB=AB, **extra)
    r = subprocess.run([sys.executable, os.path.join("tools", "unet_time.py"), "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    m = re.search(r"median ([0-9.]+) ms\s+min ([0-9.]+) ms", r.stdout)
    return (float(m.group(1)), float(m.group(2))) if m else (float("nan"), float("nan"))


def main():
    print("# batch-1 forward (UNet batch 2, 64 x 64 latents, plain route), A/B build, graph replay: median / min ms")
    base = [run({}) for _ in range(2)]
    print(f"default plan (W64: S = 1 / 2, W32: 2, W16: 4, W8: 8 .. 16): {base[0][0]:.3f} / {base[0][1]:.3f}   again: {base[1][0]:.3f} / {base[1][1]:.3f}")
    for w, cands in ((64, (1, 2, 4)), (32, (1, 2, 4, 8)), (16, (1, 2, 4, 8, 16)), (8, (2, 4, 8, 16))):
        row = []
        for s in cands:
            med, mn = run({f"LD_C8_S_W{w}": str(s)})
            row.append(f"S={s}: {med:.3f} / {mn:.3f}")
        print(f"W{w:<2d}  " + "   ".join(row), flush=True)
    again = run({})
    print(f"default plan, at the end: {again[0]:.3f} / {again[1]:.3f}")


if __name__ == "__main__":
    main()
