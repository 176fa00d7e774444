#!/usr/bin/env python3
"""Split-K factor sweep of the halo convolution (conv6, 256 x 320 tiles) inside the batch-8 forward, per image width (A/B build,
LD_V6_SK_W<width>): the launch table of the CFG-pair forward (tools/launch_table.py 8 1, a fresh process per point) with the conv3 rows of that
width summed.  v6_plan takes ceil(256 / tiles) slices capped by K / 2560, from K = 8640 on.  Usage (gpurun): python tools/conv6_split_sweep.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "lightdiffusion_amd", "libld_mi355x_ab.so")
M_OF = {16: 4096, 32: 16384}


def run(extra, width):
    env = dict(os.environ, LD_MI355X_LIB=AB, **extra)
    r = subprocess.run([sys.executable, os.path.join("tools", "launch_table.py"), "8", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    tot, rows, total_all = 0.0, [], None
    for l in r.stdout.splitlines():
        m = re.match(r"conv3\s+(\d+)\s+(\d+)\s+(\d+)\s+b\d+\s+n=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) each\s+\[(.*)\]", l)
        if m and int(m.group(1)) == M_OF[width] and "conv6" in m.group(7):
            tot += float(m.group(5))
            rows.append(f"K={m.group(3)} x{m.group(4)}: {m.group(6)}")
        m = re.search(r"timed launches, sum (\d+) us", l)
        if m:
            total_all = int(m.group(1))
    return tot, rows, total_all


def main():
    for width, cands in ((16, (None, 2, 3, 4, 6, 8)), (32, (None, 1, 2, 3, 4))):
        for c in cands:
            tot, rows, allsum = run({} if c is None else {f"LD_V6_SK_W{width}": str(c)}, width)
            print(f"W{width} sk={'plan' if c is None else c}: conv6 rows {tot:8.1f} us  (forward sum {allsum} us)   " + "  ".join(rows), flush=True)


if __name__ == "__main__":
    main()
