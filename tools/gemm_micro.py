"""Micro-benchmark of the dominant contraction shapes (for rocprofv3 --pmc runs and A/B timing)."""
import math, sys, time
import torch
sys.path.insert(0, '.')
from lightdiffusion_amd import ops

DEV = "cuda:0"
def conv_case(n, h, w, cin, cout, reps):
    x = torch.randn(n, h, w, cin, device=DEV, dtype=torch.float16)
    wt = (torch.randn(cout, 9 * cin, device=DEV, dtype=torch.float16) / math.sqrt(9 * cin))
    b = torch.zeros(cout, device=DEV, dtype=torch.float16)
    ops.conv2d(x, wt, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d(x, wt, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * n * h * w * cout * 9 * cin
    print(f"conv3 n={n} {h}x{w} {cin}->{cout}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)

def lin_case(m, n, k, reps, res=False):
    x = torch.randn(m, k, device=DEV, dtype=torch.float16)
    w = torch.randn(n, k, device=DEV, dtype=torch.float16) / math.sqrt(k)
    b = torch.zeros(n, device=DEV, dtype=torch.float16)
    r = torch.randn(m, n, device=DEV, dtype=torch.float16) if res else None
    ops.linear(x, w, b, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.linear(x, w, b, r)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"linear {m}x{n}x{k}: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:7.1f} TF/s", flush=True)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
conv_case(16, 64, 64, 320, 320, reps)
conv_case(16, 32, 32, 1280, 640, reps)     # 16384 x 640 x 11520
conv_case(16, 16, 16, 1280, 1280, reps)    # 4096 x 1280 x 11520
lin_case(65536, 320, 320, reps, True)
lin_case(8192, 8192, 8192, max(2, reps // 4))
