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

def graph_time(fn, reps):
    """GPU time per call with the launches replayed from a hipGraph (small kernels are host-bound when launched from Python)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def lin_graph(m, n, k, reps):
    x = torch.randn(m, k, device=DEV, dtype=torch.float16)
    w = torch.randn(n, k, device=DEV, dtype=torch.float16) / math.sqrt(k)
    b = torch.zeros(n, device=DEV, dtype=torch.float16)
    y = torch.empty(m, n, device=DEV, dtype=torch.float16)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    from lightdiffusion_amd._lib import lib, check
    def fn():
        check(lib().ld_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), m, n, k, 1.0, 0, ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream().cuda_stream), "lin")
    ms = graph_time(fn, reps)
    print(f"[graph] linear {m}x{n}x{k}: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:7.1f} TF/s", flush=True)

def conv_graph(n, h, w_, cin, cout, reps):
    x = torch.randn(n, h, w_, cin, device=DEV, dtype=torch.float16)
    wt = (torch.randn(cout, 9 * cin, device=DEV, dtype=torch.float16) / math.sqrt(9 * cin))
    b = torch.zeros(cout, device=DEV, dtype=torch.float16)
    y = torch.empty(n, h, w_, cout, device=DEV, dtype=torch.float16)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    from lightdiffusion_amd._lib import lib, check
    def fn():
        check(lib().ld_op_conv(x.data_ptr(), cin, None, 0, n, h, w_, h, w_, 1, 3, wt.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), cout,
                               ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "conv")
    ms = graph_time(fn, reps)
    print(f"[graph] conv3 n={n} {h}x{w_} {cin}->{cout}: {ms*1e3:8.1f} us  {2.0*n*h*w_*cout*9*cin/ms/1e9:7.1f} TF/s", flush=True)

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
if len(sys.argv) > 2 and sys.argv[2] == "small":   # the latency-bound shapes of a batch-1 step (UNet batch 2)
    for m, n, k in [(512, 1280, 1280), (2048, 640, 640), (8192, 320, 320), (512, 1280, 5120), (128, 1280, 1280), (2048, 640, 2560), (8192, 320, 1280)]:
        lin_graph(m, n, k, reps)
    conv_graph(2, 16, 16, 1280, 1280, reps); conv_graph(2, 8, 8, 1280, 1280, reps); conv_graph(2, 64, 64, 320, 320, reps)
    conv_graph(2, 32, 32, 640, 640, reps); conv_graph(2, 16, 16, 2560, 1280, reps)
    sys.exit(0)
conv_case(16, 64, 64, 320, 320, reps)
conv_case(16, 32, 32, 1280, 640, reps)     # 16384 x 640 x 11520
conv_case(16, 16, 16, 1280, 1280, reps)    # 4096 x 1280 x 11520
lin_case(65536, 320, 320, reps, True)
lin_case(8192, 8192, 8192, max(2, reps // 4))
