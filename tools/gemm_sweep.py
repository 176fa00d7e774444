"""Per-shape sweep of (bm, splitk) for every contraction of the SD1.5 UNet at UNet batch N (2 = B1, 16 = B8)."""
import ctypes, math, sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd import ops
from lightdiffusion_amd._lib import lib
L = lib(); L._handle
ovr = ctypes.CDLL(None)  # placeholder
import lightdiffusion_amd._lib as _l
raw = ctypes.CDLL(_l.LIB_PATH)
raw.ld_debug_gemm_override.argtypes = [ctypes.c_int, ctypes.c_int]
DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2

def timeit(fn, reps=10):
    """GPU time per call, launches replayed from a hipGraph: from Python the small shapes are host-bound (~12 us per call)."""
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        g.replay(); g.replay()
        e1.record(st); torch.cuda.synchronize()
    del g
    return e0.elapsed_time(e1) / (2 * reps) * 1e3

def sweep(name, fn, flops):
    res = []
    for bm in (64, 128):
        for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24):
            raw.ld_debug_gemm_override(bm, sk)
            try:
                us = timeit(fn)
            except Exception as e:
                continue
            res.append((us, bm, sk))
    raw.ld_debug_gemm_override(0, 0)
    auto = timeit(fn)
    res.sort()
    best = res[0]
    print(f"{name:34s} auto {auto:7.1f} us | best {best[0]:7.1f} us bm={best[1]:3d} sk={best[2]:2d} ({flops/best[0]/1e6:6.0f} TF/s) | " +
          " ".join(f"{b}/{s}:{u:.0f}" for u, b, s in res[1:4]), flush=True)

def conv(h, cin, cout, stride=1):
    x = torch.randn(N, h, h, cin, device=DEV, dtype=torch.float16)
    w = torch.randn(cout, 9 * cin, device=DEV, dtype=torch.float16) / math.sqrt(9 * cin)
    b = torch.zeros(cout, device=DEV, dtype=torch.float16)
    ho = (h - 1) // stride + 1
    sweep(f"conv3 M={N*ho*ho} N={cout} K={9*cin}", lambda: ops.conv2d(x, w, b, 3, stride), 2.0 * N * ho * ho * cout * 9 * cin)

def lin(m, n, k, res=True):
    x = torch.randn(m, k, device=DEV, dtype=torch.float16)
    w = torch.randn(n, k, device=DEV, dtype=torch.float16) / math.sqrt(k)
    b = torch.zeros(n, device=DEV, dtype=torch.float16)
    r = torch.randn(m, n, device=DEV, dtype=torch.float16) if res else None
    sweep(f"gemm  M={m} N={n} K={k}", lambda: ops.linear(x, w, b, r), 2.0 * m * n * k)

for h, c in ((64, 320), (32, 640), (16, 1280), (8, 1280)):
    conv(h, c, c)
conv(64, 960, 320); conv(64, 640, 320); conv(32, 1920, 640); conv(32, 1280, 640); conv(32, 960, 640); conv(32, 320, 640)
conv(16, 2560, 1280); conv(16, 1920, 1280); conv(16, 640, 1280); conv(8, 2560, 1280)
conv(64, 320, 320, 2); conv(32, 640, 640, 2); conv(16, 1280, 1280, 2)
for l, c in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
    lin(N * l, c, c); lin(N * l, 2 * c, c, False); lin(N * l, c, 4 * c)
lin(N * 4096, 320, 960, False); lin(N * 1024, 640, 1920, False); lin(N * 256, 1280, 2560, False); lin(N * 64, 1280, 2560, False)
