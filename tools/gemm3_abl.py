"""Ablations of the 128 x 160 kernel (gemm3, two workgroups per CU) on the batch-8 step's K = 640 / 1280 projections and MLP-out folds (A/B
build; wrong results on purpose, timing only): what do the prologue, the slab loop's DMA / fragment reads / MFMAs and the epilogue cost?
ws = None: no split over K, as inside the forward (the LayerNorm-fold producers / consumers never split)."""
import math, os, sys
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
src = open(os.path.join("tools", "gemm5_ab.py")).read().split("cases = [")[0]
exec(src)
def lin_nosplit(m, n, k, res=False):
    x = torch.randn(m, k, device=DEV, dtype=torch.float16)
    w = torch.randn(n, k, device=DEV, dtype=torch.float16) / math.sqrt(k)
    b = torch.randn(n, device=DEV, dtype=torch.float16) * 0.1
    y = torch.empty(m, n, device=DEV, dtype=torch.float16)
    r = torch.randn(m, n, device=DEV, dtype=torch.float16) if res else None
    def fn():
        check(L.ld_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(), m, n, k, 1.0, 0,
                             None, 0, torch.cuda.current_stream().cuda_stream), "lin")
    return fn, 2.0 * m * n * k, f"gemm  {m}x{n}x{k}{' +res' if res else ''} (no split)"
cases = [lin_nosplit(16384, 640, 640, True), lin_nosplit(4096, 1280, 1280, True), lin_nosplit(16384, 1920, 640), lin_nosplit(16384, 640, 3200, True), lin_nosplit(65536, 320, 1600, True)]
names = {0: "baseline", 1: "no DMA behind the prologue", 2: "no fragment reads", 4: "no MFMAs", 7: "only barriers (+ epilogue)", 8: "no epilogue", 15: "prologue + barriers only",
         32: "no residual loads", 64: "no output stores", 96: "no residual loads, no stores", 103: "barriers + LDS staging of the tile only"}
if len(sys.argv) > 1 and sys.argv[1] == "quick": names = {0: "baseline", 8: "no epilogue"}
if len(sys.argv) > 1 and sys.argv[1] == "stagger": names = {0: "baseline", 0x100: "second 256 blocks 1 us late", 0x200: "2 us late", 0x300: "3 us late", 0x400: "4 us late", 0x500: "5 us late", 0x600: "6 us late"}
for fn, fl, name in cases:
    print(name)
    for bits, what in names.items():
        L.ld_debug_gemm_v5_dbg(bits)
        t = min(graph_time(fn, 20) for _ in range(3))
        print(f"   {what:28s} {t * 1e3:9.2f} us  {fl / t / 1e9:7.0f} TF/s-equivalent", flush=True)
    L.ld_debug_gemm_v5_dbg(0)
