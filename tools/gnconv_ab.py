"""A/B of GroupNorm+SiLU+conv3x3 (ld_op_groupnorm_conv): fused into the halo kernel vs two-pass GroupNorm + the plain halo kernel."""
import math, os, sys
import torch
sys.path.insert(0, '.')
os.environ.setdefault("LD_MI355X_LIB", os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
src = open(os.path.join("tools", "gemm5_ab.py")).read().split("cases = [")[0]
exec(src)

def gnconv(nimg, h, cin, cout):
    x = torch.randn(nimg, h, h, cin, device=DEV, dtype=torch.float16)
    wt = torch.randn(cout, 9 * cin, device=DEV, dtype=torch.float16) / math.sqrt(9 * cin)
    b = torch.randn(cout, device=DEV, dtype=torch.float16) * 0.1
    ga = torch.ones(cin, device=DEV, dtype=torch.float16); be = torch.zeros(cin, device=DEV, dtype=torch.float16)
    y = torch.empty(nimg, h, h, cout, device=DEV, dtype=torch.float16)
    nb = L.ld_op_groupnorm_conv_ws_bytes(cin, 0, nimg, h, h, cout)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    def fn():
        check(L.ld_op_groupnorm_conv(x.data_ptr(), cin, None, 0, nimg, h, h, ga.data_ptr(), be.data_ptr(), 1e-5, wt.data_ptr(), b.data_ptr(), None, None,
                                     y.data_ptr(), cout, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "gnconv")
    return fn, 2.0 * nimg * h * h * cout * 9 * cin, f"gn+conv {nimg * h * h}x{cout}x{9 * cin}"

for fn, fl, name in [gnconv(16, 64, 320, 320), gnconv(16, 64, 640, 320), gnconv(16, 32, 640, 640), gnconv(16, 32, 1280, 640), gnconv(16, 16, 1280, 1280)]:
    t = {0: [], 4: []}
    for _ in range(3):
        for off in (4, 0):
            L.ld_debug_gemm_no_v5(off)
            t[off].append(graph_time(fn, 10))
    L.ld_debug_gemm_no_v5(0)
    a, b = min(t[4]), min(t[0])
    print(f"{name:34s} two-pass {a * 1e3:8.1f} us   fused {b * 1e3:8.1f} us   ratio {a / b:5.2f}", flush=True)
