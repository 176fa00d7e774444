import math, sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd._lib import lib, check
DEV = "cuda:0"
def case(b, heads, l, lk, d, reps=10):
    c = heads * d
    q = torch.randn(b, l, c, device=DEV, dtype=torch.float16); k = torch.randn(b, lk, c, device=DEV, dtype=torch.float16)
    lkp = (lk + 7) // 8 * 8
    vt = torch.randn(b, c, lkp, device=DEV, dtype=torch.float16); o = torch.empty_like(q)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: check(lib().ld_op_attention(q.data_ptr(), c, k.data_ptr(), c, vt.data_ptr(), lkp, o.data_ptr(), c, b, heads, l, lk, d, 1 / math.sqrt(d), 0, s), "attn")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"attn b={b} h={heads} L={l} Lk={lk} d={d}: {ms*1e3:8.1f} us  {4.0*b*heads*l*lk*d/ms/1e9:7.1f} TF/s", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    case(16, 8, 4096, 4096, 40); case(16, 8, 1024, 1024, 80)
else:
    case(16, 8, 4096, 4096, 40); case(16, 8, 1024, 1024, 80); case(16, 8, 256, 256, 160); case(16, 8, 4096, 77, 40); case(8, 8, 16384, 16384, 40, 3)
