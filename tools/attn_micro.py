"""Attention kernel timing: V^T operand vs row-major V (separate tensors, and the fused [q | k | v] layout of the UNet executor), same process,
interleaved.  Usage: python tools/attn_micro.py [quick]"""
import math, sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd._lib import lib, check
DEV = "cuda:0"
def case(b, heads, l, lk, d, reps=10):
    c = heads * d
    qkv = torch.randn(b, l, 3 * c, device=DEV, dtype=torch.float16) * 0.5
    q = qkv[:, :, :c].contiguous(); k = qkv[:, :lk, c:2 * c].contiguous(); v = qkv[:, :lk, 2 * c:].contiguous()
    lkp = (lk + 7) // 8 * 8
    vt = torch.zeros(b, c, lkp, device=DEV, dtype=torch.float16); vt[:, :, :lk] = v.transpose(1, 2)
    o = torch.empty_like(q)
    s = torch.cuda.current_stream().cuda_stream
    sc = 1 / math.sqrt(d)
    L = lib()
    runs = {"V^T": lambda: check(L.ld_op_attention(q.data_ptr(), c, k.data_ptr(), c, vt.data_ptr(), lkp, o.data_ptr(), c, b, heads, l, lk, d, sc, 0, s), "attn"),
            "rowV": lambda: check(L.ld_op_attention_rowv(q.data_ptr(), c, k.data_ptr(), c, v.data_ptr(), c, o.data_ptr(), c, b, heads, l, lk, d, sc, 0, s), "attn")}
    if l == lk:
        p = qkv.data_ptr()
        runs["qkv"] = lambda: check(L.ld_op_attention_rowv(p, 3 * c, p + 2 * c, 3 * c, p + 4 * c, 3 * c, o.data_ptr(), c, b, heads, l, lk, d, sc, 0, s), "attn")
    t = {n: [] for n in runs}
    for n, run in runs.items():
        run()
    torch.cuda.synchronize()
    for _ in range(5):
        for n, run in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
            t[n].append(e0.elapsed_time(e1) / reps)
    fl = 4.0 * b * heads * l * lk * d
    print(f"attn b={b} h={heads} L={l} Lk={lk} d={d}: " + "  ".join(f"{n} {min(v)*1e3:8.1f} us {fl/min(v)/1e9:6.1f} TF/s" for n, v in t.items()), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    case(16, 8, 4096, 4096, 40); case(16, 8, 1024, 1024, 80)
else:
    case(16, 8, 4096, 4096, 40); case(16, 8, 1024, 1024, 80); case(16, 8, 256, 256, 160); case(16, 8, 4096, 77, 40); case(2, 8, 4096, 4096, 40); case(2, 8, 1024, 1024, 80)
    case(8, 8, 16384, 16384, 40, 3)
