"""Same-process A/B: the plain forward on cat([x, x]) (N = 2B) against the CFG-pair forward (`ld_unet_forward_pair`: the layers in front of the first
cross-attention evaluated once), both graph-replayed.  Usage: python3 tools/pair_ab.py [B:hw ...]   (default 8:64 4:128 2:64 1:64)"""
import statistics, sys, torch
sys.path.insert(0, '.')
from lightdiffusion_amd import weights as W
from lightdiffusion_amd.unet import synthetic_unet


def timed(g, n=30):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


for spec in (sys.argv[1:] or ["8:64", "4:128", "2:64", "1:64"]):
    B, hw = (int(v) for v in spec.split(":"))
    u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(hw, hw))
    gen = torch.Generator().manual_seed(0)
    u.set_context(torch.randn(2 * B, 77, 768, generator=gen))
    x = (torch.randn(B, 4, hw, hw, generator=gen) * 3.0).cuda()
    s = torch.full((B,), 3.0, device="cuda")
    x2, s2 = torch.cat([x, x]).contiguous(), torch.cat([s, s]).contiguous()
    o_full, o_pair = torch.empty_like(x2), torch.empty_like(x2)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        u.forward(x2, s2, out=o_full); u.forward_pair(x, s, out=o_pair)
        torch.cuda.synchronize()
        gf, gp = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gf, stream=st):
            u.forward(x2, s2, out=o_full)
        nf, ff = u.last_launches, u.last_flops
        with torch.cuda.graph(gp, stream=st):
            u.forward_pair(x, s, out=o_pair)
        np_, fp = u.last_launches, u.last_flops
        r = []
        for _ in range(3):
            r.append((timed(gf), timed(gp)))
    tf, tp = statistics.median(a for a, _ in r), statistics.median(b for _, b in r)
    err = float((o_pair - o_full).norm() / o_full.norm())
    print(f"B={B} latent {hw}x{hw}: forward on cat([x,x]) {tf:.3f} ms ({nf} launches, {ff / 1e12:.3f} TFLOP)   CFG pair {tp:.3f} ms ({np_} launches, {fp / 1e12:.3f} TFLOP)"
          f"   {100 * (tp / tf - 1):+.1f} %   rel-L2 between them {err:.1e}", flush=True)
    del u
    torch.cuda.empty_cache()
